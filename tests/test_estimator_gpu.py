"""'Next' row f-1: the fused evaluation of the weight estimator (channel-major GEMMs + one HIP pass for
InstanceNorm+LeakyReLU) against the stock PyTorch module with the same parameters.  GPU box only."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def relerr(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.parametrize("C,R,N", [(64, 5, 100), (1024, 3, 100), (7, 2, 8), (16, 4, 500)])
def test_inorm_lrelu_kernel_matches_torch(dfepe, C, R, N):
    g = torch.Generator().manual_seed(C + N)
    Y = (torch.randn(C, R, N, generator=g) * 3 + 1).to(DEV).requires_grad_(True)
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).to(DEV).requires_grad_(True)
    beta = (0.3 * torch.randn(C, generator=g)).to(DEV).requires_grad_(True)
    G = torch.randn(C, R, N, generator=g).to(DEV)
    A = dfepe.ops.inorm_lrelu(Y, gamma, beta, 1e-5, 0.01)
    (A * G).sum().backward()
    Yr, gr, br = (t.detach().clone().double().requires_grad_(True) for t in (Y, gamma, beta))
    x = Yr.permute(1, 0, 2)  # [R, C, N] = (batch, channels, points)
    ref = torch.nn.functional.leaky_relu(torch.nn.functional.instance_norm(x, weight=gr, bias=br, eps=1e-5), 0.01).permute(1, 0, 2)
    (ref * G.double()).sum().backward()
    assert relerr(A.detach().double(), ref.detach()) < 1e-5
    assert relerr(Y.grad.double(), Yr.grad) < 1e-4
    assert relerr(gamma.grad.double(), gr.grad) < 1e-4
    assert relerr(beta.grad.double(), br.grad) < 1e-4


def test_inorm_lrelu_rejects_unsupported_shapes(dfepe):
    with pytest.raises(dfepe.DfepeError):
        dfepe.ops.inorm_lrelu(torch.zeros(4, 2, 10, device=DEV), torch.ones(4, device=DEV), torch.zeros(4, device=DEV))  # N % 4 != 0


@pytest.mark.parametrize("cin", [4, 7])
def test_fused_estimator_equals_stock_module(dfepe, cin):
    B, N = 6, 100
    EE = dfepe.compat.ErrorEstimators
    stock = EE.ErrorEstimator(cin).to(DEV)
    dfepe.synth.fill_params_deterministic(stock, seed=3)
    fused = EE.FusedErrorEstimator(cin).to(DEV)
    fused.load_state_dict(stock.state_dict())
    assert list(fused.state_dict().keys()) == list(stock.state_dict().keys())
    g = torch.Generator().manual_seed(0)
    x = torch.rand(B, cin, N, generator=g).to(DEV)
    G = torch.randn(B, 1, N, generator=g).to(DEV)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya, yb = stock(xa), fused(xb)
    assert yb.shape == ya.shape == (B, 1, N)
    assert relerr(yb.detach(), ya.detach()) < 2e-4
    (ya * G).sum().backward()
    (yb * G).sum().backward()
    assert relerr(xb.grad, xa.grad) < 2e-3
    pa, pb = dict(stock.named_parameters()), dict(fused.named_parameters())
    for name in pa:
        # every parameter takes part in the graph (DistributedDataParallel needs that; the optimizer state then matches the
        # reference's): the biases that cancel in an InstanceNorm get the exact zero the stock module computes numerically
        assert pb[name].grad is not None, name
        if pb[name].grad.abs().max().item() == 0.0:
            assert name.endswith(".bias") and pa[name].grad.abs().max().item() < 1e-3 * max(1.0, G.abs().sum().item()) * 1e-3
            continue
        assert relerr(pb[name].grad, pa[name].grad) < 5e-3, name


def test_batchnorm_variant_is_not_chunked_in_training_mode(dfepe):
    """Batch statistics span the whole batch: the chunked evaluation (MIOpen work-around) only applies without BatchNorm in
    training mode."""
    EE = dfepe.compat.ErrorEstimators
    net = EE.ErrorEstimator(4, if_bn=True).to(DEV)
    net.max_chunk = 4
    x = torch.rand(10, 4, 20, device=DEV)
    net.train()
    y = net(x)
    ref = net.fw(x)
    assert y.shape == (10, 1, 20) and torch.isfinite(y).all() and relerr(y.detach(), ref.detach()) < 1e-5
