"""Round 6 pieces of the weight estimator (csrc/est_gemm.hip; deepFEPE/models/ErrorEstimators.py:47-64 at the reference's own shapes,
N = 1000-2000 points and 4-12 pairs per batch, deepFEPE/configs/kitti_corr_baseline.yaml:12-13): split-K plain products, the
register-resident normalisation and its adjoint (which add the split-K partials), every layer's weight gradient in one launch, the
input gradient stored by its GEMM's epilogue, parameters prepared once per model forward -- each against float64 or against the launch
it replaces -- and the whole estimator / the whole model at N = 1000.  GPU box only."""
import ctypes

import pytest
import torch

from test_estimator_mfma_gpu import DEV, _split, _split_f16, planes_to_f64, relerr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,K,pairs,N,S", [(512, 1024, 8, 100, 8), (256, 512, 3, 100, 4), (128, 1024, 2, 1000, 5), (64, 64, 4, 37, 2)])
def test_splitk_products_add_up_to_the_plain_product(dfepe, M, K, pairs, N, S):
    """S workgroups per tile share the K steps: the S partial products [S][cols][M] add up to the plain product (fp32 partial sums in
    another order: 1e-6 of the largest entry), for the forward's fp16 planes (weights split scaled) and the backward's bf16 planes."""
    lib = dfepe._lib.lib()
    cols = pairs * N
    g = torch.Generator().manual_seed(M + K)
    A = (torch.randn(M, K, generator=g) / K ** 0.5).to(DEV)
    X = torch.randn(cols, K, generator=g).to(DEV)
    Ah, word, _ = _split_f16(dfepe, A, K, scaled=True)
    Xh, _, _ = _split_f16(dfepe, X, K)
    plain = torch.zeros(cols, M, device=DEV)
    assert lib.dfepe_est_gemm_nt_f16(Ah.data_ptr(), M * K, Xh.data_ptr(), cols * K, M, cols, K, word.data_ptr(), plain.data_ptr(), M, None) == 0
    parts = torch.full((S, cols, M), float("nan"), device=DEV)
    assert lib.dfepe_est_gemm_nt_f16_splitk(Ah.data_ptr(), M * K, Xh.data_ptr(), cols * K, M, cols, K, word.data_ptr(), parts.data_ptr(), M, S,
                                            cols * M, None) == 0
    torch.cuda.synchronize()
    assert relerr(parts.sum(0), plain) < 1e-6
    assert relerr(parts.double().sum(0), X.double() @ A.double().t()) < 2e-6
    Ab, Xb = _split(dfepe, A, K, 2), _split(dfepe, X, K, 2)
    plain_b = torch.zeros(cols, M, device=DEV)
    assert lib.dfepe_est_gemm_nt(Ab.data_ptr(), M * K, Xb.data_ptr(), cols * K, M, cols, K, 2, plain_b.data_ptr(), M, None) == 0
    parts_b = torch.full((S, cols, M), float("nan"), device=DEV)
    assert lib.dfepe_est_gemm_nt_splitk(Ab.data_ptr(), M * K, Xb.data_ptr(), cols * K, M, cols, K, parts_b.data_ptr(), M, S, cols * M, None) == 0
    torch.cuda.synchronize()
    assert relerr(parts_b.sum(0), plain_b) < 1e-6
    # more slices than K steps, or partials that would overlap: refused
    assert lib.dfepe_est_gemm_nt_splitk(Ab.data_ptr(), M * K, Xb.data_ptr(), cols * K, M, cols, K, parts_b.data_ptr(), M, K // 32 + 1, cols * M, None) == -1
    assert lib.dfepe_est_gemm_nt_splitk(Ab.data_ptr(), M * K, Xb.data_ptr(), cols * K, M, cols, K, parts_b.data_ptr(), M, 2, cols * M - 4, None) == -1


@pytest.mark.parametrize("C,N,pairs,S", [(64, 37, 3, 1), (128, 1000, 2, 1), (1024, 256, 2, 3), (256, 2000, 2, 1), (32, 5, 4, 1), (64, 100, 5, 8),
                                         (512, 100, 8, 4), (64, 1024, 1, 2), (32, 1025, 2, 1), (96, 2048, 1, 2)])
def test_register_resident_norm_matches_float64(dfepe, C, N, pairs, S):
    """dfepe_est_norm_fwd_r on S partial products: planes = split(leaky_relu(instance_norm(sum of the partials))), rstd -- one launch,
    the pair's block in registers (every template: N <= 128 / 512 / 1024 / 2048, the ragged last row group included)."""
    lib = dfepe._lib.lib()
    cols, ld = pairs * N, C + 8
    g = torch.Generator().manual_seed(C + N)
    Yd = (torch.randn(cols, ld, generator=g) * 3 + 40.0).to(DEV)  # a mean far above the deviation: the two-pass variance matters
    w = torch.rand(S, generator=g) + 0.5
    w = (w / w.sum()).to(DEV)
    parts = (Yd[None] * w[:, None, None]).contiguous()  # S partials that add up to Yd (to fp32 rounding)
    Ysum = parts[0].clone()
    for s in range(1, S):
        Ysum += parts[s]  # the kernel's order of additions
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).to(DEV)
    beta = (0.3 * torch.randn(C, generator=g)).to(DEV)
    out = torch.zeros(2, cols, C, device=DEV, dtype=torch.float16)
    out_b = torch.zeros(2, cols, C, device=DEV, dtype=torch.bfloat16)
    rstd = torch.zeros(pairs, C, device=DEV)
    rc = lib.dfepe_est_norm_fwd_r(parts.data_ptr(), ld, S, cols * ld, C, pairs, N, gamma.data_ptr(), beta.data_ptr(), 1e-5, 0.01, out.data_ptr(),
                                  cols * C, out_b.data_ptr(), cols * C, rstd.data_ptr(), None)
    assert rc == 0
    torch.cuda.synchronize()
    Y = Ysum[:, :C].double().view(pairs, N, C).permute(0, 2, 1)
    var = Y.var(2, unbiased=False)
    ref = torch.nn.functional.leaky_relu((Y - Y.mean(2, keepdim=True)) / torch.sqrt(var + 1e-5)[..., None] * gamma.double()[None, :, None]
                                         + beta.double()[None, :, None], 0.01)
    got = planes_to_f64(out).view(pairs, N, C).permute(0, 2, 1)
    assert float((got - ref).abs().max()) < 2e-5 * float(ref.abs().max())
    assert float((planes_to_f64(out_b) - planes_to_f64(out)).abs().max()) <= 2.0 ** -16 * float(ref.abs().max())
    assert relerr(rstd, 1.0 / torch.sqrt(var + 1e-5)) < 5e-6
    # without the backward's planes (a forward under no_grad): same fp16 planes
    out2 = torch.zeros_like(out)
    assert lib.dfepe_est_norm_fwd_r(parts.data_ptr(), ld, S, cols * ld, C, pairs, N, gamma.data_ptr(), beta.data_ptr(), 1e-5, 0.01, out2.data_ptr(),
                                    cols * C, None, 0, rstd.data_ptr(), None) == 0
    torch.cuda.synchronize()
    assert torch.equal(out2, out)
    assert lib.dfepe_est_norm_fwd_r(parts.data_ptr(), ld, S, cols * ld, C, pairs, 2049, gamma.data_ptr(), beta.data_ptr(), 1e-5, 0.01, out2.data_ptr(),
                                    cols * C, None, 0, rstd.data_ptr(), None) == -1  # beyond the registers: the strided kernel's job


@pytest.mark.parametrize("C,N,pairs,head,S", [(64, 37, 3, False, 1), (256, 1000, 2, True, 1), (1024, 250, 2, False, 3), (64, 100, 4, True, 1),
                                              (128, 100, 8, False, 8), (64, 2000, 2, False, 2), (32, 1025, 1, False, 1), (64, 5, 2, False, 1)])
def test_register_resident_adjoint_matches_autograd(dfepe, C, N, pairs, head, S):
    """dfepe_est_in_bwd_r: the InstanceNorm + LeakyReLU adjoint from S partial data gradients (or the head's rank-one form) against
    float64 autograd: dY planes, per-pair d gamma / d beta."""
    lib = dfepe._lib.lib()
    cols = pairs * N
    g = torch.Generator().manual_seed(C + N)
    Y = (torch.randn(pairs, C, N, generator=g, dtype=torch.float64) * 2 + 0.5).requires_grad_(True)
    gamma = (1 + 0.2 * torch.randn(C, generator=g, dtype=torch.float64)).requires_grad_(True)
    beta = (0.3 * torch.randn(C, generator=g, dtype=torch.float64)).requires_grad_(True)
    a = torch.nn.functional.leaky_relu(torch.nn.functional.instance_norm(Y, weight=gamma, bias=beta, eps=1e-5), 0.01)
    if head:
        dl = torch.randn(cols, generator=g, dtype=torch.float64)
        wh = torch.randn(C, generator=g, dtype=torch.float64)
        G = (dl.view(pairs, 1, N) * wh.view(1, C, 1))
    else:
        G = torch.randn(pairs, C, N, generator=g, dtype=torch.float64)
    (a * G).sum().backward()
    P = _split(dfepe, a.detach().permute(0, 2, 1).reshape(cols, C).float().to(DEV), C)
    rstd = (1.0 / torch.sqrt(Y.detach().var(2, unbiased=False) + 1e-5)).float().to(DEV).contiguous()
    dA = G.permute(0, 2, 1).reshape(cols, C).float().to(DEV).contiguous()
    w = torch.rand(S, generator=g) + 0.5
    parts = (dA[None] * (w / w.sum()).to(DEV)[:, None, None]).contiguous()
    dY = torch.zeros(2, cols, C, device=DEV, dtype=torch.bfloat16)
    dg, db = torch.zeros(pairs, C, device=DEV), torch.zeros(pairs, C, device=DEV)
    gm, bt = gamma.detach().float().to(DEV), beta.detach().float().to(DEV)
    if head:
        dl_d, wh_d = dl.float().to(DEV), wh.float().to(DEV)
        rc = lib.dfepe_est_in_bwd_r(None, 0, 1, 0, dl_d.data_ptr(), wh_d.data_ptr(), P.data_ptr(), cols * C, rstd.data_ptr(), gm.data_ptr(),
                                    bt.data_ptr(), 0.01, C, pairs, N, dY.data_ptr(), cols * C, dg.data_ptr(), db.data_ptr(), None)
    else:
        rc = lib.dfepe_est_in_bwd_r(parts.data_ptr(), C, S, cols * C, None, None, P.data_ptr(), cols * C, rstd.data_ptr(), gm.data_ptr(),
                                    bt.data_ptr(), 0.01, C, pairs, N, dY.data_ptr(), cols * C, dg.data_ptr(), db.data_ptr(), None)
    assert rc == 0
    torch.cuda.synchronize()
    assert relerr(planes_to_f64(dY).cpu(), Y.grad.permute(0, 2, 1).reshape(cols, C)) < 5e-5
    assert relerr(dg.sum(0).cpu(), gamma.grad) < 5e-5
    assert relerr(db.sum(0).cpu(), beta.grad) < 5e-5


def test_weight_gradients_of_all_layers_in_one_launch(dfepe):
    """dfepe_est_gemm_tn_multi: the five layers' split-K partials from ONE launch are bit for bit those of five dfepe_est_gemm_tn launches
    (slice counts on and off the multiple-of-eight XCD order; the reference's widths at 8 pairs x 100 points)."""
    lib = dfepe._lib.lib()
    cols = 800
    shapes = [(64, 32, 3), (128, 64, 3), (1024, 128, 8), (512, 1024, 16), (256, 512, 1)]
    g = torch.Generator().manual_seed(5)
    dYs, Xs, single, multi = [], [], [], []
    for Co, Ci, sl in shapes:
        dY = _split(dfepe, torch.randn(cols, Co, generator=g).to(DEV), Co, 2)
        X = _split(dfepe, torch.randn(cols, Ci, generator=g).to(DEV), Ci, 2)
        p1 = torch.full((sl, Co, Ci), float("nan"), device=DEV)
        assert lib.dfepe_est_gemm_tn(dY.data_ptr(), cols * Co, Co, X.data_ptr(), cols * Ci, Ci, cols, sl, p1.data_ptr(), None) == 0
        dYs.append(dY); Xs.append(X); single.append(p1); multi.append(torch.full((sl, Co, Ci), float("nan"), device=DEV))
    n = len(shapes)
    vp = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
    sz = lambda vs: (ctypes.c_size_t * n)(*vs)
    it = lambda vs: (ctypes.c_int * n)(*vs)
    rc = lib.dfepe_est_gemm_tn_multi(n, vp(dYs), sz([cols * s[0] for s in shapes]), it([s[0] for s in shapes]), vp(Xs), sz([cols * s[1] for s in shapes]),
                                     it([s[1] for s in shapes]), cols, it([s[2] for s in shapes]), vp(multi), None)
    assert rc == 0
    torch.cuda.synchronize()
    for a, b in zip(single, multi):
        assert torch.equal(a, b)
    ref = planes_to_f64(dYs[3]).t() @ planes_to_f64(Xs[3])
    assert relerr(multi[3].double().sum(0), ref) < 1e-5


@pytest.mark.parametrize("C0,pairs,N", [(7, 5, 100), (4, 3, 37), (7, 2, 1000)])
def test_input_gradient_is_stored_by_the_gemm_epilogue(dfepe, C0, pairs, N):
    """dfepe_est_gemm_nt_gx writes the first layer's data gradient as gx [pairs][C0][N] (or channel-major [C0][pairs][N]): bit for bit
    the plain product transposed and cropped to the C0 real input channels."""
    lib = dfepe._lib.lib()
    cols, K0, Co = pairs * N, 32, 64
    g = torch.Generator().manual_seed(C0 + N)
    WT = _split(dfepe, torch.randn(K0, Co, generator=g).to(DEV), Co, 2)  # W^T [K0][Co]
    dY = _split(dfepe, torch.randn(cols, Co, generator=g).to(DEV), Co, 2)
    dA = torch.zeros(cols, K0, device=DEV)
    assert lib.dfepe_est_gemm_nt(WT.data_ptr(), K0 * Co, dY.data_ptr(), cols * Co, K0, cols, Co, 2, dA.data_ptr(), K0, None) == 0
    gx = torch.full((pairs, C0, N), float("nan"), device=DEV)
    assert lib.dfepe_est_gemm_nt_gx(WT.data_ptr(), K0 * Co, dY.data_ptr(), cols * Co, K0, cols, Co, gx.data_ptr(), C0, N, C0 * N, N, None) == 0
    gxc = torch.full((C0, pairs, N), float("nan"), device=DEV)  # the model's channel-major layout: same values, other strides
    assert lib.dfepe_est_gemm_nt_gx(WT.data_ptr(), K0 * Co, dY.data_ptr(), cols * Co, K0, cols, Co, gxc.data_ptr(), C0, N, N, pairs * N, None) == 0
    torch.cuda.synchronize()
    assert torch.equal(gx, dA[:, :C0].reshape(pairs, N, C0).permute(0, 2, 1).contiguous())
    assert torch.equal(gxc.permute(1, 0, 2), gx)


def test_channel_major_input_views_are_read_and_differentiated_in_place(dfepe):
    """The model hands the estimator [B, C, N] VIEWS of channel-major [C, B, N] buffers (ops.estimator_input): logits and parameter
    gradients equal those of the dense copy bit for bit, and the input gradient comes back in the view's own layout (its rows dense)."""
    EE = dfepe.compat.ErrorEstimators
    net = EE.FusedErrorEstimator(7).to(DEV)
    dfepe.synth.fill_params_deterministic(net, seed=4)
    g = torch.Generator().manual_seed(3)
    B, N = 6, 100
    store = torch.rand(7, B, N, generator=g).to(DEV)
    G = torch.randn(B, 1, N, generator=g).to(DEV)
    outs = []
    for dense in (False, True):
        net.zero_grad(set_to_none=True)
        x = store.permute(1, 0, 2)
        x = (x.contiguous() if dense else x).detach().requires_grad_(True)
        assert x.is_contiguous() == dense
        y = net(x)
        (y * G).sum().backward()
        outs.append((y.detach().clone(), x.grad, [p.grad.clone() for p in net.parameters()]))
    (ya, ga, pa), (yb, gb, pb) = outs
    assert torch.equal(ya, yb) and torch.equal(ga, gb)
    assert ga.stride() == (N, B * N, 1) and ga[:, 4, :].is_contiguous() and gb.is_contiguous()
    for a, b in zip(pa, pb):
        assert torch.equal(a, b)


def _fused_and_stock(dfepe, cin, seed):
    EE = dfepe.compat.ErrorEstimators
    stock = EE.ErrorEstimator(cin)
    dfepe.synth.fill_params_deterministic(stock, seed=seed)
    fused = EE.FusedErrorEstimator(cin).to(DEV)
    fused.load_state_dict(stock.state_dict())
    return fused, stock.double()


@pytest.mark.parametrize("cin,B,N", [(7, 8, 1000), (4, 4, 1000), (7, 12, 2000), (7, 8, 100), (4, 32, 100), (7, 3, 300)])
def test_whole_estimator_at_the_reference_shapes_matches_float64(dfepe, cin, B, N):
    """The stack as a train_good.py user runs it (N = 1000-2000 points, 4-12 pairs: plain products, K-heavy layers split over K,
    register-resident normalisations, one weight-gradient launch) and the small-batch N = 100 path (fused epilogues for the K-light
    layers, split-K for the K-heavy ones) against the stock module in float64.  Logits to the fp32 class (3e-6); gradients to the two-plane class
    (1e-4 of their norm) when the float64 run keeps every pre-activation 2e-6 away from the LeakyReLU kink, 5e-3 otherwise (a few flipped
    elements among >= 1e5: the stock fp32 module deviates the same way there, tests/test_estimator_mfma_gpu.py)."""
    fused, stock = _fused_and_stock(dfepe, cin, 9)
    margin = [float("inf")]
    hooks = [m.register_forward_hook(lambda _m, _i, o: margin.__setitem__(0, min(margin[0], float(o.detach().abs().min()))))
             for m in stock.fw if isinstance(m, torch.nn.InstanceNorm1d)]
    g = torch.Generator().manual_seed(N + B)
    x = torch.rand(B, cin, N, generator=g)
    G = torch.randn(B, 1, N, generator=g)
    xa = x.double().requires_grad_(True)
    xb = x.to(DEV).requires_grad_(True)
    ya = stock(xa)
    for h in hooks:
        h.remove()
    yb = fused(xb)
    assert type(yb.grad_fn).__name__.startswith("_EstimatorPackedFunction")
    assert float((yb.detach().cpu().double() - ya.detach()).abs().max()) < 6e-6
    (ya * G.double()).sum().backward()
    (yb * G.to(DEV)).sum().backward()
    tol = 1e-4 if margin[0] > 2e-6 else 5e-3  # (12 x 2000 points: a handful of the 2.4e7 pre-activations of the 1024-wide layer flip)
    rel2 = lambda a, b: float((a.cpu().double() - b).norm() / b.norm())
    assert rel2(xb.grad, xa.grad) < tol
    pa, pb = dict(stock.named_parameters()), dict(fused.named_parameters())
    worst = 0.0
    for name in pa:
        assert pb[name].grad is not None, name
        if pb[name].grad.abs().max().item() == 0.0:
            assert name.endswith(".bias") and pa[name].grad.abs().max().item() < 1e-9
            continue
        worst = max(worst, rel2(pb[name].grad, pa[name].grad))
        assert rel2(pb[name].grad, pa[name].grad) < tol, name
    print(f"B={B} N={N}: kink margin {margin[0]:.1e}, worst parameter-gradient error {worst:.1e}")


def test_a_kink_case_is_no_worse_than_the_stock_fp32_module(dfepe):
    """VERDICT r5 hygiene: the whole-estimator gradient tests run on searched kink-free seeds; this one runs a case whose float64 run
    has pre-activations within fp32 rounding of the LeakyReLU kink (12 pairs x 1000 points, parameter seed 9: margin ~4e-8, 25 of 60
    random cases are like that: profiles/r05_estimator_stress.log).  There ANY fp32 evaluation may take the other branch of an element
    and move a gradient by a finite amount -- so the claim is relative: every parameter gradient of the matrix-core stack is no further
    from the float64 truth than 3 x the stock fp32 module's worst on the same case (+ the two-plane class), and the logits are closer."""
    cin, B, N = 7, 12, 1000
    fused, stock64 = _fused_and_stock(dfepe, cin, 9)
    stock32 = dfepe.compat.ErrorEstimators.ErrorEstimator(cin).to(DEV)
    stock32.load_state_dict(fused.state_dict())
    g = torch.Generator().manual_seed(N + B)
    x = torch.rand(B, cin, N, generator=g)
    G = torch.randn(B, 1, N, generator=g)
    margin = [float("inf")]
    hooks = [m.register_forward_hook(lambda _m, _i, o: margin.__setitem__(0, min(margin[0], float(o.detach().abs().min()))))
             for m in stock64.fw if isinstance(m, torch.nn.InstanceNorm1d)]
    ya = stock64(x.double())
    for h in hooks:
        h.remove()
    (ya * G.double()).sum().backward()
    ref = {n: p.grad for n, p in stock64.named_parameters()}
    assert margin[0] < 1e-6, margin  # it IS a kink case
    rel = lambda a, b: float((a.cpu().double() - b).norm() / b.norm())
    res = {}
    for label, model in (("fused", fused), ("stock", stock32)):
        y = model(x.to(DEV))
        (y * G.to(DEV)).sum().backward()
        res[label] = (float((y.detach().cpu().double() - ya.detach()).abs().max()),
                      {n: rel(p.grad, ref[n]) for n, p in model.named_parameters() if float(ref[n].abs().max()) > 1e-9 and float(p.grad.abs().max()) > 0})
    worst_f, worst_s = max(res["fused"][1].values()), max(res["stock"][1].values())
    print(f"kink margin {margin[0]:.1e}: logits fused {res['fused'][0]:.1e} / stock {res['stock'][0]:.1e}; worst gradient fused {worst_f:.1e} / stock {worst_s:.1e}")
    assert res["fused"][0] <= max(res["stock"][0], 4e-6)
    assert worst_f <= 3 * worst_s + 1e-4
    assert worst_f < 1e-2


@pytest.mark.parametrize("B,N,calls", [(8, 100, 4), (4, 1000, 3)])
def test_parameters_prepared_once_serve_every_call_of_a_forward(dfepe, B, N, calls):
    """FusedErrorEstimator.shared_parameters(): k calls on ONE preparation (packed parameter vector + weight planes) give bit for bit
    the logits and input gradients of k self-prepared calls, and parameter gradients equal to their sum (autograd adds k packed vectors
    instead of k gradients per parameter: another order of the same fp32 additions)."""
    EE = dfepe.compat.ErrorEstimators
    net = EE.FusedErrorEstimator(7).to(DEV)
    dfepe.synth.fill_params_deterministic(net, seed=4)
    g = torch.Generator().manual_seed(B * N)
    xs = [torch.rand(B, 7, N, generator=g).to(DEV) for _ in range(calls)]
    Gs = [torch.randn(B, 1, N, generator=g).to(DEV) for _ in range(calls)]

    def run(shared):
        net.zero_grad(set_to_none=True)
        xr = [x.clone().requires_grad_(True) for x in xs]
        if shared:
            with net.shared_parameters():
                assert net._prepared is not None
                ys = [net(x) for x in xr]
            assert net._prepared is None
        else:
            ys = [net(x) for x in xr]
        sum((y * G).sum() for y, G in zip(ys, Gs)).backward()
        return [y.detach().clone() for y in ys], [x.grad.clone() for x in xr], [p.grad.clone() for p in net.parameters()]

    ya, xa, pa = run(False)
    yb, xb, pb = run(True)
    for a, b in zip(ya + xa, yb + xb):
        assert torch.equal(a, b)
    for (name, _), a, b in zip(net.named_parameters(), pa, pb):
        assert float((a - b).abs().max()) <= 2e-6 * float(a.abs().max()) + 1e-30, name
    # the prepared parameters are only used for the objects they were made of
    with net.shared_parameters():
        other = EE.FusedErrorEstimator(7).to(DEV)
        dfepe.synth.fill_params_deterministic(other, seed=5)
        other._prepared = net._prepared
        with torch.no_grad():
            assert not torch.equal(other(xs[0]), net(xs[0]))
            alone = EE.FusedErrorEstimator(7).to(DEV)
            alone.load_state_dict(other.state_dict())
            assert torch.equal(other(xs[0]), alone(xs[0]))


@pytest.mark.parametrize("B,N", [(4, 1000), (6, 2000)])
def test_whole_model_at_the_reference_point_counts_through_the_fused_estimators(dfepe, B, N):
    """compat.DeepFNet (depth 3, the shapes of deepFEPE/configs/kitti_corr_baseline.yaml: 1000-2000 points, a few pairs) + F-loss + pose
    loss + backward with the matrix-core estimators against the SAME model on the stock PyTorch estimators (identical parameters, both
    fp32): logits, F per layer, both losses, and every parameter gradient.  The estimators' heads are scaled down (near-uniform
    weights on an outlier-free scene): with the deterministic random parameters as they are the fits are garbage whose eigenvector
    adjoints amplify the 1e-4 distance between two fp32 evaluations of the logits to percents of a gradient
    (scripts/whole_model_fd_check.py: the objective's own central differences then sit closer to this package's gradient than to the
    stock path's) -- no yardstick.  The N = 100 golden test pins the same code path to the reference's gradients."""
    depth = 3
    D = dfepe.compat.DeepFNet
    net = D.DeepFNet(depth=depth, image_size=[376, 1241, 3], if_quality=False).to(DEV)
    dfepe.synth.fill_params_deterministic(net, seed=5)
    with torch.no_grad():
        for est in (net.input_weights, net.update_weights):
            est.fw[-1].weight.mul_(0.05)
    ref = D.DeepFNet(depth=depth, image_size=[376, 1241, 3], if_quality=False, fused_estimator=False).to(DEV)
    ref.load_state_dict(net.state_dict())
    sc = dfepe.synth.make_scene(B, N, seed=3, outlier_ratio=0.0, noise_px=0.5)
    b = {k: sc[k].to(DEV) for k in ("matches_xy_ori", "pts1_virt_ori", "pts2_virt_ori", "Ks", "delta_Rtijs_4_4", "qs_cam", "ts_cam")}
    tg = dfepe.compat.train_good_utils

    def step(model):
        lp = {"depth": depth, "clamp_at": 0.02, "if_tri_depth": False, "if_sample_loss": False, "topK": 8, "matches_good_unique_nums": None}
        outs = model({"matches_xy_ori": b["matches_xy_ori"], "matches_good_unique_nums": None, "t_scene_scale": None})
        losses, _, _, _, _, _, E_layers = tg.get_all_loss_DeepF(outs, b["pts1_virt_ori"], b["pts2_virt_ori"], b["Ks"], lp, get_residual_summaries=False)
        geo = tg.get_Rt_loss(E_layers, b["Ks"], None, None, b["delta_Rtijs_4_4"], b["qs_cam"], b["ts_cam"], device=DEV)
        lq = torch.clamp(torch.stack(geo["q_l2_error_layers_list"]), 0.0, 0.1).mean()
        lt = torch.clamp(torch.stack(geo["t_l2_error_layers_list"]), 0.0, 0.5).mean()
        loss = losses["loss_F"] + lq + 0.1 * lt
        loss.backward()
        return outs, (float(losses["loss_F"].detach()), float(lq.detach()), float(lt.detach()))

    oa, la = step(net)
    assert type(oa["logits_layers"][1].grad_fn).__name__.startswith("_EstimatorPackedFunction")
    ob, lb = step(ref)
    for l in range(depth):
        assert float((oa["logits_layers"][l].detach() - ob["logits_layers"][l].detach()).abs().max()) < 1e-4, l
        Fa, Fb = oa["out_layers"][l].detach().flatten(1), ob["out_layers"][l].detach().flatten(1)
        Fa, Fb = Fa / Fa.norm(dim=1, keepdim=True), Fb / Fb.norm(dim=1, keepdim=True)
        s = torch.sign((Fa * Fb).sum(1, keepdim=True))
        assert float((Fa * s - Fb).norm(dim=1).max()) < 2e-5, l
    for x, y in zip(la, lb):
        assert abs(x - y) < 1e-4 * abs(y) + 1e-7, (la, lb)
    top = max(float(p.grad.norm()) for p in ref.parameters())
    worst = 0.0
    for (name, pa), (_, pb) in zip(net.named_parameters(), ref.named_parameters()):
        if float(pa.grad.abs().max()) == 0.0:  # biases that cancel in an InstanceNorm: exact zero here, rounding noise there
            assert name.endswith(".bias") and float(pb.grad.norm()) < 1e-5 * top, name
            continue
        dist = float((pa.grad - pb.grad).norm())
        worst = max(worst, dist / max(float(pb.grad.norm()), 1e-3 * top))
        assert dist < 1e-2 * float(pb.grad.norm()) + 1e-4 * top, (name, dist, float(pb.grad.norm()), top)
    print(f"whole model at {B} x {N}: worst parameter-gradient distance between the fused and the stock estimators {worst:.1e}")
