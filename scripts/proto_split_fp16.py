"""Would two fp16 planes (22 mantissa bits, three products a0b0 + a0b1 + a1b0) do for the estimator's forward what three bf16 planes
(six products) do now?  Same emulation as proto_split_bf16.py (fp64 arithmetic on the rounded planes), weights optionally pre-scaled by a
power of two per layer (exact; InstanceNorm follows, the accumulators are scaled back before the statistics) so that their low plane
stays out of fp16's subnormal range.   python scripts/proto_split_fp16.py"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = importlib.import_module("pytorch-deepfepe_amd")


def planes(x, n, dt):
    out, r = [], x.clone()
    for _ in range(n):
        p = r.to(dt).to(x.dtype)
        out.append(p)
        r = r - p
    return out


def split_mm(W, X, n, dt, order):
    Wp, Xp = planes(W, n, dt), planes(X, n, dt)
    acc = torch.zeros(W.shape[0], X.shape[1], dtype=torch.float64)
    for i in range(n):
        for j in range(n):
            if i + j <= order:
                acc += Wp[i] @ Xp[j]
    return acc


def run(net, x, mode, n=0, dt=None, order=None, wscale=False, f32acc=False):
    B, C0, N = x.shape
    mods = list(net.fw)
    h = x.permute(1, 0, 2).reshape(C0, B * N)
    i = 0
    while i < len(mods):
        conv = mods[i]
        W = conv.weight[:, :, 0]
        if mode == "f64":
            y = W.double() @ h.double()
        elif mode == "f32":
            y = (W.float() @ h.float())
        else:
            Wd = W.double()
            s = 1.0
            if wscale:  # largest |w| to [2^3, 2^4)
                s = 2.0 ** (3 - torch.floor(torch.log2(Wd.abs().max())).item())
            y = split_mm(Wd * s, h.double(), n, dt, order) / s
            if f32acc:
                y = y.float().double()
        if i + 2 < len(mods) and isinstance(mods[i + 1], torch.nn.InstanceNorm1d):
            inorm, act = mods[i + 1], mods[i + 2]
            y = y.view(W.shape[0], B, N)
            t = y.dtype
            mean = y.mean(2, keepdim=True); var = y.var(2, unbiased=False, keepdim=True)
            z = (y - mean) / torch.sqrt(var + inorm.eps) * inorm.weight.to(t)[:, None, None] + inorm.bias.to(t)[:, None, None]
            a = torch.where(z > 0, z, z * act.negative_slope)
            if mode == "split":
                a = a.float().double()
            h = a.reshape(W.shape[0], B * N)
            i += 3
        else:
            h = y + conv.bias.to(y.dtype)[:, None]
            i += 1
    return h.double()


for seed in (0, 1, 2):
    torch.manual_seed(seed)
    net = d.compat.ErrorEstimators.ErrorEstimator(4)
    d.synth.fill_params_deterministic(net, 1 + seed)
    x = torch.rand(8, 4, 100) * 2 - 1
    with torch.no_grad():
        t = run(net, x, "f64")
        f = run(net, x, "f32")
        print(f"seed {seed}: logit scale {t.abs().max().item():.3f}  fp32 stock vs fp64: {(f - t).abs().max().item():.2e}")
        for name, kw in (("bf16 3 planes / 6 products", dict(n=3, dt=torch.bfloat16, order=2)),
                         ("bf16 2 planes / 3 products", dict(n=2, dt=torch.bfloat16, order=1)),
                         ("fp16 2 planes / 3 products", dict(n=2, dt=torch.float16, order=1)),
                         ("fp16 2 planes / 3 products, weights pre-scaled", dict(n=2, dt=torch.float16, order=1, wscale=True)),
                         ("fp16 2 planes / 4 products, weights pre-scaled", dict(n=2, dt=torch.float16, order=2, wscale=True))):
            s = run(net, x, "split", **kw)
            print(f"   {name:52s} max {(s - t).abs().max().item():.2e}  rms {(s - t).pow(2).mean().sqrt().item():.2e}")
