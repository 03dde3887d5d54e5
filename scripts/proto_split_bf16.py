"""How many bf16 planes does the estimator need?  Emulates the split-bf16 products of the fused estimator chain on the CPU
(fp64 arithmetic on bf16-rounded planes = what fp32-accumulating MFMAs give up to accumulation rounding) and compares the
logits of one ErrorEstimator with the fp32 stock evaluation and the fp64 truth.   python scripts/proto_split_bf16.py"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = importlib.import_module("pytorch-deepfepe_amd")


def planes(x, n):
    out, r = [], x.clone()
    for _ in range(n):
        p = r.to(torch.bfloat16).to(x.dtype)
        out.append(p)
        r = r - p
    return out


def split_mm(W, X, n):
    Wp, Xp = planes(W, n), planes(X, n)
    acc = torch.zeros(W.shape[0], X.shape[1], dtype=torch.float64)
    for i in range(n):
        for j in range(n - i):
            acc += Wp[i] @ Xp[j]
    return acc


def run(net, x, mode, n=0):
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
            y = split_mm(W.double(), h.double(), n)
        if i + 2 < len(mods) and isinstance(mods[i + 1], torch.nn.InstanceNorm1d):
            inorm, act = mods[i + 1], mods[i + 2]
            y = y.view(W.shape[0], B, N)
            dt = y.dtype
            mean = y.mean(2, keepdim=True); var = y.var(2, unbiased=False, keepdim=True)
            z = (y - mean) / torch.sqrt(var + inorm.eps) * inorm.weight.to(dt)[:, None, None] + inorm.bias.to(dt)[:, None, None]
            a = torch.where(z > 0, z, z * act.negative_slope)
            if mode == "split":
                a = a.float().double()  # activations are handed on as fp32 values (then split again)
            h = a.reshape(W.shape[0], B * N)
            i += 3
        else:
            h = y + conv.bias.to(y.dtype)[:, None]
            i += 1
    return h.double()


torch.manual_seed(0)
net = d.compat.ErrorEstimators.ErrorEstimator(4)
d.synth.fill_params_deterministic(net, 1)
x = torch.rand(8, 4, 100)
with torch.no_grad():
    t = run(net, x, "f64")
    f = run(net, x, "f32")
    print("logit scale", t.abs().max().item(), t.std().item())
    print("fp32 stock vs fp64:", (f - t).abs().max().item())
    for n in (1, 2, 3):
        s = run(net, x, "split", n)
        print(f"{n} plane(s) ({n*(n+1)//2} products) vs fp64:", (s - t).abs().max().item())
