"""Reproducer hunt for the ROCm 7.2 hipGraph fault this package works around with DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 (pytorch-deepfepe_amd/__init__.py):
a captured forward + backward replays correctly ONCE and then leaves some gradient buffers unwritten.  VERDICT r5 item 5 asks whether the fault
is the runtime's or this library's launches from the autograd engine's thread.  The script captures forward + torch.autograd.grad of several
models, replays each graph six times over three inputs and reports the first replay whose gradients differ from the eager ones:

  torch-modules   the estimator's architecture in stock torch.nn modules (Conv1d(k=1) -> InstanceNorm1d -> LeakyReLU x5 -> Conv1d)
  torch-function  the same arithmetic behind ONE torch.autograd.Function whose backward issues ~80 small torch kernels (matmuls, per-pair
                  reductions, fills, a final stack of column sums) from the autograd engine's thread, on caller-side temporaries -- the launch
                  pattern of this package's per-launch host code, with no kernel of this package
  dfepe-launch    this package's estimator through its per-launch host code (estimator.USE_PASS = False: ~75 launches per call, torch reductions
                  for the head bias, torch allocations per launch)
  dfepe-pass      ... through one library call per pass (the default since round 5; round 6: parameters packed by torch.cat)

Run it with the runtime's default and with the workaround:
    DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 python scripts/repro_graph_packet_capture.py
    DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 python scripts/repro_graph_packet_capture.py
"""
import importlib, os, sys
import torch
import torch.nn as nn
import torch.nn.functional as F

DEV = "cuda:0"
B, N = int(os.environ.get("DBG_B", "48")), 100
WIDTHS = [4, 64, 128, 1024, 512, 256]


def stock():
    layers = []
    for ci, co in zip(WIDTHS[:-1], WIDTHS[1:]):
        layers += [nn.Conv1d(ci, co, 1), nn.InstanceNorm1d(co, affine=True), nn.LeakyReLU()]
    layers.append(nn.Conv1d(WIDTHS[-1], 1, 1))
    return nn.Sequential(*layers).to(DEV)


class Chain(torch.autograd.Function):
    """The whole stack as one node: forward and backward as explicit small torch kernels, temporaries allocated per launch."""

    @staticmethod
    def forward(ctx, x, *params):
        n = (len(params) - 2) // 4
        a = x
        saved = []
        for l in range(n):
            W, _b, g, bt = params[4 * l:4 * l + 4]
            y = torch.matmul(W[:, :, 0], a)                       # [B, Co, N]
            mu = y.mean(2, keepdim=True)
            var = (y - mu).square().mean(2, keepdim=True)
            rstd = torch.rsqrt(var + 1e-5)
            xh = (y - mu) * rstd
            z = xh * g[None, :, None] + bt[None, :, None]
            saved += [a, xh, rstd, z]
            a = F.leaky_relu(z, 0.01)
        Wh, bh = params[-2], params[-1]
        out = torch.matmul(Wh[:, :, 0], a) + bh[None, :, None]
        ctx.n = n
        ctx.save_for_backward(*params, *saved, a)
        return out

    @staticmethod
    def backward(ctx, go):
        n = ctx.n
        sv = ctx.saved_tensors
        params, rest = sv[:4 * n + 2], sv[4 * n + 2:]
        a_last = rest[-1]
        grads = [None] * (4 * n + 2)
        Wh = params[-2]
        grads[-1] = go.sum((0, 2))
        grads[-2] = torch.einsum("bon,bcn->oc", go, a_last).unsqueeze(2)
        da = torch.matmul(Wh[:, :, 0].t(), go)
        for l in range(n - 1, -1, -1):
            W, _b, g, _bt = params[4 * l:4 * l + 4]
            a_in, xh, rstd, z = rest[4 * l:4 * l + 4]
            dz = torch.where(z > 0, da, da * 0.01)
            grads[4 * l + 3] = dz.sum((0, 2))
            grads[4 * l + 2] = (dz * xh).sum((0, 2))
            dxh = dz * g[None, :, None]
            dy = rstd * (dxh - dxh.mean(2, keepdim=True) - xh * (dxh * xh).mean(2, keepdim=True))
            grads[4 * l] = torch.einsum("bon,bcn->oc", dy, a_in).unsqueeze(2)
            grads[4 * l + 1] = torch.zeros_like(_b)
            da = torch.matmul(W[:, :, 0].t(), dy)
        return (da, *grads)


def flat_params(net):
    mods = list(net)
    out, i = [], 0
    while i + 2 < len(mods):
        out += [mods[i].weight, mods[i].bias, mods[i + 1].weight, mods[i + 1].bias]
        i += 3
    return out + [mods[i].weight, mods[i].bias]


def hunt(label, run, params):
    xs = [torch.randn(B, WIDTHS[0], N, device=DEV) for _ in range(3)]
    x_static = xs[0].clone()
    refs = [tuple(g.clone() for g in run(x)) for x in xs]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        run(x_static)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        grads = run(x_static)
    first_bad, worst = None, 0.0
    for it in range(6):
        k = it % 3
        x_static.copy_(xs[k])
        g.replay()
        torch.cuda.synchronize()
        rel = max(float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)) for a, b in zip(grads, refs[k]))
        worst = max(worst, rel)
        if rel > 1e-4 and first_bad is None:
            first_bad = it
    print(f"{label:15s}: " + ("every replay equals the eager gradients" if first_bad is None else f"WRONG from replay {first_bad} on") + f" (worst relative difference {worst:.1e})", flush=True)
    return first_bad


def main():
    torch.manual_seed(0)
    print(f"DEBUG_CLR_GRAPH_PACKET_CAPTURE={os.environ.get('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '(unset: the runtime default, packet capture on)')}, "
          f"torch {torch.__version__}, HIP {torch.version.hip}, {B} x {N}")
    net = stock()
    params = list(net.parameters())
    hunt("torch-modules", lambda x: torch.autograd.grad(net(x).square().mean(), params), params)
    fp = flat_params(net)
    hunt("torch-function", lambda x: torch.autograd.grad(Chain.apply(x, *fp).square().mean(), fp), fp)
    if "--no-dfepe" in sys.argv:
        return
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    d = importlib.import_module("pytorch-deepfepe_amd")
    est = d.compat.ErrorEstimators.FusedErrorEstimator(WIDTHS[0]).to(DEV)
    d.synth.fill_params_deterministic(est, 3)
    ep = list(est.parameters())
    for label, use in (("dfepe-launch", False), ("dfepe-pass", True)):
        d.estimator.USE_PASS = use
        hunt(label, lambda x: torch.autograd.grad(est(x).square().mean(), ep), ep)


if __name__ == "__main__":
    main()
