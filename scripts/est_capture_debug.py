"""Does the estimator's backward survive graph replays with changing inputs?  (weight gradients of the small layers read zero after the
first replay inside compat.CapturedStep)"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = importlib.import_module("pytorch-deepfepe_amd")
DEV = "cuda:0"
B, N = int(os.environ.get("DBG_B", "48")), 100
net = d.compat.DeepFNet.DeepFNet(depth=2, image_size=[376, 1241, 3], if_quality=False).to(DEV)
d.synth.fill_params_deterministic(net, 3)
est = net.input_weights
params = list(est.parameters())
names = [n for n, _ in est.named_parameters()]
xs = [torch.randn(B, 4, N, device=DEV) for _ in range(3)]
x_static = xs[0].clone()


def run(x):
    y = est(x).square().mean()
    return y, torch.autograd.grad(y, params)


refs = [tuple(g.clone() for g in run(x)[1]) for x in xs]
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    run(x_static)
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
if os.environ.get("DBG_DOT"):
    g.enable_debug_mode()
with torch.cuda.graph(g, stream=side):
    y, grads = run(x_static)
if os.environ.get("DBG_DOT"):
    try:
        g.debug_dump(os.environ["DBG_DOT"])
        print("dot written")
    except Exception as e:
        print("debug_dump failed:", repr(e)[:200])
for it in range(6):
    k = it % 3
    x_static.copy_(xs[k])
    g.replay(); torch.cuda.synchronize()
    rel = [float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)) for a, b in zip(grads, refs[k])]
    bad = [(n, f"{r:.1e}", f"|g|max {float(a.abs().max()):.2e}") for n, r, a in zip(names, rel, grads) if r > 1e-4]
    print(f"replay {it} input {k}: worst {max(rel):.2e}  bad: {bad}", flush=True)
