import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = importlib.import_module("pytorch-deepfepe_amd")
t = importlib.import_module("test_captured_step_gpu")
DEV, depth, N = "cuda:0", 3, 100
net = d.compat.DeepFNet.DeepFNet(depth=depth, image_size=[376, 1241, 3], if_quality=False).to(DEV)
d.synth.fill_params_deterministic(net, 3)
fn = t._make_step(d, net, depth, False)
use_module = os.environ.get("PROBE_MODULE", "1") == "1"
step = d.compat.CapturedStep(fn, net if use_module else list(net.parameters()), warmup=2)
names = [n for n, _ in net.named_parameters()]
batches = [t._batch(d, 48, N, 100 + k) for k in range(2)]
PRE = os.environ.get("PROBE_PRE", "0") == "1"
pre = [t._eager(net, fn, b)[:2] for b in batches] if PRE else None
for rnd in range(4):
    for bi, b in enumerate(batches):
        if PRE:
            ref_loss, ref_g = pre[bi]
            ref_g2 = ref_g
        else:
            ref_loss, ref_g, _ = t._eager(net, fn, b)
            ref_loss2, ref_g2, _ = t._eager(net, fn, b)
        net.zero_grad(set_to_none=True)
        before = (step.n_eager, step.n_captures, step.n_replays)
        loss, aux = step(b)
        torch.cuda.synchronize()
        kind = "eager" if step.n_eager > before[0] else ("capture+replay" if step.n_captures > before[1] else "replay")
        rel = [float((p.grad - g).abs().max() / g.abs().max().clamp_min(1e-30)) for p, g in zip(net.parameters(), ref_g)]
        rel2 = [float((a - g).abs().max() / g.abs().max().clamp_min(1e-30)) for a, g in zip(ref_g2, ref_g)]
        worst = int(np.argmax(rel))
        if rnd == 2 and bi == 0:
            print("   per-parameter rel diff:", " ".join(f"{n.split('.')[0][:3]}.{n.split('.')[2]}.{n.split('.')[3][0]}={r:.1e}" for n, r in zip(names, rel)))
        print(f"rnd {rnd} batch {bi} {kind:15s} loss diff {abs(float(loss) - float(ref_loss)):.2e}  worst rel grad diff {max(rel):.2e} at {names[worst]}  "
              f"(eager vs eager: {max(rel2):.2e}); first param {rel[0]:.2e}", flush=True)
