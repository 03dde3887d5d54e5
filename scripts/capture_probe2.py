"""Bisect of the CapturedStep crash seen under pytest (hipStreamEndCapture segfault): variants of tests/test_captured_step_gpu.py, each
in a subprocess."""
import importlib, os, subprocess, sys
VARIANTS = ["static_inputs_only", "with_eager_reference", "with_realise", "with_host_gt", "full_test_pose_gt0", "full_test_pose_gt1", "gc_disabled"]
if len(sys.argv) == 1:
    for v in VARIANTS:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), v], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        tail = [l for l in r.stdout.strip().splitlines() if l.strip()][-2:]
        print(f"{v:24s} rc {r.returncode:4d}  {' | '.join(tail)[:200]}", flush=True)
    sys.exit(0)
import gc
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = importlib.import_module("pytorch-deepfepe_amd")
t = importlib.import_module("test_captured_step_gpu")
v = sys.argv[1]
DEV, depth, N = "cuda:0", 3, 100
if v.startswith("full_test"):
    t.test_captured_step_equals_the_eager_sequence_over_batches(d, v.endswith("1"))
    print("OK", v)
    sys.exit(0)
net = d.compat.DeepFNet.DeepFNet(depth=depth, image_size=[376, 1241, 3], if_quality=False).to(DEV)
d.synth.fill_params_deterministic(net, 3)
fn = t._make_step(d, net, depth, False)
step = d.compat.CapturedStep(fn, net if os.environ.get('PROBE_MODULE', '1') == '1' else net.parameters(), warmup=2)
if v == "gc_disabled":
    gc.disable()
batches = [t._batch(d, 48, N, 100 + k, host_gt=(v == "with_host_gt" and k == 1)) for k in range(3)]
for rnd in range(3):
    for b in batches:
        if v in ("with_eager_reference", "with_realise", "with_host_gt", "gc_disabled"):
            bd = {k: torch.as_tensor(x).to(DEV) for k, x in b.items()}
            ref = t._eager(net, fn, bd)
        net.zero_grad(set_to_none=True)
        loss, aux = step(b)
        torch.cuda.synchronize()
        if v in ("with_realise", "with_host_gt", "gc_disabled"):
            d.compat.CapturedStep.realise(aux)
print("OK", v, "captures", step.n_captures, "replays", step.n_replays, float(loss))
