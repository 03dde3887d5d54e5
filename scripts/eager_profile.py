"""Where the host time of the EAGER reference call sequence goes (1.75 ms per step = 35 us of Python + ctypes +
allocator per launch against 0.24 ms of GPU work): cProfile over N eager steps of pipeline.reference_call_sequence + backward at the
benchmark's size, fixed logits.   python scripts/eager_profile.py [steps] [pose_gt 0|1]"""
import cProfile
import importlib
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = importlib.import_module("pytorch-deepfepe_amd")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
gt_in = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
if len(sys.argv) > 3:
    d.compat.train_good_utils.LAZY_HOST_METRICS = bool(int(sys.argv[3]))
B, N, L = 4096, 100, 5
sc = d.pipeline.scene_to_device(d.synth.make_scene(B, N, seed=1000, outlier_ratio=0.2, noise_px=0.5, depth_layers=L), "cuda:0")
rows = [sc["logits_layers"][l].detach().clone().unsqueeze(1).requires_grad_(True) for l in range(L)]
net = d.pipeline.make_api_net(L, [376, 1241, 3], rows)
seed = {}


def body():
    loss, outs, losses, geo = d.pipeline.reference_call_sequence(net, sc, L, pose_gt_in_loss_params=gt_in)
    torch.autograd.grad(loss, rows, grad_outputs=seed.setdefault("s", torch.ones_like(loss)))


for _ in range(20):
    body()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    body()
torch.cuda.synchronize()
print(f"eager: {(time.perf_counter() - t0) * 1e3 / steps:.4f} ms per step (pose_gt_in_loss_params={gt_in}, lazy host metrics={d.compat.train_good_utils.LAZY_HOST_METRICS})")
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    body()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(40)
st.sort_stats("cumulative").print_stats(45)
