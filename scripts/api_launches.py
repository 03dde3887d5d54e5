#!/usr/bin/env python3
"""Kernel launches of ONE step of the reference's call sequence (pipeline.reference_call_sequence + backward), in launch order,
grouped by who issued them.  GPU box:  python scripts/api_launches.py [--gt] [--probe] [--batch 4096]"""
import argparse
import importlib
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
IMAGE_SIZE = [376, 1241, 3]

ap = argparse.ArgumentParser()
ap.add_argument("--gt", action="store_true", help="ground truth also in loss_params (fused tail)")
ap.add_argument("--probe", action="store_true", help="linear probe estimator (recurrent backward shape)")
ap.add_argument("--batch", type=int, default=4096)
args = ap.parse_args()
dfepe = importlib.import_module("pytorch-deepfepe_amd")
dev = torch.device("cuda:0")
L, N = 5, 100
d = dfepe.pipeline.scene_to_device(dfepe.synth.make_scene(args.batch, N, seed=1000, outlier_ratio=0.2, noise_px=0.5, depth_layers=L), dev)
rows = [d["logits_layers"][l].detach().clone().unsqueeze(1).requires_grad_(True) for l in range(L)]
net = dfepe.pipeline.make_api_net(L, IMAGE_SIZE, rows, recurrent_probe=args.probe)
seed = {}


def body():
    loss, outs, losses, geo = dfepe.pipeline.reference_call_sequence(net, d, L, pose_gt_in_loss_params=args.gt)
    torch.cuda.synchronize()  # marks the forward / backward boundary in the trace order below (timestamps)
    return torch.autograd.grad(loss, rows, grad_outputs=seed.setdefault("s", torch.ones_like(loss)))


for _ in range(3):
    body()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile

with profile(activities=[ProfilerActivity.CUDA]) as prof:
    body()
    torch.cuda.synchronize()
evs = [e for e in prof.events() if str(getattr(e, "device_type", "")).endswith("CUDA")]
evs.sort(key=lambda e: e.time_range.start)
ours = ("w8pt", "loss_tail", "loss_stats", "floss", "pose_", "geo_misc", "deepf_input", "row_dot")
n_ours = 0
for i, e in enumerate(evs):
    mine = any(t in e.name for t in ours)
    n_ours += mine
    print(f"{i:3d} {'HIP ' if mine else 'glue'} {e.time_range.elapsed_us():8.1f} us  {e.name[:150]}")
print(f"total {len(evs)}  library kernels {n_ours}  torch glue {len(evs) - n_ours}")
