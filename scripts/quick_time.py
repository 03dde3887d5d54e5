"""Scratch timing of the forward kernel (HIP events) — superseded by bench.py."""
import importlib, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = importlib.import_module("pytorch-deepfepe_amd")
for B, N in ((4096, 100), (1024, 100), (32768, 100), (4096, 1000), (512, 1000)):
    sc = d.synth.make_scene(B, N, seed=1, outlier_ratio=0.2) if B * N <= 4096 * 1000 else None
    m = sc["matches_xy_ori"].cuda(); w = torch.softmax(sc["logits_layers"][0], 1).cuda()
    for _ in range(5):
        d.ops.w8pt_forward(m, None, w, True, 1241., 376., 0.5, True, True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        d.ops.w8pt_forward(m, None, w, True, 1241., 376., 0.5, True, True)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / 50
    byts = B * (28 * N + 36)
    print(f"B={B} N={N}: {us:.1f} us/fit  {B/us:.2f} Mpairs/s  {byts/us/1e3:.1f} GB/s algorithmic")
