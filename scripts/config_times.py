"""Timings for the BASELINE.json configs other than the bench line (HIP events, resident inputs)."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = importlib.import_module("pytorch-deepfepe_amd")
IMG = [376, 1241, 3]
def t(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
# C2: B=1024, N=100, one fit forward (+E)
sc = d.pipeline.scene_to_device(d.synth.make_scene(1024, 100, seed=0), "cuda:0")
w = torch.softmax(sc["logits_layers"][0], 1)
us = t(lambda: d.ops.w8pt_forward(sc["matches_xy_ori"], None, w, True, 1241., 376., 0.5, True, False))
print(f"C2  B=1024 N=100 single fit forward: {us:.1f} us  -> {1024/us:.1f} Mpairs/s")
# C4 per GPU: B=4096, 40% outliers, depth 5, qt loss fwd+bwd (eager; bench.py replays it from a hipGraph)
sc = d.pipeline.scene_to_device(d.synth.make_scene(4096, 100, seed=0, outlier_ratio=0.4), "cuda:0")
us = t(lambda: d.pipeline.hot_path_step(sc, IMG, 5, 0.02, qt=True), 20)
print(f"C4  B=4096/GPU N=100 40% outliers depth 5 qt fwd+bwd (eager launches): {us:.1f} us/step -> {4096/us:.2f} Mpairs/s")
# C5: B=4096 (1 GPU) and 512 (per GPU of 8), N=1000, depth 1 + cheirality
for B in (4096, 512):
    sc = d.pipeline.scene_to_device(d.synth.make_scene(B, 1000, seed=0, outlier_ratio=0.2), "cuda:0")
    w = torch.softmax(sc["logits_layers"][0], 1)
    T = torch.tensor([[2.0 / 1241, 0, -1.0], [0, 2.0 / 376, -1.0], [0, 0, 1.0]], device="cuda:0")
    TK = (T @ sc["Ks"]).contiguous()  # per-pair constant, formed once
    def c5():
        F, r, e, s, _ = d.ops.w8pt_forward(sc["matches_xy_ori"], None, w, True, 1241., 376., 0.5, True, False)
        E = d.ops.congruence(F, TK)  # E = K^T T^T F T K; the (1,1,0) projection is implied by the decomposition inside cheirality
        return d.ops.cheirality(E, sc["Ks"], sc["matches_xy_ori"], 50.0)
    us_all = t(c5, 10)
    us_fit = t(lambda: d.ops.w8pt_forward(sc["matches_xy_ori"], None, w, True, 1241., 376., 0.5, True, False), 10)
    E = sc["Ks"].transpose(1, 2) @ T.t() @ sc["F_gt"] @ T @ sc["Ks"]
    us_ch = t(lambda: d.ops.cheirality(sc["E_gt"], sc["Ks"], sc["matches_xy_ori"], 50.0), 10)
    print(f"C5  B={B} N=1000: fit {us_fit:.1f} us, cheirality {us_ch:.1f} us, fit + E-from-F + cheirality {us_all:.1f} us -> {B/us_all:.2f} Mpairs/s")
