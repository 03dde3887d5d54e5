"""One-off randomized sweep of the whole training step (fwd + bwd) against the fp64 oracle."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = importlib.import_module("pytorch-deepfepe_amd")
oracle = importlib.import_module("oracle.deepf_oracle")
IMG = [376, 1241, 3]
t0 = time.time()
for N, B, depth in ((100, 384, 5), (20, 256, 3), (300, 128, 2)):
    for seed, (outl, noise, scale) in enumerate(((0.0, 0.0, 1.0), (0.2, 0.5, 1.0), (0.4, 0.5, 1.0), (0.2, 0.5, 2.5), (0.5, 2.0, 1.0), (0.1, 0.2, 4.0))):
        sc = d.synth.make_scene(B, N, seed=7000 + 10 * N + seed, outlier_ratio=outl, noise_px=noise, depth_layers=depth)
        sc["logits_layers"] = sc["logits_layers"] * scale
        ours = d.pipeline.hot_path_step(d.pipeline.scene_to_device(sc, "cuda:0"), IMG, depth, 0.02, qt=True)
        ref = oracle.hot_path_step({k: v.double() for k, v in sc.items()}, IMG, depth, 0.02, qt=True, mode="batched")
        g, gr = ours["grad_logits"].cpu().double(), ref["grad_logits"]
        finite = bool(torch.isfinite(ours["grad_logits"]).all()) and bool(torch.isfinite(ours["loss"]))
        # per (layer, pair) relative gradient error
        num = (g - gr).flatten(2).norm(dim=2); den = gr.flatten(2).norm(dim=2).clamp_min(1e-30)
        rel = (num / den).flatten()
        tot = (g - gr).norm() / gr.norm()
        dl = abs(ours["loss"].item() - ref["loss"].item())
        print(f"N={N} outl={outl} noise={noise} scale={scale}: finite {finite} |dloss| {dl:.1e}  grad rel err total {tot:.1e}  per-pair median {rel.median():.1e} p99 {rel.quantile(0.99):.1e} max {rel.max():.1e}", flush=True)
print(f"{time.time()-t0:.1f} s")
