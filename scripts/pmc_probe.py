"""Launch the hot kernels a few times (for rocprofv3 --pmc passes): 6 fused training steps, then 4 stand-alone
w8pt_fwd launches in exactly the configuration bench.py's roofline probe times (weights in, epi + save out)."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = importlib.import_module("pytorch-deepfepe_amd")
B, N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, int(sys.argv[2]) if len(sys.argv) > 2 else 100
sc = d.pipeline.scene_to_device(d.synth.make_scene(B, N, seed=1, outlier_ratio=0.2), "cuda:0")
for _ in range(6):
    d.pipeline.hot_path_step(sc, [376, 1241, 3], 5, 0.02, qt=True)
torch.cuda.synchronize()
w = torch.softmax(sc["logits_layers"][0], dim=1).contiguous()
for _ in range(4):
    d.ops.w8pt_forward(sc["matches_xy_ori"], None, w, True, 1241.0, 376.0, 0.5, True, True)
torch.cuda.synchronize()
