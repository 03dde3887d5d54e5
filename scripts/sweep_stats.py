import importlib, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = importlib.import_module("pytorch-deepfepe_amd")
for outl, noise in ((0.2, 0.5), (0.0, 0.5), (0.0, 0.0), (0.4, 0.5)):
    sc = d.synth.make_scene(4096, 100, seed=1, outlier_ratio=outl, noise_px=noise)
    m = sc["matches_xy_ori"].cuda(); w = torch.softmax(sc["logits_layers"][0], 1).cuda()
    F, r, e, sv, _ = d.ops.w8pt_forward(m, None, w, True, 1241., 376., 0.5, True, True)
    sw, rf = sv[:, 119], sv[:, 120]
    print(f"outl {outl} noise {noise}: sweeps mean {sw.mean():.2f} max {sw.max():.0f} hist {torch.bincount(sw.long()).tolist()} | refine mean {rf.mean():.2f} max {rf.max():.0f} hist {torch.bincount(rf.long()).tolist()}")
