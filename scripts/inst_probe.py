"""One launch of w8pt_fwd per diagnostic variant, in a fixed order, for a rocprofv3 --pmc pass (dynamic instruction
counts per phase): 0 normal | 1: no Jacobi, no polish | 2: 4 forced sweeps, no polish | 3: 5 forced sweeps, no polish |
4: 4 forced sweeps + polish."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = importlib.import_module("pytorch-deepfepe_amd")
sc = d.synth.make_scene(4096, 100, seed=1, outlier_ratio=0.2)
m = sc["matches_xy_ori"].cuda(); w = torch.softmax(sc["logits_layers"][0], 1).cuda()
for diag in (0, 1, 5, 6, 0x105):
    d.ops.w8pt_forward(m, None, w, True, 1241., 376., 0.5, True, True, diag=diag)
    torch.cuda.synchronize()
F, r, e, sv, _ = d.ops.w8pt_forward(m, None, w, True, 1241., 376., 0.5, True, True)
print("sweeps mean", sv[:, 119].mean().item(), "refine mean", sv[:, 120].mean().item(), torch.bincount(sv[:, 120].long()).tolist())
