import importlib, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = importlib.import_module("pytorch-deepfepe_amd")
B, N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, int(sys.argv[2]) if len(sys.argv) > 2 else 100
sc = d.synth.make_scene(B, N, seed=1, outlier_ratio=0.2)
m = sc["matches_xy_ori"].cuda(); w = torch.softmax(sc["logits_layers"][0], 1).cuda()
for _ in range(3):
    F, r, e, sv, _ = d.ops.w8pt_forward(m, None, w, True, 1241., 376., 0.5, True, True)
torch.cuda.synchronize()
ph = sv[:, 121:128]
names = ["0 load+sums", "1 hartley", "2 accumulate", "3 reduce-scatter", "4a jacobi", "4b select+polish", "5 rank2+denorm"]
tot = ph.sum(1)
print(f"B={B} N={N}: per-wave cycles (mean / max over pairs); sum of phases mean {tot.mean():.0f} max {tot.max():.0f}")
for k, n in enumerate(names):
    print(f"  phase {n:18s} mean {ph[:, k].mean():8.0f}  max {ph[:, k].max():8.0f}  ({100*ph[:, k].mean()/tot.mean():.1f}%)")
import numpy as np
t = tot.cpu().numpy()
print("percentiles of per-wave total cycles:", {p: int(np.percentile(t, p)) for p in (1, 10, 50, 90, 99, 99.9, 100)})
jac = ph[:, 4].cpu().numpy(); sw = sv[:, 119].cpu().numpy()
for s_ in np.unique(sw):
    print(f"  sweeps={int(s_)}: n={int((sw == s_).sum())} jacobi cycles mean {jac[sw == s_].mean():.0f} total mean {t[sw == s_].mean():.0f} max {t[sw == s_].max():.0f}")
