"""One-off randomized parity sweep (GPU box): forward fit against the fp64 oracle over many seeds, sizes and scene kinds,
cheirality counts against the oracle's DLT on a sample.  Prints the worst cases; the committed tests hold the tolerances."""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = importlib.import_module("pytorch-deepfepe_amd")
oracle = importlib.import_module("oracle.deepf_oracle")
IMG = [376, 1241, 3]
worst = {}
t0 = time.time()
for N, B in ((8, 512), (9, 512), (12, 512), (20, 1024), (100, 4096), (300, 512), (600, 256), (1000, 256), (2000, 64)):
    for seed, (outl, noise) in enumerate(((0.0, 0.0), (0.0, 0.5), (0.2, 0.5), (0.4, 0.5), (0.2, 2.0), (0.6, 1.0))):
        sc = d.synth.make_scene(B, N, seed=1000 * N + seed, outlier_ratio=outl, noise_px=noise)
        m = sc["matches_xy_ori"]
        w = torch.softmax(sc["logits_layers"][0] * (1.0 + seed), dim=1)
        F, res, epi = d.ops.w8pt_raw(m.cuda(), w.cuda(), IMG[1], IMG[0], clamp_at=0.5, want_epi=True)
        assert torch.isfinite(F).all() and torch.isfinite(res).all() and torch.isfinite(epi).all()
        p1, p2, _ = oracle.normalize_hw(m.double(), IMG)
        o_out, o_res, _ = oracle.fit_forward(p1, p2, w.double().unsqueeze(1))
        a = F.cpu().double().flatten(1); r = o_out.flatten(1)
        a = a / a.norm(dim=1, keepdim=True); r = r / r.norm(dim=1, keepdim=True)
        s = torch.sign((a * r).sum(1, keepdim=True)); s[s == 0] = 1
        err = (a * s - r).norm(dim=1)
        # conditioning of each problem: relative gap between the selected eigenvalue of X^T X and its neighbours
        h1, _ = oracle.hartley(p1); h2, _ = oracle.hartley(p2)
        pp = torch.cat((h2[:, :, 0:1] * h1, h2[:, :, 1:2] * h1, h1), 2)
        pp = pp / pp.norm(dim=2, keepdim=True).clamp_min(1e-12)
        X = pp * w.double().unsqueeze(2)
        ev = torch.linalg.eigvalsh(X.transpose(1, 2) @ X)
        ks = max(0, 9 - N)
        gap = (ev[:, ks + 1] - ev[:, ks]) if ks == 0 else torch.minimum(ev[:, ks + 1] - ev[:, ks], ev[:, ks] - ev[:, ks - 1])
        gap = gap / ev[:, -1]
        key = (N, outl, noise)
        ok12 = gap > 1e-12
        worst[key] = (err.max().item(), err.median().item(), int((err > 1e-5).sum()), int(((err > 1e-4) & ok12).sum()), int(ok12.sum()),
                      float((err[ok12] * gap[ok12]).max()) if ok12.any() else 0.0)
for k, v in worst.items():
    print(f"N={k[0]:5d} outl={k[1]:.1f} noise={k[2]:.1f}: |dF| max {v[0]:.2e} median {v[1]:.2e}  above 1e-5: {v[2]:4d} | of the {v[4]} pairs with eigen-gap > 1e-12 trace: above 1e-4: {v[3]}, max err*gap {v[5]:.1e}")
# cheirality against the oracle DLT
sc = d.synth.make_scene(64, 500, seed=77, outlier_ratio=0.3, noise_px=1.0)
Rt, win, cnt = d.ops.cheirality(sc["E_gt"].cuda(), sc["Ks"].cuda(), sc["matches_xy_ori"].cuda(), 50.0)
bad = 0
for b in range(64):
    _, w_o, c_o = oracle.cheirality_select(sc["E_gt"][b].double(), sc["Ks"][b].numpy(), sc["matches_xy_ori"][b, :, :2].double().numpy(),
                                           sc["matches_xy_ori"][b, :, 2:].double().numpy(), 50.0)
    # the SVD gauges differ, so the candidate order inside {R1,R2} x {t,-t} may differ: compare the multisets of counts
    if np.abs(np.sort(np.array(c_o)) - np.sort(cnt[b].cpu().numpy())).max() > 2:
        bad += 1
print("cheirality: pairs whose sorted candidate counts differ from the oracle's DLT by more than 2:", bad, "of 64")
print(f"{time.time() - t0:.1f} s")
