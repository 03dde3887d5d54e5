"""Randomized sweep of the cheirality kernel (packed-fp32 eigen stage + one fp64 Rayleigh-quotient iteration per DLT) against
the oracle's fp64 SVD-based DLT: per-candidate in-front counts (as a multiset: the candidate order follows the SVD gauge) and
the selected pose, over outlier ratios, noise levels, depth thresholds and both launch shapes (one / four wavefronts per pair)."""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = importlib.import_module("pytorch-deepfepe_amd")
oracle = importlib.import_module("oracle.deepf_oracle")
t0 = time.time()
hist = {}
for B, N in ((2304, 150), (96, 1000), (64, 64)):
    for seed, (outl, noise, thr) in enumerate(((0.0, 0.0, 50.0), (0.2, 0.5, 50.0), (0.5, 2.0, 50.0), (0.7, 1.0, 20.0), (0.3, 0.5, 5.0))):
        sc = d.synth.make_scene(B, N, seed=900 + 13 * seed + N, outlier_ratio=outl, noise_px=noise)
        E = sc["E_gt"].float().cuda().contiguous()
        Rt, win, cnt = d.ops.cheirality(E, sc["Ks"].cuda(), sc["matches_xy_ori"].cuda(), thr)
        cnt = cnt.cpu().numpy(); Rt = Rt.cpu().numpy(); win = win.cpu().numpy()
        worst, bad_pose = 0, 0
        idx = np.linspace(0, B - 1, 48).astype(int)
        for b in idx:
            m = sc["matches_xy_ori"][b].double().numpy()
            Rt_o, win_o, counts_o = oracle.cheirality_select(E[b].cpu().double(), sc["Ks"][b].double().numpy(), m[:, :2], m[:, 2:], thr)
            diff = int(np.abs(np.sort(np.array(counts_o)) - np.sort(cnt[b])).max())
            worst = max(worst, diff)
            hist[diff] = hist.get(diff, 0) + 1
            top2 = np.sort(np.array(counts_o))[-2:]
            if top2[1] - top2[0] > 2 * diff + 2 and max(counts_o) > 0:  # an unambiguous winner: same pose
                if np.abs(Rt[b] - np.asarray(Rt_o)[:3, :4]).max() > 1e-4:
                    bad_pose += 1
        print(f"B={B} N={N} outl={outl} noise={noise} thr={thr}: max |sorted count difference| {worst}, unambiguous winners with another pose: {bad_pose}", flush=True)
print("histogram of the per-pair max count difference:", dict(sorted(hist.items())), f"{time.time() - t0:.0f} s")
