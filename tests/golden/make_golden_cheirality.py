#!/usr/bin/env python3
"""Golden vectors that pin the reference's OWN cheirality logic (utils_F._E_to_M_train, deepFEPE/dsac_tools/utils_F.py:679-763):
candidate order of _get_M2s (:478-498), the 0 < Z < depth_thres test in both cameras (:718-725), first arg-max (:730) and
utils_misc._inv_Rt of the winner.  The reference itself is run (imported unmodified, see make_golden.py); only the per-point
triangulation goes through the DLT stand-in for cv2.triangulatePoints (OpenCV is not in the image), so per-point depths are
"stubbed-cv2" while everything decided from them by the reference's code is pinned: per-candidate counts, winner, Rt_cam.

    python tests/golden/make_golden_cheirality.py      # rewrites tests/golden/cheirality.npz (build container only)
"""
import contextlib
import io
import os
import re
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (stubs, synth, quiet)


def run_case(utils_F, B, N, seed, outl, noise, e_noise, depth_thres):
    sc = mg.synth.make_scene(B, N, seed=seed, outlier_ratio=outl, noise_px=noise)
    g = torch.Generator().manual_seed(seed + 1)
    E = sc["E_gt"] / sc["E_gt"].flatten(1).norm(dim=1)[:, None, None]
    E = (E + e_noise * torch.randn(B, 3, 3, generator=g)).float()  # what the kernel sees: fp32
    m = sc["matches_xy_ori"].float()
    K = sc["Ks"].float()
    counts, winners, Rts = [], [], []
    for b in range(B):
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            _, _, Rt_cam = utils_F._E_to_M_train(E[b].double(), K[b].double().numpy(), m[b, :, :2].double().numpy(), m[b, :, 2:].double().numpy(),
                                                 depth_thres=depth_thres, show_debug=True, show_result=False)
        txt = buf.getvalue()
        c = re.findall(r"^\[(?:np\.int64\()?(\d+)\)?, (?:np\.int64\()?(\d+)\)?, (?:np\.int64\()?(\d+)\)?, (?:np\.int64\()?(\d+)\)?\]$", txt, flags=re.M)
        assert len(c) == 1, txt[-400:]
        counts.append([int(v) for v in c[0]])
        w = re.findall(r"The (\d+)_th \(0-based\) Rt meets", txt)
        winners.append(int(w[0]) if w else -1)
        Rts.append(mg.npy(Rt_cam) if Rt_cam is not None else np.full((3, 4), np.nan))
    return {"E": mg.npy(E), "K": mg.npy(K), "matches": mg.npy(m), "counts": np.array(counts, dtype=np.int32),
            "winner": np.array(winners, dtype=np.int32), "Rt_cam": np.stack(Rts), "depth_thres": np.array(depth_thres)}


def main():
    mg.install_stubs()
    with mg.quiet():
        import deepFEPE.dsac_tools.utils_F as utils_F
    out = {}
    cases = {
        # outliers, pixel noise, a perturbed E and a depth bound inside the scene (Z spans 5..35 m): every branch of the test decides
        "mixed": dict(B=12, N=150, seed=51, outl=0.3, noise=0.5, e_noise=2e-3, depth_thres=20.0),
        # config-5 shape: dense correspondences, the default bound
        "dense1000": dict(B=16, N=1000, seed=52, outl=0.2, noise=0.5, e_noise=1e-3, depth_thres=50.0),
        # hopeless E: most pairs keep few or no points in front of both cameras
        "garbage": dict(B=6, N=64, seed=53, outl=0.0, noise=0.0, e_noise=1.0, depth_thres=50.0),
    }
    for name, kw in cases.items():
        for k, v in run_case(utils_F, **kw).items():
            out[f"{name}_{k}"] = v
        print(name, "winners", out[f"{name}_winner"].tolist())
    np.savez_compressed(os.path.join(HERE, "cheirality.npz"), **out)
    with open(os.path.join(HERE, "MANIFEST.txt"), "a") as f:
        f.write(f"cheirality.npz: {len(out)} arrays, {os.path.getsize(os.path.join(HERE, 'cheirality.npz'))} bytes "
                "(make_golden_cheirality.py: the reference's _E_to_M_train with the DLT stand-in for cv2.triangulatePoints)\n")


if __name__ == "__main__":
    main()
