#!/usr/bin/env python3
"""Golden vectors for the last two corners of the mirrored call surface, from the reference itself:

* Fit(normalize_SVD=False) (deepFEPE/models/DeepFNet.py:124,211-214): rows of X are w_i p_i, not w_i p_i / |p_i|; forward
  (out, residual) in fp32 and fp64, and d/d(weights) of a fixed linear functional of both outputs by the reference's autograd;
* utils_F._E_from_XY / _F_from_XY with a DENSE W [N,N] left-multiplying the design matrix (utils_F.py:129-130,245-246).

    python tests/golden/make_golden_surface.py      # rewrites tests/golden/surface.npz (build container only)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402


def main():
    mg.install_stubs()
    with mg.quiet():
        from deepFEPE.models.DeepFNet import Fit, NormalizeAndExpand_HW
        import deepFEPE.dsac_tools.utils_F as utils_F
    out = {}
    B, N = 6, 100
    sc = mg.synth.make_scene(B, N, seed=71, outlier_ratio=0.2, noise_px=0.5, dtype=torch.float64)
    g = torch.Generator().manual_seed(72)
    GF = torch.randn(B, 3, 3, generator=g, dtype=torch.float64)
    GR = torch.randn(B, N, generator=g, dtype=torch.float64)
    w64 = torch.softmax(sc["logits_layers"][0], dim=1).unsqueeze(1)
    out.update({"nosvdnorm_matches": mg.npy(sc["matches_xy_ori"]), "nosvdnorm_weights": mg.npy(w64), "nosvdnorm_GF": mg.npy(GF), "nosvdnorm_GR": mg.npy(GR)})
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        with mg.quiet():
            norm = NormalizeAndExpand_HW(mg.IMAGE_SIZE, is_cuda=False)
            norm.ones_b = norm.ones_b.to(dt)
            p1, p2, _, _ = norm(sc["matches_xy_ori"].to(dt))
            pts1, pts2 = p1.permute(0, 2, 1), p2.permute(0, 2, 1)
            fit = Fit(is_cuda=False, normalize_SVD=False)
            for a in ("ones_b", "zero_b", "T_b", "mask"):
                setattr(fit, a, getattr(fit, a).to(dt))
            w = w64.to(dt).clone().requires_grad_(True)
            o, r = fit(pts1, pts2, w)
            # sign-invariant functional: both outputs flip with the SVD gauge, so each is multiplied by the sign it has against fixed probes
            s = torch.sign((o.detach() * GF.to(dt)).flatten(1).sum(1))
            ((s[:, None, None] * o * GF.to(dt)).sum() + (s[:, None] * r * GR.to(dt)).sum()).backward()
        out[f"nosvdnorm_out_{tag}"] = mg.npy(o)
        out[f"nosvdnorm_residual_{tag}"] = mg.npy(r)
        out[f"nosvdnorm_grad_w_{tag}"] = mg.npy(w.grad)
    # ---- dense W
    N2 = 40
    sc = mg.synth.make_scene(4, N2, seed=73, outlier_ratio=0.1, noise_px=0.5, dtype=torch.float64)
    g = torch.Generator().manual_seed(74)
    Ws = torch.stack([torch.diag(0.5 + torch.rand(N2, generator=g, dtype=torch.float64)) + 0.05 * torch.randn(N2, N2, generator=g, dtype=torch.float64)
                      for _ in range(4)])
    Es, Fs, Fn = [], [], []
    for b in range(4):
        X, Y, K = sc["matches_xy_ori"][b, :, :2], sc["matches_xy_ori"][b, :, 2:], sc["Ks"][b]
        with mg.quiet():
            Es.append(utils_F._E_from_XY(X, Y, K, W=Ws[b]))
            Fs.append(utils_F._F_from_XY(X, Y, W=Ws[b]))
            Fn.append(utils_F._F_from_XY(X, Y, W=Ws[b], normalize=False))
    out.update({"densew_matches": mg.npy(sc["matches_xy_ori"]), "densew_K": mg.npy(sc["Ks"]), "densew_W": mg.npy(Ws), "densew_E": mg.npy(Es),
                "densew_F": mg.npy(Fs), "densew_F_nonorm": mg.npy(Fn)})
    np.savez_compressed(os.path.join(HERE, "surface.npz"), **out)
    with open(os.path.join(HERE, "MANIFEST.txt"), "a") as f:
        f.write(f"surface.npz: {len(out)} arrays, {os.path.getsize(os.path.join(HERE, 'surface.npz'))} bytes (make_golden_surface.py: "
                "Fit(normalize_SVD=False) forward + autograd, _E_from_XY / _F_from_XY with a dense W)\n")
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
