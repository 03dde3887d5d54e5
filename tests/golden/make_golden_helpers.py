#!/usr/bin/env python3
"""Golden vectors for the small helpers of the mirrored dsac_tools modules, from the reference itself:
utils_F._E_F_from_Rt / E_F_from_Rt_np / E_to_F_np (deepFEPE/dsac_tools/utils_F.py:820-846,471-476) and
utils_geo.R_to_q_np / q_to_R_np / _rot_angle_error / vectors_angle (utils_geo.py:88-117,137-147,158-163,184-190).

    python tests/golden/make_golden_helpers.py      # rewrites tests/golden/helpers.npz (build container only)
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
        import deepFEPE.dsac_tools.utils_F as utils_F
        import deepFEPE.dsac_tools.utils_geo as utils_geo
    out = {}
    B = 12
    sc = mg.synth.make_scene(B, 10, seed=91, dtype=torch.float64)
    delta = sc["delta_Rtijs_4_4"].numpy()
    R, t, K = delta[:, :3, :3].copy(), delta[:, :3, 3:4].copy(), sc["Ks"].numpy()
    # rotations that reach all four branches of the trace method: the scene's small rotations plus half-turn-ish ones
    g = np.random.RandomState(5)
    big = []
    for ax in range(3):
        for ang in (3.0, 3.1):
            w = np.zeros(3); w[ax] = ang
            w += 0.05 * g.randn(3)
            th = np.linalg.norm(w); k = w / th
            Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
            big.append(np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx)
    Rs = np.concatenate((R, np.stack(big)))
    out.update({"R": Rs, "t": t, "K": K})
    EF = [utils_F.E_F_from_Rt_np(R[b], t[b], K[b]) for b in range(B)]
    out["E_np"], out["F_np"] = np.stack([e for e, _ in EF]), np.stack([f for _, f in EF])
    with mg.quiet():
        EFt = [utils_F._E_F_from_Rt(R[b], t[b], K[b]) for b in range(B)]
        Eb, Fb = utils_F._E_F_from_Rt(torch.from_numpy(R), torch.from_numpy(t), torch.from_numpy(K), tensor_input=True)
    out["E_th"], out["F_th"] = np.stack([mg.npy(e) for e, _ in EFt]), np.stack([mg.npy(f) for _, f in EFt])
    out["E_th_batch"], out["F_th_batch"] = mg.npy(Eb), mg.npy(Fb)
    out["E_to_F_np"] = np.stack([utils_F.E_to_F_np(out["E_np"][b], K[b]) for b in range(B)])
    qs = np.stack([utils_geo.R_to_q_np(Rs[i]) for i in range(len(Rs))])
    out["q"] = qs
    out["q_branch"] = np.array([(0 if Rs[i].T[2, 2] >= 0 and not Rs[i].T[0, 0] < -Rs[i].T[1, 1] else 1) for i in range(len(Rs))])
    out["R_from_q"] = np.stack([utils_geo.q_to_R_np(qs[i].astype(np.float64) * (1.0 + 0.3 * i)) for i in range(len(Rs))])  # unnormalised input
    with mg.quiet():
        out["rot_angle"] = np.array([float(utils_geo._rot_angle_error(torch.from_numpy(Rs[i]), torch.from_numpy(Rs[(i + 1) % len(Rs)]))) for i in range(len(Rs))])
    v1, v2 = g.randn(20, 3), g.randn(20, 3)
    out.update({"v1": v1, "v2": v2, "vectors_angle": utils_geo.vectors_angle(v1, v2)})
    np.savez_compressed(os.path.join(HERE, "helpers.npz"), **out)
    lines = [l for l in open(os.path.join(HERE, "MANIFEST.txt")).read().splitlines() if not l.startswith("helpers.npz")]
    lines.append(f"helpers.npz: {len(out)} arrays, {os.path.getsize(os.path.join(HERE, 'helpers.npz'))} bytes (make_golden_helpers.py: "
                 "_E_F_from_Rt / E_F_from_Rt_np / E_to_F_np, R_to_q_np / q_to_R_np / _rot_angle_error / vectors_angle)")
    open(os.path.join(HERE, "MANIFEST.txt"), "w").write("\n".join(lines) + "\n")
    print({k: np.asarray(v).shape for k, v in out.items()})


if __name__ == "__main__":
    main()
