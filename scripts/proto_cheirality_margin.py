"""Prototype (numpy) for an ADAPTIVE cheirality DLT: decide the depth tests from the packed-fp32 eigenvector alone and send a group
of 64 correspondences through the fp64 stage (fp64 normal matrices + Rayleigh-quotient iteration) only when one of its lanes has a
test within a margin of its bound.  Measures, against numpy.linalg.eigh decisions: how many decisions of the fp32 stage differ, how
many of those the margin rule flags, and which fraction of correspondences / of 64-groups it flags.
    python scripts/proto_cheirality_margin.py"""
import importlib
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
spec = importlib.util.spec_from_file_location("proto_eig4", os.path.join(HERE, "proto_eig4.py"))
pe = importlib.util.module_from_spec(spec)
spec.loader.exec_module(pe)
d = importlib.import_module("pytorch-deepfepe_amd")
oracle = importlib.import_module("oracle.deepf_oracle")


def tests(X, R, t, thr, margin=None, d4=None):
    """decision pair (pos, neg) like the kernel's division-free tests; with `margin`: also the ambiguity flag."""
    wq, z1n = X[:, 3], X[:, 2]
    z2n = X[:, :3] @ R[2] + t[2] * wq
    aw = thr * np.abs(wq)
    inr = (np.abs(z1n) < aw) & (np.abs(z2n) < aw) & (wq != 0)
    s1p, s2p = (z1n > 0) == (wq > 0), (z2n > 0) == (wq > 0)
    nz = (z1n != 0) & (z2n != 0)
    pos, neg = inr & nz & s1p & s2p, inr & nz & ~s1p & ~s2p
    if margin is None:
        return pos, neg
    n = np.abs(X).max(1)  # scale of the (unnormalised) vector
    if d4 is not None:  # a-posteriori bound of the fp32 stage: |x~ - x| <= 3e-8 / prod_j (lam_j - lam_4) (measured); margin = safety factor
        dlt = np.maximum(margin * 4e-8 / np.maximum(d4, 1e-30), 1e-6) * np.linalg.norm(X, axis=1)
    else:
        dlt = margin * n
    amb = (np.abs(wq) < dlt) | (np.abs(z1n) < dlt) | (np.abs(z2n) < 2 * dlt) | (np.abs(np.abs(z1n) - aw) < (1 + thr) * dlt) | \
          (np.abs(np.abs(z2n) - aw) < (2 + thr) * dlt)
    return pos, neg, amb


def main():
    for outl, noise, thr in ((0.2, 0.5, 50.0), (0.4, 0.5, 50.0), (0.5, 2.0, 50.0), (0.0, 0.0, 50.0), (0.3, 0.5, 5.0), (0.7, 1.0, 20.0)):
        sc = d.synth.make_scene(12, 1000, seed=3, outlier_ratio=outl, noise_px=noise)
        tot = dict(n=0, diff=0)
        stats = {m: dict(flag=0, groups=0, ngroups=0, missed=0) for m in (1e-4, 3e-4, 1e-3, "gap x4", "gap x16")}
        for b in range(12):
            K = sc["Ks"][b].double().numpy()
            Rs, ts = oracle.get_M2s(sc["E_gt"][b].double())
            m = sc["matches_xy_ori"][b].double().numpy()
            P1 = K @ np.hstack((np.eye(3), np.zeros((3, 1))))
            x1, y1, x2, y2 = m.T
            for R in Rs:
                Rn, tn = R.numpy(), ts[0].numpy().ravel()
                P2 = K @ np.hstack((Rn, tn[:, None]))
                A = np.stack((x1[:, None] * P1[2] - P1[0], y1[:, None] * P1[2] - P1[1], x2[:, None] * P2[2] - P2[0], y2[:, None] * P2[2] - P2[1]), 1)
                S64 = np.einsum("nki,nkj->nij", A, A)
                w_, V_ = np.linalg.eigh(S64 / np.trace(S64, axis1=1, axis2=2)[:, None, None])
                xr = V_[:, :, 0]
                d4 = (w_[:, 1] - w_[:, 0]) * (w_[:, 2] - w_[:, 0]) * (w_[:, 3] - w_[:, 0])
                # the fast path's matrix: A rounded to fp32, S = A^T A accumulated in fp32, unit trace in fp32
                A32 = A.astype(np.float32)
                S32 = np.einsum("nki,nkj->nij", A32, A32).astype(np.float32)
                S32 = (S32 / np.trace(S32, axis1=1, axis2=2)[:, None, None]).astype(np.float32)
                with np.errstate(all="ignore"):
                    x0 = np.nan_to_num(pe.smallest_eigvec4(S32)[0].astype(np.float64))
                rp, rn = tests(xr, Rn, tn, thr)
                tot["n"] += 2 * len(A)
                for mg, st in stats.items():
                    if isinstance(mg, str):
                        p, n_, amb = tests(x0.astype(np.float32).astype(np.float64), Rn, tn, thr, float(mg.split("x")[1]), d4)
                    else:
                        p, n_, amb = tests(x0.astype(np.float32).astype(np.float64), Rn, tn, thr, mg)
                    differ = (p != rp) | (n_ != rn)
                    if mg == 1e-4:
                        tot["diff"] += int(differ.sum())
                    st["flag"] += int(amb.sum())
                    st["missed"] += int((differ & ~amb).sum())
                    g = amb[: len(amb) // 64 * 64].reshape(-1, 64).any(1)
                    st["groups"] += int(g.sum()); st["ngroups"] += len(g)
        print(f"outl {outl} noise {noise} thr {thr}: fp32-only decisions differing from eigh: {tot['diff']} of {tot['n']}")
        for mg, st in stats.items():
            print(f"    margin {mg}: flagged {st['flag'] / (tot['n'] / 2):.4%} of correspondences, {st['groups'] / st['ngroups']:.1%} of 64-groups; "
                  f"differing decisions NOT flagged: {st['missed']}")


if __name__ == "__main__":
    main()
