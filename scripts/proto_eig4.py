"""Prototype (numpy, vectorised over many matrices) of the cheap smallest-eigenvector routine for the 4x4 DLT normal
matrix: Householder tridiagonalisation -> Laguerre iteration from below on the tridiagonal characteristic recurrence
(monotone for real-rooted polynomials) -> null vector as an adjugate column (division-free twisted factorisation)
-> back-transformation.
Checked against numpy.linalg.eigh on DLT matrices of synthetic scenes (inliers and outliers)."""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def smallest_eigvec4(S, iters=12):
    S = S.copy()
    n = S.shape[0]
    # --- Householder 1 on x = S[1:4, 0]
    x = S[:, 1:4, 0].copy()
    nx = np.sqrt((x * x).sum(1))
    alpha = -np.copysign(nx, x[:, 0])
    v = x.copy(); v[:, 0] -= alpha
    vv = (v * v).sum(1)
    ok1 = vv > 0
    beta = np.where(ok1, 2.0 / np.where(ok1, vv, 1.0), 0.0)
    A = S[:, 1:4, 1:4]
    p = beta[:, None] * np.einsum("bij,bj->bi", A, v)
    Kc = 0.5 * beta * (p * v).sum(1)
    q = p - Kc[:, None] * v
    A2 = A - v[:, :, None] * q[:, None, :] - q[:, :, None] * v[:, None, :]
    d0 = S[:, 0, 0]; e0 = np.where(ok1, alpha, x[:, 0])
    # --- Householder 2 on y = A2[1:3, 0]  (2-vector)
    y = A2[:, 1:3, 0].copy()
    ny = np.sqrt((y * y).sum(1))
    alpha2 = -np.copysign(ny, y[:, 0])
    v2 = y.copy(); v2[:, 0] -= alpha2
    vv2 = (v2 * v2).sum(1)
    ok2 = vv2 > 0
    beta2 = np.where(ok2, 2.0 / np.where(ok2, vv2, 1.0), 0.0)
    Bm = A2[:, 1:3, 1:3]
    p2 = beta2[:, None] * np.einsum("bij,bj->bi", Bm, v2)
    K2 = 0.5 * beta2 * (p2 * v2).sum(1)
    q2 = p2 - K2[:, None] * v2
    B2 = Bm - v2[:, :, None] * q2[:, None, :] - q2[:, :, None] * v2[:, None, :]
    d1 = A2[:, 0, 0]; e1 = np.where(ok2, alpha2, y[:, 0])
    d2 = B2[:, 0, 0]; d3 = B2[:, 1, 1]; e2 = B2[:, 0, 1]
    d = np.stack((d0, d1, d2, d3), 1); e = np.stack((e0, e1, e2), 1)
    # --- Laguerre from below on p(lam) = det(T - lam I)
    scale = np.abs(d).max(1) + np.abs(e).max(1)
    lam = np.zeros(n)
    e2s = e * e
    done = np.zeros(n, bool)
    for it in range(iters):
        a = d - lam[:, None]
        p0 = np.ones(n); dp0 = np.zeros(n); ddp0 = np.zeros(n)
        p1 = a[:, 0]; dp1 = -np.ones(n); ddp1 = np.zeros(n)
        for k in range(1, 4):
            pk = a[:, k] * p1 - e2s[:, k - 1] * p0
            dpk = a[:, k] * dp1 - p1 - e2s[:, k - 1] * dp0
            ddpk = a[:, k] * ddp1 - 2 * dp1 - e2s[:, k - 1] * ddp0
            p0, dp0, ddp0, p1, dp1, ddp1 = p1, dp1, ddp1, pk, dpk, ddpk
        P, dP, ddP = p1, dp1, ddp1
        good = (P > 0) & ~done          # below the smallest root p > 0 (PSD); p <= 0: at / past the root by rounding
        G = np.where(good, dP / np.where(good, P, 1.0), 0.0)
        H = G * G - np.where(good, ddP / np.where(good, P, 1.0), 0.0)
        disc = np.maximum(3.0 * (4.0 * H - G * G), 0.0)
        den = G - np.sqrt(disc)          # G < 0 below the smallest root: the larger magnitude is G - sqrt
        step = np.where(good & (den < 0), -4.0 / np.where(den < 0, den, -1.0), 0.0)
        newlam = lam + step
        done = done | ~good | (step <= 1e-16 * scale) | (newlam == lam)
        lam = np.where(done, lam, newlam)
        lam = np.where(good & ~done, newlam, lam)
        if done.all():
            break
    nit = it + 1
    # --- null vector of T - lam: the column of adj(T - lam I) with the largest diagonal cofactor.  adj[i][j] =
    # (-1)^(i+j) P_i e_i..e_(j-1) Q_(j+1) (i <= j), P_i / Q_j the leading / trailing principal minors -- the vector a twisted
    # factorisation gives (its pivot gamma_r = det / (P_r Q_(r+1)) is smallest where that cofactor is largest), division-free.
    a = d - lam[:, None]
    P1 = a[:, 0]; P2 = a[:, 1] * a[:, 0] - e2s[:, 0]; P3 = a[:, 2] * P2 - e2s[:, 1] * P1
    Q3 = a[:, 3]; Q2 = a[:, 2] * a[:, 3] - e2s[:, 2]; Q1 = a[:, 1] * Q2 - e2s[:, 1] * Q3
    e0, e1, e2_ = e[:, 0], e[:, 1], e[:, 2]
    cols = np.stack([np.stack([Q1, -e0 * Q2, e0 * e1 * Q3, -e0 * e1 * e2_], 1),
                     np.stack([-e0 * Q2, P1 * Q2, -P1 * e1 * Q3, P1 * e1 * e2_], 1),
                     np.stack([e0 * e1 * Q3, -P1 * e1 * Q3, P2 * Q3, -P2 * e2_], 1),
                     np.stack([-e0 * e1 * e2_, P1 * e1 * e2_, -P2 * e2_, P3], 1)], 1)  # [n, r, component]
    r = np.abs(np.stack([Q1, P1 * Q2, P2 * Q3, P3], 1)).argmax(1)
    yv = cols[np.arange(n), r].copy()
    # --- back-transform x = H1 H2 y  (H2 acts on components 2,3; H1 on 1..3)
    t2 = beta2 * (v2[:, 0] * yv[:, 2] + v2[:, 1] * yv[:, 3])
    yv[:, 2] -= t2 * v2[:, 0]; yv[:, 3] -= t2 * v2[:, 1]
    t1 = beta * (v[:, 0] * yv[:, 1] + v[:, 1] * yv[:, 2] + v[:, 2] * yv[:, 3])
    yv[:, 1] -= t1 * v[:, 0]; yv[:, 2] -= t1 * v[:, 1]; yv[:, 3] -= t1 * v[:, 2]
    return yv / np.linalg.norm(yv, axis=1, keepdims=True), lam, nit


if __name__ == "__main__":
    d = importlib.import_module("pytorch-deepfepe_amd")
    oracle = importlib.import_module("oracle.deepf_oracle")
    sc = d.synth.make_scene(8, 1000, seed=0, outlier_ratio=0.25, noise_px=0.5)
    mats, meta = [], []
    for b in range(8):
        K = sc["Ks"][b].double().numpy()
        Rs, ts = oracle.get_M2s(sc["E_gt"][b].double())
        m = sc["matches_xy_ori"][b].double().numpy()
        P1 = K @ np.hstack((np.eye(3), np.zeros((3, 1))))
        for ri, R in enumerate(Rs):
            P2 = K @ np.hstack((R.numpy(), ts[0].numpy()))
            x1, y1, x2, y2 = m.T
            A = np.stack((x1[:, None] * P1[2] - P1[0], y1[:, None] * P1[2] - P1[1], x2[:, None] * P2[2] - P2[0], y2[:, None] * P2[2] - P2[1]), 1)
            mats.append(np.einsum("nki,nkj->nij", A, A))
            meta.append((R.numpy(), ts[0].numpy()))
    S = np.concatenate(mats)
    x, lam, nit = smallest_eigvec4(S)
    w, V = np.linalg.eigh(S)
    xr = V[:, :, 0]
    sgn = np.sign((x * xr).sum(1)); sgn[sgn == 0] = 1
    err = np.linalg.norm(x * sgn[:, None] - xr, axis=1)
    gap = (w[:, 1] - w[:, 0]) / w[:, 3]
    print("matrices", len(S), "Laguerre iterations used", nit)
    print("eigenvalue abs err / lam_max: max", (np.abs(lam - w[:, 0]) / w[:, 3]).max())
    print("eigenvector err percentiles 50/99/max:", np.percentile(err, [50, 99, 100]), " worst err*gap", (err * gap).max())
    # cheirality decisions
    flips = 0
    for k, (R, t) in enumerate(meta):
        for X, tag in ((x[k * 1000:(k + 1) * 1000], "ours"),):
            Xr = xr[k * 1000:(k + 1) * 1000]
            def dec(Xh):
                P = Xh[:, :3] / Xh[:, 3:4]
                z1 = P[:, 2]; z2 = (P @ R.T + t.ravel())[:, 2]
                return (z1 > 0) & (z1 < 50) & (z2 > 0) & (z2 < 50), (z1 < 0) & (z1 > -50) & (z2 < 0) & (z2 > -50)
            a1, a2 = dec(X); b1, b2 = dec(Xr)
            flips += int((a1 != b1).sum() + (a2 != b2).sum())
    print("cheirality decisions that differ from eigh:", flips, "of", 2 * len(S))
