"""Prototype of the round-2 eigen phase of w8pt_fwd/bwd (fp64, vectorised over pairs instead of lanes):
Householder tridiagonalisation of M = X^T X / trace, the (skip+1)-th smallest eigenvalue by 16-way multisection on the
division-free Sturm sequence, its eigenvector by twisted factorisation, back-transformation -- and, for the backward,
u = -(M - lam I)^+ g through the same tridiagonal form (two Thomas sweeps around the twist index).
Compared with numpy.linalg.eigh on the moment matrices of synthetic scenes (incl. peaked weights -> clusters, N < 9)."""
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
dfepe = importlib.import_module("pytorch-deepfepe_amd")
oracle = importlib.import_module("oracle.deepf_oracle")


def householder_tridiag(M):
    """M [B,9,9] symmetric -> d [B,9], e [B,8], V [B,7,9] (reflector vectors), beta [B,7]:  H_0..H_6, H_k = I - beta v v^T,
    T = H_6..H_0 M H_0..H_6."""
    A = M.copy()
    B = A.shape[0]
    d = np.zeros((B, 9)); e = np.zeros((B, 8)); V = np.zeros((B, 7, 9)); beta = np.zeros((B, 7))
    for k in range(7):
        x = A[:, :, k].copy()
        x[:, : k + 1] = 0.0
        x1 = x[:, k + 1]
        sig = (x[:, k + 2 :] ** 2).sum(1)
        nrm = np.sqrt(x1 * x1 + sig)
        alpha = np.where(x1 > 0, -nrm, nrm)
        v = x.copy()
        v[:, k + 1] = x1 - alpha
        vtv = sig + v[:, k + 1] ** 2
        ok = (sig > 0) & (vtv > 0)
        b = np.where(ok, 2.0 / np.where(ok, vtv, 1.0), 0.0)
        alpha = np.where(ok, alpha, x1)
        p = b[:, None] * np.einsum("bij,bj->bi", A, v)
        K = 0.5 * b * (v * p).sum(1)
        w = p - K[:, None] * v
        A = A - v[:, :, None] * w[:, None, :] - w[:, :, None] * v[:, None, :]
        d[:, k] = A[:, k, k]
        e[:, k] = alpha
        V[:, k] = v
        beta[:, k] = b
    d[:, 7] = A[:, 7, 7]; d[:, 8] = A[:, 8, 8]; e[:, 7] = A[:, 8, 7]
    return d, e, V, beta


def sturm_count(d, e2, x):
    """number of eigenvalues < x; division-free three-term recurrence with sign tracking.  d [B,9], e2 [B,8], x [B,L]."""
    pm2 = np.ones_like(x)
    pm1 = d[:, 0:1] - x
    cnt = (pm1 < 0).astype(np.int64)
    for k in range(1, 9):
        p = (d[:, k : k + 1] - x) * pm1 - e2[:, k - 1 : k] * pm2
        # a zero takes the sign opposite to its predecessor
        neg_prev = pm1 < 0
        neg = np.where(p == 0, ~neg_prev, p < 0)
        cnt += (neg != neg_prev)
        # keep the recurrence going with the substituted sign for exact zeros
        pm2 = pm1
        pm1 = np.where(p == 0, np.where(neg, -1e-300, 1e-300), p)
    return cnt


def multisection(d, e, kth, rounds=13):
    B = d.shape[0]
    e2 = e * e
    lo = np.full(B, -1e-3); hi = np.full(B, 1.0 + 1e-3)
    fr = (np.arange(16) + 1.0) / 17.0
    for _ in range(rounds):
        x = lo[:, None] + (hi - lo)[:, None] * fr[None]
        c = sturm_count(d, e2, x)
        m = (c <= kth[:, None]).sum(1)
        w = (hi - lo) / 17.0
        lo, hi = lo + w * m, lo + w * (m + 1)
    return 0.5 * (lo + hi)


def twisted_vector(d, e, lam):
    B = d.shape[0]
    tiny = 1e-290
    dp = np.zeros((B, 9)); dm = np.zeros((B, 9))
    dp[:, 0] = d[:, 0] - lam
    for k in range(8):
        q = np.where(np.abs(dp[:, k]) < tiny, -tiny, dp[:, k])
        dp[:, k] = q
        dp[:, k + 1] = d[:, k + 1] - lam - e[:, k] ** 2 / q
    dm[:, 8] = d[:, 8] - lam
    for k in range(7, -1, -1):
        q = np.where(np.abs(dm[:, k + 1]) < tiny, -tiny, dm[:, k + 1])
        dm[:, k + 1] = q
        dm[:, k] = d[:, k] - lam - e[:, k] ** 2 / q
    gam = dp + dm - (d - lam[:, None])
    r = np.abs(gam).argmin(1)
    z = np.zeros((B, 9))
    idx = np.arange(B)
    z[idx, r] = 1.0
    for k in range(7, -1, -1):
        q = np.where(np.abs(dp[:, k]) < tiny, -tiny, dp[:, k])
        z[:, k] = np.where(k < r, -e[:, k] / q * z[:, k + 1], z[:, k])
    for k in range(1, 9):
        q = np.where(np.abs(dm[:, k]) < tiny, -tiny, dm[:, k])
        z[:, k] = np.where(k > r, -e[:, k - 1] / q * z[:, k - 1], z[:, k])
    z /= np.linalg.norm(z, axis=1, keepdims=True)
    return z, r


def back_transform(V, beta, z):
    y = z.copy()
    for k in range(6, -1, -1):
        s = beta[:, k] * (V[:, k] * y).sum(1)
        y = y - s[:, None] * V[:, k]
    return y


def fwd_transform(V, beta, g):
    y = g.copy()
    for k in range(7):
        s = beta[:, k] * (V[:, k] * y).sum(1)
        y = y - s[:, None] * V[:, k]
    return y


def tri_pinv_apply(d, e, lam, z, r, g):
    """y = (T - lam I)^+ g for g _|_ z ... g is projected first.  Solve with y_r = 0: two independent SPD-ish tridiagonal
    blocks (above and below the twist index), then project out z."""
    B = d.shape[0]
    g = g - z * (z * g).sum(1, keepdims=True)
    a = d - lam[:, None]
    y = np.zeros((B, 9))
    # upper block 0..r-1: forward elimination from the top, back substitution from r-1 upwards
    cp = np.zeros((B, 9)); gp = np.zeros((B, 9))
    q = a[:, 0].copy(); cp[:, 0] = q; gp[:, 0] = g[:, 0]
    for k in range(1, 9):
        m = e[:, k - 1] / cp[:, k - 1]
        cp[:, k] = a[:, k] - m * e[:, k - 1]
        gp[:, k] = g[:, k] - m * gp[:, k - 1]
    cm = np.zeros((B, 9)); gm = np.zeros((B, 9))
    cm[:, 8] = a[:, 8]; gm[:, 8] = g[:, 8]
    for k in range(7, -1, -1):
        m = e[:, k] / cm[:, k + 1]
        cm[:, k] = a[:, k] - m * e[:, k]
        gm[:, k] = g[:, k] - m * gm[:, k + 1]
    for k in range(7, -1, -1):  # rows k < r: y_k = (gp_k - e_k y_{k+1}) / cp_k with y_r = 0
        yk = (gp[:, k] - e[:, k] * y[:, k + 1]) / cp[:, k]
        y[:, k] = np.where(k < r, yk, y[:, k])
    for k in range(1, 9):
        yk = (gm[:, k] - e[:, k - 1] * y[:, k - 1]) / cm[:, k]
        y[:, k] = np.where(k > r, yk, y[:, k])
    y = y - z * (z * y).sum(1, keepdims=True)
    return y


def moments(B, N, seed, outl, noise, scale):
    sc = dfepe.synth.make_scene(B, N, seed=seed, outlier_ratio=outl, noise_px=noise)
    m = sc["matches_xy_ori"].double()
    w = torch.softmax(sc["logits_layers"][0].double() * scale, dim=1)
    p1, p2, _ = oracle.normalize_hw(m, [376, 1241, 3])
    _, X, _, _ = oracle.fit_rows(p1, p2, w.unsqueeze(1))
    M = (X.transpose(1, 2) @ X).numpy()
    return M, X.numpy()


def run(B, N, seed, outl, noise, scale):
    M, X = moments(B, N, seed, outl, noise, scale)
    tr = np.trace(M, axis1=1, axis2=2)
    Mn = M / tr[:, None, None]
    d, e, V, beta = householder_tridiag(Mn)
    skip = 0 if N >= 9 else 9 - N
    kth = np.full(B, skip)
    lam = multisection(d, e, kth)
    z, r = twisted_vector(d, e, lam)
    f = back_transform(V, beta, z)
    ev, Q = np.linalg.eigh(Mn)
    fr = Q[:, :, skip]
    s = np.sign((f * fr).sum(1))
    err = np.linalg.norm(f * s[:, None] - fr, axis=1)
    gap = np.minimum(ev[:, skip + 1] - ev[:, skip], ev[:, skip] - ev[:, skip - 1] if skip > 0 else 1.0)
    lerr = np.abs(lam - ev[:, skip])
    # adjoint: u = sum_k q_k (q_k.g)/(lam_sel - lam_k)
    rng = np.random.default_rng(seed)
    g = rng.standard_normal((B, 9))
    den = ev[:, skip : skip + 1] - ev
    den[:, skip] = 1.0
    coef = np.einsum("bck,bc->bk", Q, g) / den
    coef[:, skip] = 0.0
    u_ref = np.einsum("bck,bk->bc", Q, coef)
    gt = fwd_transform(V, beta, g)
    y = -tri_pinv_apply(d, e, lam, z, r, gt)
    u = back_transform(V, beta, y)
    uerr = np.linalg.norm(u - u_ref, axis=1) / np.linalg.norm(u_ref, axis=1)
    well = gap > 1e-11
    print(f"B={B} N={N} outl={outl} noise={noise} scale={scale}: |lam err| max {lerr.max():.2e}; vec err max {err.max():.2e} "
          f"(well-separated {err[well].max():.2e}, min gap {gap.min():.1e}, err*gap max {(err * gap).max():.1e}); "
          f"adjoint rel err max {uerr[well].max():.2e} median {np.median(uerr):.1e}")


if __name__ == "__main__":
    run(1024, 100, 1, 0.2, 0.5, 1.0)
    run(1024, 100, 2, 0.4, 0.5, 1.0)
    run(1024, 100, 3, 0.0, 0.0, 1.0)
    run(1024, 100, 4, 0.2, 0.5, 4.0)
    run(512, 12, 5, 0.2, 0.5, 2.0)
    run(256, 8, 6, 0.2, 0.5, 1.0)
    run(256, 5, 7, 0.2, 0.5, 1.0)
    run(256, 1000, 8, 0.2, 0.5, 1.0)
