"""Prototype (numpy, float32 arithmetic) of an alternative to the systolic Jacobi of w8pt_fwd's phase 4a: Householder
tridiagonalisation + implicit QL with Wilkinson shift (EISPACK tred2/tql2 shape) on the 9x9 normal matrices of synthetic
scenes.  Reports accuracy against float64 eigh and the iteration / rotation counts that set the length of the dependent
chain on the GPU (lead 7).  CPU only."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
f32 = np.float32


def tred2(A):
    """Householder reduction of a symmetric matrix (float32) to tridiagonal form; returns d, e, Q with A = Q T Q^T."""
    n = A.shape[0]
    A = A.astype(f32).copy()
    Q = np.eye(n, dtype=f32)
    for k in range(n - 2):
        x = A[k + 1:, k].copy()
        alpha = -np.copysign(np.sqrt((x * x).sum(dtype=f32)), x[0]).astype(f32)
        v = x.copy(); v[0] -= alpha
        vv = (v * v).sum(dtype=f32)
        if vv == 0:
            continue
        beta = f32(2) / vv
        sub = A[k + 1:, k + 1:]
        p = beta * (sub @ v)
        K = f32(0.5) * beta * (p @ v)
        q = p - K * v
        A[k + 1:, k + 1:] = sub - np.outer(v, q) - np.outer(q, v)
        A[k + 1:, k] = 0; A[k + 1, k] = alpha
        A[k, k + 1:] = 0; A[k, k + 1] = alpha
        Q[:, k + 1:] -= np.outer(Q[:, k + 1:] @ v, beta * v)
    return np.diag(A).copy(), np.diag(A, 1).copy(), Q


def tql2(d, e, Z):
    """Implicit QL with Wilkinson shift, float32; returns eigenvalues, eigenvectors, (#iterations, #rotations)."""
    n = len(d)
    d = d.astype(f32).copy(); e = np.append(e.astype(f32), f32(0)); Z = Z.astype(f32).copy()
    iters = rots = 0
    eps = np.finfo(f32).eps
    for l in range(n):
        for _ in range(60):
            m = l
            while m < n - 1:
                if abs(e[m]) <= eps * (abs(d[m]) + abs(d[m + 1])):
                    break
                m += 1
            if m == l:
                break
            iters += 1
            g = (d[l + 1] - d[l]) / (f32(2) * e[l])
            r = np.hypot(g, f32(1)).astype(f32)
            g = d[m] - d[l] + e[l] / (g + np.copysign(r, g))
            s = c = f32(1); p = f32(0)
            for i in range(m - 1, l - 1, -1):
                f = s * e[i]; b = c * e[i]
                r = np.hypot(f, g).astype(f32)
                e[i + 1] = r
                if r == 0:
                    d[i + 1] -= p; e[m] = 0
                    break
                s = f / r; c = g / r
                g = d[i + 1] - p
                r = (d[i] - g) * s + f32(2) * c * b
                p = s * r
                d[i + 1] = g + p
                g = c * r - b
                zi1 = Z[:, i + 1].copy()
                Z[:, i + 1] = s * Z[:, i] + c * zi1
                Z[:, i] = c * Z[:, i] - s * zi1
                rots += 1
            else:
                d[l] -= p; e[l] = g; e[m] = 0
    return d, Z, iters, rots


if __name__ == "__main__":
    import torch
    dm = importlib.import_module("pytorch-deepfepe_amd")
    oracle = importlib.import_module("oracle.deepf_oracle")
    IMG = [376, 1241, 3]
    for label, scale, outl in (("bench-like logits", 1.0, 0.2), ("peaked logits x3", 3.0, 0.2)):
        sc = dm.synth.make_scene(300, 100, seed=3, outlier_ratio=outl, noise_px=0.5)
        w = torch.softmax(sc["logits_layers"][0] * scale, 1).double()
        p1, p2, _ = oracle.normalize_hw(sc["matches_xy_ori"].double(), IMG)
        h1, _ = oracle.hartley(p1); h2, _ = oracle.hartley(p2)
        rows = torch.cat((h2[:, :, 0:1] * h1, h2[:, :, 1:2] * h1, h1), 2)
        rows = rows / rows.norm(dim=2, keepdim=True).clamp_min(1e-12)
        X = rows * w.unsqueeze(2)
        M = (X.transpose(1, 2) @ X).numpy()
        M = M / np.trace(M, axis1=1, axis2=2)[:, None, None]
        it_l, rot_l, ev_err, res, orth = [], [], [], [], []
        for A in M:
            d, e, Q = tred2(A)
            lam, Z, it, ro = tql2(d, e, Q)
            w64 = np.linalg.eigvalsh(A)
            ev_err.append(np.abs(np.sort(lam.astype(np.float64)) - w64).max())
            res.append(np.abs(A @ Z.astype(np.float64) - Z.astype(np.float64) * lam.astype(np.float64)).max())
            orth.append(np.abs(Z.T.astype(np.float64) @ Z.astype(np.float64) - np.eye(9)).max())
            it_l.append(it); rot_l.append(ro)
        print(f"{label}: QL iterations mean {np.mean(it_l):.1f} max {max(it_l)}, rotations mean {np.mean(rot_l):.0f} max {max(rot_l)}; "
              f"|lam - eigh| max {max(ev_err):.1e} (unit trace), residual max {max(res):.1e}, |Z^T Z - I| max {max(orth):.1e}")
