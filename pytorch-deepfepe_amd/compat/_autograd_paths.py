"""Differentiable evaluation of the legacy dsac_tools helpers.

The reference's _sym_epi_dist, _sampson_dist, _epi_distance, compute_epi_residual, _get_M2s, _R_to_q, _F_from_XY and _E_from_XY
are plain torch code and therefore differentiable in every argument (deepFEPE/dsac_tools/utils_F.py:104-155,223-275,291-361,
400-413,478-498; utils_geo.py:58-86); its legacy callers use that (train_good_utils.py:55-61, dsac_tools/dsac.py:138-176).  The
mirrors in compat.utils_F / compat.utils_geo are raw kernel launches with no adjoint (the hot path has its own fused adjoints:
ops.w8pt, ops.floss, ops.pose_errors).  When -- and only when -- one of their inputs requires grad, the mirrors evaluate the same
formulas here instead, as torch expressions on the tensors' own (GPU) device, so that autograd gives the caller the reference's
gradients; without a gradient request nothing in this file runs.  Each function restates the published formula, batched over
leading dimensions; the GPU tests hold the values to the kernels' and the gradients to float64 autograd of the oracle."""
import torch


def wants_grad(*tensors) -> bool:
    return torch.is_grad_enabled() and any(torch.is_tensor(t) and t.requires_grad for t in tensors)


def _h(P, if_homo):
    return P if if_homo else torch.cat((P, torch.ones_like(P[..., :1])), -1)


def _epipolar(F, X, Y, if_homo):
    """(y^T F x, F x, F^T y) of every correspondence; F [...,3,3], X, Y [...,N,2|3]."""
    x, y = _h(X, if_homo), _h(Y, if_homo)
    Fx = x @ F.transpose(-1, -2)
    Fty = y @ F
    return (y * Fx).sum(-1), Fx, Fty


def sym_epi_dist(F, X, Y, if_homo=False, clamp_at=None, eps=0.0):
    """(y^T F x)^2 (1 / |(F x)_{12}|^2 + 1 / |(F^T y)_{12}|^2)   (utils_F.py:310-339; eps = 1e-10 in the batched form only)."""
    s, a, b = _epipolar(F, X, Y, if_homo)
    d = s.square() * (1.0 / (a[..., :2].square().sum(-1) + eps) + 1.0 / (b[..., :2].square().sum(-1) + eps))
    return d if clamp_at is None else d.clamp(max=clamp_at)


def sampson_dist(F, X, Y, if_homo=False):
    """(y^T F x)^2 / (|(F x)_{12}|^2 + |(F^T y)_{12}|^2)   (utils_F.py:291-308)."""
    s, a, b = _epipolar(F, X, Y, if_homo)
    return s.square() / (a[..., :2].square().sum(-1) + b[..., :2].square().sum(-1))


def epi_distance(F, X, Y, if_homo=False):
    """Point-to-epipolar-line distances in both images: ((d1 + d2) / 2, d1, d2)   (utils_F.py:341-361)."""
    s, a, b = _epipolar(F, X, Y, if_homo)
    d1 = s.abs() / a[..., :2].norm(dim=-1)
    d2 = s.abs() / b[..., :2].norm(dim=-1)
    return (d1 + d2) / 2.0, d1, d2


def epi_residual(pts1, pts2, F, clamp_at):
    """|x2^T F x1| (1 / (|l1_{12}| + 1e-6) + 1 / (|l2_{12}| + 1e-6)), clamped   (compute_epi_residual, utils_F.py:400-413);
    pts [B,N,3] homogeneous."""
    l1 = pts2 @ F                      # rows x2^T F
    l2 = pts1 @ F.transpose(-1, -2)    # rows (F x1)^T
    dd = (pts1 * l1).sum(-1)
    d = dd.abs() * (1.0 / (l1[..., :2].norm(dim=-1) + 1e-6) + 1.0 / (l2[..., :2].norm(dim=-1) + 1e-6))
    return d.clamp(max=clamp_at)


def decompose_essential(E):
    """The two rotations U W V^T, U W^T V^T (W negated when det < 0) and t = u3 / |u3| of E [3,3]   (utils_F.py:478-498);
    torch.linalg.svd fixes the gauge, as torch.svd does in the reference."""
    U, _, Vh = torch.linalg.svd(E)
    W = E.new_tensor([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
    if torch.det(U @ W @ Vh) < 0:
        W = -W
    t = U[:, 2:3] / U[:, 2:3].norm()
    return U @ W @ Vh, U @ W.t() @ Vh, t


def rot_to_quat(R):
    """Unit quaternion (w,x,y,z), w >= 0, of rotations R [...,3,3]: Shepperd's method with the reference's pivot rule
    (utils_geo.py:58-86), as one gather over the four candidate rows."""
    d0, d1, d2 = R[..., 0, 0], R[..., 1, 1], R[..., 2, 2]
    ax, ay, az = R[..., 2, 1] - R[..., 1, 2], R[..., 0, 2] - R[..., 2, 0], R[..., 1, 0] - R[..., 0, 1]
    sxy, syz, szx = R[..., 0, 1] + R[..., 1, 0], R[..., 1, 2] + R[..., 2, 1], R[..., 2, 0] + R[..., 0, 2]
    rows = torch.stack((torch.stack((1 + d0 + d1 + d2, ax, ay, az), -1), torch.stack((ax, 1 + d0 - d1 - d2, sxy, szx), -1),
                        torch.stack((ay, sxy, 1 - d0 + d1 - d2, syz), -1), torch.stack((az, szx, syz, 1 - d0 - d1 + d2), -1)), -2)
    k = torch.where(d2 < 0, torch.where(d0 > d1, 1, 2), torch.where(d0 < -d1, 3, 0))
    row = rows.gather(-2, k[..., None, None].expand(*k.shape, 1, 4)).squeeze(-2)
    q = row * (0.5 / row.gather(-1, k[..., None]).sqrt())
    return torch.where(q[..., :1] < 0, -q, q)


def _hartley_sqrt2(P):
    """[N,2] -> (normalised [N,2], T [3,3]): centroid to the origin, mean distance sqrt(2)   (utils_F.py:15-41), the divisions by
    the homogeneous coordinate + 1e-10 of _de_homo included."""
    c = P.mean(0)
    s = (2.0 ** 0.5) / (P - c).norm(dim=1).mean()
    zero, one = s.new_zeros(()), s.new_ones(())
    T = torch.stack((torch.stack((s, zero, -s * c[0])), torch.stack((zero, s, -s * c[1])), torch.stack((zero, zero, one))))
    Ph = torch.cat((P, torch.ones_like(P[:, :1])), 1) @ T.t()
    return Ph[:, :2] / (Ph[:, 2:3] + 1e-10), T


def eight_point(X, Y, W=None, essential=False, normalize=True):
    """Normalised 8-point F (rank 2) or E (singular values 1, 1, 0) from X, Y [N,2], optional W [N] / [N,N] left-multiplying the
    design matrix   (utils_F.py:104-155,223-275 after the K^-1 step)."""
    T1 = T2 = None
    if normalize:
        X, T1 = _hartley_sqrt2(X)
        Y, T2 = _hartley_sqrt2(Y)
    x1, y1, x2, y2 = X[:, 0], X[:, 1], Y[:, 0], Y[:, 1]
    A = torch.stack((x2 * x1, x2 * y1, x2, y2 * x1, y2 * y1, y2, x1, y1, torch.ones_like(x1)), 1)
    if W is not None:
        A = (W[:, None] * A) if W.dim() == 1 else (W @ A)
    f = torch.linalg.svd(A, full_matrices=False)[2][-1]
    U, S, Vh = torch.linalg.svd(f.reshape(3, 3))
    S = S.new_tensor([1.0, 1.0, 0.0]) if essential else S * S.new_tensor([1.0, 1.0, 0.0])
    M = U @ torch.diag(S) @ Vh
    return T2.t() @ M @ T1 if normalize else M
