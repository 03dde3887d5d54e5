"""Mirror of the pose helpers of deepFEPE/dsac_tools/utils_geo.py used on the hot path, batched on the GPU.
The per-sample reference functions (_R_to_q :58-86, _l2_error :165-167, rot12_to_angle_error :150-155,
vector_angle :175-179) are evaluated inside the pose kernel; these wrappers expose them for single matrices too."""
import numpy as np
import torch

from .. import ops
from . import utils_misc


def _as_batch(x, shape):
    return x.reshape(-1, *shape)


def _R_to_q(R):
    """R [3,3] (or [B,3,3]) -> unit quaternion [4,1] (or [B,4,1]), q0 >= 0, trace method on R^T (utils_geo.py:58-86)."""
    if torch.is_grad_enabled() and R.requires_grad:  # the reference's is differentiable torch code: same formula through autograd
        from . import _autograd_paths as _ap

        return _ap.rot_to_quat(R).unsqueeze(-1)
    single = R.dim() == 2
    q = ops.rot_to_quat(_as_batch(R, (3, 3)))
    return q[0].unsqueeze(-1) if single else q.unsqueeze(-1)


def _l2_error(t0, t1):
    return torch.norm(t0 - t1, 2)


def rot12_to_angle_error(R0, R1):
    """Angle of R0 R1^T in degrees.  The reference goes through cv2.Rodrigues (utils_geo.py:150-155); same value."""
    R0 = torch.as_tensor(R0, dtype=torch.float32)
    R1 = torch.as_tensor(R1, dtype=torch.float32)
    dev = R0.device if R0.is_cuda else torch.device("cuda")
    return float(ops.rot_angle_deg(R0.to(dev).reshape(-1, 3, 3), R1.to(dev).reshape(-1, 3, 3))[0].item())


def vector_angle(v1, v2):
    """acos(clip(v1.v2 / ((|v1|+1e-10)(|v2|+1e-10)+1e-10))) in degrees, 0..180 (utils_geo.py:175-179)."""
    v1 = torch.as_tensor(v1, dtype=torch.float32)
    v2 = torch.as_tensor(v2, dtype=torch.float32)
    dev = v1.device if v1.is_cuda else torch.device("cuda")
    return float(ops.vector_angle_deg(v1.to(dev).reshape(1, 3), v2.to(dev).reshape(1, 3))[0].item())


def invert_Rt(R21, t21):
    """(R12, t12) of the inverse rigid transform, numpy in / numpy out (utils_geo.py:192-196; the reference pads to 4x4 and
    calls np.linalg.inv, this is the closed form of the same matrix)."""
    Rt = utils_misc.inv_Rt_np(np.hstack((np.asarray(R21), np.asarray(t21).reshape(3, 1))))
    return Rt[:, :3], Rt[:, 3:4]


# ---- the small host-side helpers of the module (numpy / elementwise torch, batched over leading dimensions) --------------------
def R_to_q_np(matrix):
    """Rotation matrix [3,3] -> unit quaternion [4,1] float32 with q0 >= 0 (utils_geo.py:88-117: the numpy twin of _R_to_q,
    what the ground truth of SURVEY §8d is built with).  A stack [n,3,3] gives [n,4,1].  The candidate table and the
    reference's branch rule live in synth.shepperd_rows; the float32 rounding happens where the reference's does (the picked
    row is rounded to float32 BEFORE the scale 1/(2 sqrt(pivot)) is applied in double)."""
    from ..synth import shepperd_rows

    S, k = shepperd_rows(np.asarray(matrix))
    row = np.take_along_axis(S, k[..., None, None], axis=-2)[..., 0, :]
    half_inv = 0.5 / np.sqrt(np.take_along_axis(row, k[..., None], axis=-1))
    q = (row.astype(np.float32).astype(half_inv.dtype) * half_inv).astype(np.float32)
    q = np.where(q[..., :1] < 0.0, -q, q)
    return q[..., None]


def _quat_blocks(q, sign):
    """[[a, -v^T], [v, a I + sign [v]_x]] for q = (a, v) given as [4,1]: sign +1 is the matrix of q (x) . , -1 of . (x) q."""
    q = np.asarray(q)
    a, v = q[0, 0], q[1:, :]
    out = np.empty((4, 4), dtype=q.dtype)
    out[0, 0], out[0, 1:], out[1:, 0] = a, -v[:, 0], v[:, 0]
    out[1:, 1:] = a * np.eye(3, dtype=q.dtype) + sign * utils_misc.skew_symmetric_np(v)
    return out


def q_matrix_np(q):
    """Left-multiplication matrix of the quaternion q [4,1] (utils_geo.py:119-126)."""
    return _quat_blocks(q, 1.0)


def q_bar_matrix_np(q):
    """Right-multiplication matrix of the quaternion q [4,1] (utils_geo.py:128-135)."""
    return _quat_blocks(q, -1.0)


def q_to_R_np(q):
    """Quaternion [4,1] (or [n,4,1]) -> rotation matrix [3,3] ([n,3,3]); q is normalised first with the reference's +1e-10
    (utils_geo.py:137-147, which multiplies the two 4x4 matrices above; this is the vector part of that product written out:
    (a^2 - v.v) I + 2 v v^T + 2 a [v]_x)."""
    q = np.asarray(q)
    q = q / (np.sqrt(np.sum(q * q, axis=(-2, -1), keepdims=True)) + 1e-10)
    a, v = q[..., :1, :], q[..., 1:, :]
    vx = np.cross(v[..., None, :, 0], -np.eye(3, dtype=q.dtype))  # rows e_i x v... = [v]_x
    eye = np.eye(3, dtype=q.dtype)
    return (a * a - np.sum(v * v, axis=-2, keepdims=True)) * eye + 2.0 * (v * np.swapaxes(v, -1, -2)) + 2.0 * a * vx


def _rot_angle_error(R0, R1):
    """Rotation angle of R0 R1^T in degrees, torch in / 0-dim torch out, differentiable like the reference's
    (utils_geo.py:158-163): tr(R0 R1^T) is the elementwise inner product <R0, R1>."""
    return torch.rad2deg(torch.acos(((R0 * R1).sum() - 1.0).div(2.0).clamp(-1.0, 1.0)))


def dotproducts(v1s, v2s):
    """Row-wise inner products [N,D],[N,D] -> [N,1] (utils_geo.py:181-182)."""
    return np.einsum("nd,nd->n", v1s, v2s)[:, None]


def vectors_angle(v1s, v2s):
    """Row-wise angle in degrees between v1s and v2s [N,3] -> [N,1]; unlike vector_angle, no epsilons and no clipping
    (utils_geo.py:184-190)."""
    cos = dotproducts(v1s, v2s) / (np.linalg.norm(v1s, axis=1, keepdims=True) * np.linalg.norm(v2s, axis=1, keepdims=True))
    return np.degrees(np.arccos(cos))
