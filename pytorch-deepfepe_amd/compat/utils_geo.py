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
    if torch.is_grad_enabled() and R.requires_grad:
        from .. import _lib
        raise _lib.DfepeError("compat.utils_geo._R_to_q: not differentiable in this library (the pose loss has its own adjoint, ops.pose_errors)")
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


# ---- the small host-side helpers of the module (plain numpy / elementwise torch like the reference's: no kernel needed) --------
def R_to_q_np(matrix):
    """Rotation matrix [3,3] -> unit quaternion [4,1] float32 with q0 >= 0: the trace method on R^T with its four branches
    (utils_geo.py:88-117; the numpy twin of _R_to_q and what the synthetic ground truth of SURVEY §8d is built with)."""
    m = np.asarray(matrix).conj().transpose()
    if m[2, 2] < 0:
        if m[0, 0] > m[1, 1]:
            t = 1 + m[0, 0] - m[1, 1] - m[2, 2]
            q = [m[1, 2] - m[2, 1], t, m[0, 1] + m[1, 0], m[2, 0] + m[0, 2]]
        else:
            t = 1 - m[0, 0] + m[1, 1] - m[2, 2]
            q = [m[2, 0] - m[0, 2], m[0, 1] + m[1, 0], t, m[1, 2] + m[2, 1]]
    else:
        if m[0, 0] < -m[1, 1]:
            t = 1 - m[0, 0] - m[1, 1] + m[2, 2]
            q = [m[0, 1] - m[1, 0], m[2, 0] + m[0, 2], m[1, 2] + m[2, 1], t]
        else:
            t = 1 + m[0, 0] + m[1, 1] + m[2, 2]
            q = [t, m[1, 2] - m[2, 1], m[2, 0] - m[0, 2], m[0, 1] - m[1, 0]]
    q = np.array(q, dtype=np.float32)
    q *= 0.5 / np.sqrt(t)
    if q[0] < 0.0:
        q = -q
    return q.reshape(-1, 1)


def q_matrix_np(q):
    """Left-multiplication matrix of the quaternion q [4,1] (utils_geo.py:119-126)."""
    a, b, c, d = (q[i, 0] for i in range(4))
    return np.array([[a, -b, -c, -d], [b, a, -d, c], [c, d, a, -b], [d, -c, b, a]])


def q_bar_matrix_np(q):
    """Right-multiplication matrix of the quaternion q [4,1] (utils_geo.py:128-135)."""
    a, b, c, d = (q[i, 0] for i in range(4))
    return np.array([[a, -b, -c, -d], [b, a, d, -c], [c, -d, a, b], [d, c, -b, a]])


def q_to_R_np(q):
    """Quaternion [4,1] -> rotation matrix [3,3]; q is normalised first with the reference's +1e-10 (utils_geo.py:137-147)."""
    q = np.asarray(q)
    q = q / (np.linalg.norm(q) + 1e-10)
    product_matrix = np.dot(q_matrix_np(q), q_bar_matrix_np(q).conj().transpose())
    return product_matrix[1:][:, 1:]


def _rot_angle_error(R0, R1):
    """acos(clamp((tr(R0 R1^T) - 1) / 2)) in degrees, torch in / 0-dim torch out, differentiable like the reference's
    (utils_geo.py:158-163)."""
    rot_error = torch.acos(torch.clamp((torch.trace(R0 @ (R1.t())) - 1) / 2, -1.0, 1.0))
    return rot_error / np.pi * 180.0


def dotproducts(v1s, v2s):
    return np.sum(v1s * v2s, axis=1, keepdims=True)


def vectors_angle(v1s, v2s):
    """Row-wise angle in degrees between v1s and v2s [N,3] -> [N,1]; unlike vector_angle, no epsilons and no clipping
    (utils_geo.py:184-190)."""
    dot_v1sv2s = dotproducts(v1s, v2s)
    length_v1s = np.sqrt(dotproducts(v1s, v1s))
    length_v2s = np.sqrt(dotproducts(v2s, v2s))
    return np.arccos(dot_v1sv2s / (length_v1s * length_v2s)) / np.pi * 180.0
