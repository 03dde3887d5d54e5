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
