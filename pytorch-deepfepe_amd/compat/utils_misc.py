"""Mirror of the helpers of deepFEPE/dsac_tools/utils_misc.py that the hot path and its callers use: homogeneous-coordinate
helpers, the cross-product matrix, rigid-transform inversion / padding and the crop-or-pad index draw.  These are tiny
host-side or elementwise operations (no kernel needed); signatures and numerics follow the reference."""
import numpy as np
import torch


def identity_Rt(dtype=np.float32):
    """[I | 0] as a 3x4 numpy array (utils_misc.py:17-18)."""
    return np.hstack((np.eye(3, dtype=dtype), np.zeros((3, 1), dtype=dtype)))


def _skew_symmetric(v):
    """[v]_x for v [3,1] -> [3,3], or [B,3,1] -> [B,3,3] (utils_misc.py:20-37)."""
    if v.dim() == 2:
        x, y, z = v[0, 0], v[1, 0], v[2, 0]
        o = torch.zeros_like(x)
        return torch.stack((o, -z, y, z, o, -x, -y, x, o)).view(3, 3)
    x, y, z = v[:, 0, 0], v[:, 1, 0], v[:, 2, 0]
    o = torch.zeros_like(x)
    return torch.stack((o, -z, y, z, o, -x, -y, x, o), dim=1).view(-1, 3, 3)


def skew_symmetric_np(v):
    """numpy twin of _skew_symmetric (utils_misc.py:39-56)."""
    v = np.asarray(v)
    if v.ndim == 2:
        x, y, z = v[0, 0], v[1, 0], v[2, 0]
        o = np.zeros_like(x)
        return np.stack((o, -z, y, z, o, -x, -y, x, o)).reshape(3, 3)
    x, y, z = v[:, 0, 0], v[:, 1, 0], v[:, 2, 0]
    o = np.zeros_like(x)
    return np.stack((o, -z, y, z, o, -x, -y, x, o), axis=1).reshape(-1, 3, 3)


def _homo(x):
    """Append a column of ones: [N,2] -> [N,3] or [B,N,2] -> [B,N,3] (utils_misc.py:58-69).  The reference prints the
    tensor's size, dtype and device on every call (:60); that debugging output is not reproduced."""
    assert x.dim() in (2, 3)
    return torch.cat((x, torch.ones(*x.shape[:-1], 1, dtype=x.dtype, device=x.device)), x.dim() - 1)


def _de_homo(x_homo):
    """Divide by the last coordinate + 1e-10 and drop it (utils_misc.py:71-80)."""
    assert x_homo.dim() in (2, 3)
    return x_homo[..., :-1] / (x_homo[..., -1] + 1e-10).unsqueeze(-1)


def homo_np(x):
    """[N,D] -> [N,D+1] (utils_misc.py:82-87)."""
    return np.hstack((x, np.ones((x.shape[0], 1), dtype=x.dtype)))


def de_homo_np(x_homo):
    """[N,D] -> [N,D-1], D in {3,4} (utils_misc.py:89-96)."""
    assert x_homo.shape[1] in (3, 4)
    return x_homo[:, :-1] / np.expand_dims(x_homo[:, -1] + 1e-10, -1)


def Rt_pad(Rt):
    """3x4 [R|t] -> 4x4 [[R|t],[0,0,0,1]] (utils_misc.py:98-101)."""
    assert Rt.shape == (3, 4)
    return np.vstack((Rt, np.array([[0.0, 0.0, 0.0, 1.0]], dtype=Rt.dtype)))


def Rt_depad(Rt01):
    """4x4 -> 3x4 (utils_misc.py:123-126)."""
    assert Rt01.shape == (4, 4)
    return Rt01[:3, :]


def inv_Rt_np(Rt):
    """[R|t]^-1 = [R^T | -R^T t], numpy (utils_misc.py:109-115)."""
    assert Rt.shape == (3, 4)
    R, t = Rt[:, :3], Rt[:, 3:4]
    return np.hstack((R.T, -R.T @ t))


def _inv_Rt(Rt):
    """[R|t]^-1 = [R^T | -R^T t] for a 3x4 tensor (utils_misc.py:117-123); differentiable."""
    assert tuple(Rt.shape) == (3, 4)
    R, t = Rt[:, :3], Rt[:, 3:4]
    return torch.cat((R.t(), -R.t() @ t), 1)


def crop_or_pad_choice(in_num_points, out_num_points, shuffle=False):
    """Indices that crop or pad ``in_num_points`` items to ``out_num_points`` (utils_misc.py:139-161).  Host-side and
    drawn from numpy's global RNG with the reference's call sequence (one permutation when shuffling, one
    ``np.random.choice`` when padding), so a seeded run selects the same correspondences as the reference."""
    choice = np.random.permutation(in_num_points) if shuffle else np.arange(in_num_points)
    assert out_num_points > 0, "out_num_points = %d must be positive int!" % out_num_points
    if in_num_points >= out_num_points:
        return choice[:out_num_points]
    pad = np.random.choice(choice, out_num_points - in_num_points, replace=True)
    return np.concatenate([choice, pad])
