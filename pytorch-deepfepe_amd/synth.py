"""Seeded synthetic two-view scenes for the weighted-8-point hot path.

Host-side measurement/test utility (CPU tensors; callers move them to the GPU).
It follows the *input contract* of the reference's training batch
(deepFEPE/Train_model_pipeline.py:433-446, deepFEPE/datasets/kitti_odo_corr.py:200-211):

  matches_xy_ori [B,N,4]  pixel coordinates (x1,y1,x2,y2)
  Ks             [B,3,3]  intrinsics (KITTI-like, configs/kitti_corr_baseline.yaml:24 image 376x1241)
  delta_Rtijs_4_4[B,4,4]  *scene* motion  X2 = R X1 + t   (E = [t]x R, x2^T E x1 = 0)
  qs_cam [B,4,1], ts_cam [B,3,1]   quaternion / translation of the *camera* motion (inverse of the above)
  pts1_virt_ori, pts2_virt_ori [B,M,3]  homogeneous pixel "virtual" correspondences lying exactly on the
                                        ground-truth epipolar geometry (stand-in for cv2.correctMatches grid,
                                        deepFEPE/dsac_tools/utils_misc.py:163-199)
There is no dataset and no OpenCV in this environment, so everything is generated.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch

IMAGE_H, IMAGE_W = 376, 1241
KITTI_K = ((718.856, 0.0, 607.1928), (0.0, 718.856, 185.2157), (0.0, 0.0, 1.0))


def _expm_so3(omega: torch.Tensor) -> torch.Tensor:
    """Rodrigues formula, omega [B,3] -> R [B,3,3]."""
    theta = omega.norm(dim=1, keepdim=True).clamp_min(1e-12)
    k = omega / theta
    Kx = torch.zeros(omega.shape[0], 3, 3, dtype=omega.dtype)
    Kx[:, 0, 1], Kx[:, 0, 2] = -k[:, 2], k[:, 1]
    Kx[:, 1, 0], Kx[:, 1, 2] = k[:, 2], -k[:, 0]
    Kx[:, 2, 0], Kx[:, 2, 1] = -k[:, 1], k[:, 0]
    s = torch.sin(theta).unsqueeze(-1)
    c = torch.cos(theta).unsqueeze(-1)
    eye = torch.eye(3, dtype=omega.dtype).expand_as(Kx)
    return eye + s * Kx + (1 - c) * (Kx @ Kx)


def shepperd_rows(R: np.ndarray):
    """The symmetric 4x4 matrix S(R) whose rows are the four un-normalised quaternion candidates (w,x,y,z) of Shepperd's
    trace method, for R [...,3,3]: diagonal 1 +- R00 +- R11 +- R22 (the pivot 4 q_k^2), first row / column the antisymmetric
    part of R (4 q_w q_k), the rest its symmetric off-diagonal sums (4 q_j q_k).  Row k is 4 q_k * q, so any row with a
    safely positive pivot gives q after division by 2 sqrt(pivot).  Returns (S [...,4,4], k [...]): k is the row the
    dataset-side helper of the reference picks (deepFEPE/dsac_tools/utils_geo.py:88-117: R22 < 0 ? (R00 > R11 ? x : y) :
    (R00 < -R11 ? z : w)); pivots are summed left to right like there, so values agree to the bit."""
    R = np.asarray(R)
    d0, d1, d2 = R[..., 0, 0], R[..., 1, 1], R[..., 2, 2]
    S = np.empty(R.shape[:-2] + (4, 4), dtype=np.result_type(R.dtype, np.float32))
    S[..., 0, 0] = 1 + d0 + d1 + d2
    S[..., 1, 1] = 1 + d0 - d1 - d2
    S[..., 2, 2] = 1 - d0 + d1 - d2
    S[..., 3, 3] = 1 - d0 - d1 + d2
    for k, (i, j) in enumerate(((2, 1), (0, 2), (1, 0)), start=1):  # (i, j, k-1) cyclic
        S[..., 0, k] = S[..., k, 0] = R[..., i, j] - R[..., j, i]
        a, b = (k % 3) + 1, ((k + 1) % 3) + 1  # the two other vector components
        S[..., a, b] = S[..., b, a] = R[..., j, i] + R[..., i, j]
    pick = np.where(d2 < 0, np.where(d0 > d1, 1, 2), np.where(d0 < -d1, 3, 0))
    return S, pick


def rotation_to_quaternion_np(R: np.ndarray) -> np.ndarray:
    """Trace-method quaternion (w,x,y,z), w >= 0, of rotations R [...,3,3] -> [...,4] float64 (the branch rule of the helper
    the reference builds its ground truth with, deepFEPE/dsac_tools/utils_geo.py:88-117; see shepperd_rows)."""
    S, k = shepperd_rows(np.asarray(R, dtype=np.float64))
    row = np.take_along_axis(S, k[..., None, None], axis=-2)[..., 0, :]
    piv = np.take_along_axis(row, k[..., None], axis=-1)
    q = row * (0.5 / np.sqrt(piv))
    return np.where(q[..., :1] < 0, -q, q)


def make_scene(
    B: int,
    N: int,
    seed: int = 0,
    outlier_ratio: float = 0.0,
    noise_px: float = 0.5,
    planar: bool = False,
    M_virt: int = 100,
    depth_layers: int = 5,
    dtype: torch.dtype = torch.float32,
) -> Dict[str, torch.Tensor]:
    """Generate one batch of synthetic pairs (CPU tensors).

    3-D points are drawn inside the first camera's frustum (uniform pixel, depth in [5,35] m),
    moved by a small random rigid motion (rotation ~ N(0, 0.03^2) rad per axis, translation
    ~ N(0, diag(.2,.1,1)^2) m, i.e. mostly forward like KITTI) and projected into both views.
    ``outlier_ratio`` of the rows (the first ones) get a uniformly random second-view point.
    ``planar`` puts the points on the plane Z = 10 + 0.2 X (degenerate for the 8-point system).
    """
    g = torch.Generator().manual_seed(int(seed))
    f64 = torch.float64
    K = torch.tensor(KITTI_K, dtype=f64)
    Kinv = torch.linalg.inv(K)

    def rand(*s):
        return torch.rand(*s, generator=g, dtype=f64)

    def randn(*s):
        return torch.randn(*s, generator=g, dtype=f64)

    def sample_points(n):
        uv = torch.stack((rand(B, n) * IMAGE_W, rand(B, n) * IMAGE_H, torch.ones(B, n, dtype=f64)), -1)
        ray = uv @ Kinv.T  # [B,n,3], z = 1
        if planar:
            # Z = 10 + 0.2 X with X = z * ray_x  ->  z = 10 / (1 - 0.2 ray_x)
            z = 10.0 / (1.0 - 0.2 * ray[..., 0])
        else:
            z = 5.0 + 30.0 * rand(B, n)
        return ray * z.unsqueeze(-1)

    omega = 0.03 * randn(B, 3)
    R = _expm_so3(omega)
    t = randn(B, 3) * torch.tensor([0.2, 0.1, 1.0], dtype=f64)

    def project(X):
        x = X @ K.T
        return x[..., :2] / x[..., 2:3]

    X1 = sample_points(N)
    X2 = X1 @ R.transpose(1, 2) + t.unsqueeze(1)
    x1 = project(X1) + noise_px * randn(B, N, 2)
    x2 = project(X2) + noise_px * randn(B, N, 2)
    n_out = int(math.floor(outlier_ratio * N))
    if n_out > 0:
        x2[:, :n_out, 0] = rand(B, n_out) * IMAGE_W
        x2[:, :n_out, 1] = rand(B, n_out) * IMAGE_H
    matches = torch.cat((x1, x2), -1)

    V1 = sample_points(M_virt)
    V2 = V1 @ R.transpose(1, 2) + t.unsqueeze(1)
    ones = torch.ones(B, M_virt, 1, dtype=f64)
    pts1_virt = torch.cat((project(V1), ones), -1)
    pts2_virt = torch.cat((project(V2), ones), -1)

    delta = torch.eye(4, dtype=f64).repeat(B, 1, 1)
    delta[:, :3, :3] = R
    delta[:, :3, 3] = t
    R_cam = R.transpose(1, 2)
    t_cam = -(R_cam @ t.unsqueeze(-1))  # [B,3,1]
    q_cam = torch.from_numpy(np.stack([rotation_to_quaternion_np(r) for r in R_cam.numpy()])).unsqueeze(-1)

    tx = torch.zeros(B, 3, 3, dtype=f64)
    tx[:, 0, 1], tx[:, 0, 2] = -t[:, 2], t[:, 1]
    tx[:, 1, 0], tx[:, 1, 2] = t[:, 2], -t[:, 0]
    tx[:, 2, 0], tx[:, 2, 1] = -t[:, 1], t[:, 0]
    E_gt = tx @ R
    F_gt = Kinv.T @ E_gt @ Kinv

    logits = randn(depth_layers, B, N)

    out = {
        "matches_xy_ori": matches,
        "Ks": K.expand(B, 3, 3).clone(),
        "delta_Rtijs_4_4": delta,
        "qs_cam": q_cam,
        "ts_cam": t_cam,
        "pts1_virt_ori": pts1_virt,
        "pts2_virt_ori": pts2_virt,
        "E_gt": E_gt,
        "F_gt": F_gt,
        "logits_layers": logits,
    }
    return {k: v.to(dtype).contiguous() for k, v in out.items()}


def fill_params_deterministic(module: torch.nn.Module, seed: int = 0) -> None:
    """Overwrite every parameter of ``module`` with seeded values that depend only on the parameter's
    name order and shape (not on construction order), so the reference's ErrorEstimator (in the golden
    generator) and this repo's compat ErrorEstimator get identical weights without shipping a
    state_dict: there are no pretrained blobs."""
    with torch.no_grad():
        for idx, (name, p) in enumerate(sorted(module.named_parameters(), key=lambda kv: kv[0])):
            g = torch.Generator().manual_seed(int(seed) * 1000003 + idx)
            r = torch.randn(p.shape, generator=g, dtype=torch.float64)
            if p.dim() >= 2:  # conv weight [out,in,1]
                r = r / math.sqrt(p.shape[1])
            elif name.endswith("weight"):  # instance-norm scale
                r = 1.0 + 0.1 * r
            else:  # biases
                r = 0.05 * r
            p.copy_(r.to(p.dtype))
