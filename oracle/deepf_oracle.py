"""CPU oracle for the deepFEPE weighted-8-point hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch CPU restatement (plain PyTorch on CPU, dtype-generic so the same
code runs in fp32 like the reference and in fp64 as the accuracy yard-stick) of the reference's
algorithm for the path SURVEY.md §8 scopes.  It is *not* part of the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it, and only as
the checker / the CPU baseline.  The product path (``pytorch-deepfepe_amd``) never imports it
and fails loudly when its HIP library is missing.

Pinning: the reference ships no golden vectors for this path, so this oracle is
pinned against outputs of the reference itself, imported in the build container by
``tests/golden/make_golden.py``; the resulting arrays are committed under ``tests/golden/`` and
``tests/test_oracle_golden.py`` checks every function below against them.
OpenCV-backed pieces (``cv2.Rodrigues``, ``cv2.triangulatePoints``, ``cv2.recoverPose``) have
no reference arithmetic in the tree (third-party ``opencv-python==3.4.2.16``,
requirements.txt:7-8): those rows are "parity unpinned" and are validated against geometric
ground truth instead (rotation_angle_deg, triangulate_dlt, cheirality_select below).

All ``file:line`` citations are into /root/reference/deepFEPE/.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, Optional, Sequence, Tuple

import numpy as np
import torch

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------
# a1: image-size normalisation  (models/DeepFNet.py:93-120  NormalizeAndExpand_HW)
# --------------------------------------------------------------------------------------
def hw_matrix(image_size: Sequence[int], dtype=torch.float32) -> Tensor:
    """T = [[2/W,0,-1],[0,2/H,-1],[0,0,1]] with H,W = image_size[0:2] (DeepFNet.py:103,111)."""
    H, W = float(image_size[0]), float(image_size[1])
    return torch.tensor([[2.0 / W, 0.0, -1.0], [0.0, 2.0 / H, -1.0], [0.0, 0.0, 1.0]], dtype=dtype)


def normalize_hw(matches: Tensor, image_size: Sequence[int]) -> Tuple[Tensor, Tensor, Tensor]:
    """matches [B,N,4] pixel -> pts1, pts2 [B,N,3] in [-1,1]^2 homogeneous, T [B,3,3].

    DeepFNet.py:108-120 builds [B,3,N]; get_input (:385-386) permutes to [B,N,3], which is what
    every consumer sees, so that is what is returned here.
    """
    B, N, _ = matches.shape
    T = hw_matrix(image_size, matches.dtype)
    ones = torch.ones(B, N, 1, dtype=matches.dtype)
    p1 = torch.cat((matches[:, :, 0:2], ones), 2) @ T.T
    p2 = torch.cat((matches[:, :, 2:4], ones), 2) @ T.T
    return p1, p2, T.unsqueeze(0).expand(B, 3, 3)


def estimator_input(pts1: Tensor, pts2: Tensor) -> Tensor:
    """[B,4,N] = cat((x1+1)/2, (x2+1)/2) (DeepFNet.py:388-392, no-quality branch)."""
    return torch.cat(((pts1[:, :, :2] + 1) / 2, (pts2[:, :, :2] + 1) / 2), 2).permute(0, 2, 1)


# --------------------------------------------------------------------------------------
# a3: Hartley normalisation with unit weights  (models/DeepFNet.py:148-179, called :198-199)
# --------------------------------------------------------------------------------------
def hartley(pts: Tensor) -> Tuple[Tensor, Tensor]:
    """pts [B,N,3] -> (normalised [B,N,3], T [B,3,3]); scale literal 1.4142 (DeepFNet.py:168)."""
    c = pts.mean(1)  # [B,3]
    d = (pts[:, :, :2] - c[:, None, :2]).pow(2).sum(2).sqrt().mean(1)  # [B]
    s = 1.4142 / d
    T = torch.zeros(pts.shape[0], 3, 3, dtype=pts.dtype)
    T[:, 0, 0] = s
    T[:, 1, 1] = s
    T[:, 2, 2] = 1
    T[:, 0, 2] = -c[:, 0] * s
    T[:, 1, 2] = -c[:, 1] * s
    return pts @ T.transpose(1, 2), T


# --------------------------------------------------------------------------------------
# a4/a5: the weighted normalised 8-point fit  (models/DeepFNet.py:181-257, 278-295)
# --------------------------------------------------------------------------------------
def fit_rows(pts1: Tensor, pts2: Tensor, weights: Tensor, normalize_svd: bool = True):
    """Rows of the design matrix: p (unit rows unless Fit(normalize_SVD=False), :203-212) and X = p*w (:214; w not sqrt'ed)."""
    w = weights.reshape(weights.shape[0], -1, 1)  # [B,N,1]  (:192)
    a, T1 = hartley(pts1)
    b, T2 = hartley(pts2)
    p = torch.cat((b[:, :, 0:1] * a, b[:, :, 1:2] * a, a), 2)  # [x2*x1,x2*y1,x2, y2*x1,y2*y1,y2, x1,y1,1]
    if normalize_svd:
        p = p / p.norm(dim=2, keepdim=True).clamp_min(1e-12)  # F.normalize(dim=2) (:211-212)
    return p, p * w, T1, T2


def _smallest_right_singular_vector(X: Tensor, mode: str) -> Tensor:
    """V[:, -1] of svd(X[b]) for every b (:232-235)."""
    if mode == "loop":  # reference-shaped: one LAPACK call per sample in a Python loop
        return torch.stack([torch.linalg.svd(X[b], full_matrices=False)[2][-1] for b in range(X.shape[0])])
    return torch.linalg.svd(X, full_matrices=False)[2][:, -1, :]


def _rank2(Fm: Tensor, mode: str) -> Tensor:
    """U diag(S*[1,1,0]) V^T of the 3x3 (:236-237)."""
    if mode == "loop":
        out = []
        for b in range(Fm.shape[0]):
            U, S, Vh = torch.linalg.svd(Fm[b])
            out.append(U @ torch.diag(S * torch.tensor([1.0, 1.0, 0.0], dtype=Fm.dtype)) @ Vh)
        return torch.stack(out)
    U, S, Vh = torch.linalg.svd(Fm)
    S = S * torch.tensor([1.0, 1.0, 0.0], dtype=Fm.dtype)
    return U @ torch.diag_embed(S) @ Vh


def fit_forward(pts1: Tensor, pts2: Tensor, weights: Tensor, mode: str = "batched", normalize_svd: bool = True):
    """Fit.forward: pts [B,N,3], weights [B,1,N] -> (out [B,3,3], residual [B,N], aux).

    out = T2^T F' T1 (:256), residual = X f/|f| (:251).  The sign of f is whatever the SVD
    returns (the reference has no convention); callers compare up to a per-pair sign.
    """
    p, X, T1, T2 = fit_rows(pts1, pts2, weights, normalize_svd)
    f = _smallest_right_singular_vector(X, mode)  # [B,9]
    fhat = f / f.norm(dim=1, keepdim=True)
    Fp = _rank2(f.reshape(-1, 3, 3), mode)
    residual = (X @ fhat.unsqueeze(-1)).squeeze(-1)
    out = T2.transpose(1, 2) @ Fp @ T1
    return out, residual, {"p": p, "X": X, "T1": T1, "T2": T2, "f": fhat, "F_rank2": Fp}


# --------------------------------------------------------------------------------------
# a6: symmetric epipolar residual  (dsac_tools/utils_F.py:400-413)
# --------------------------------------------------------------------------------------
def compute_epi_residual(pts1: Tensor, pts2: Tensor, F: Tensor, clamp_at: float = 0.5) -> Tensor:
    l1 = pts2 @ F  # rows: F^T x2
    l2 = pts1 @ F.transpose(1, 2)  # rows: F x1
    dd = (pts1 * l1).sum(2)
    d = dd.abs() * (1 / (l1[:, :, :2].norm(dim=2) + 1e-6) + 1 / (l2[:, :, :2].norm(dim=2) + 1e-6))
    return torch.clamp(d, max=clamp_at)


# --------------------------------------------------------------------------------------
# a7: the recurrent forward  (models/DeepFNet.py:429-554)
# --------------------------------------------------------------------------------------
def deepf_forward(
    matches: Tensor,
    image_size: Sequence[int],
    depth: int,
    input_weights: Optional[Callable[[Tensor], Tensor]] = None,
    update_weights: Optional[Callable[[Tensor], Tensor]] = None,
    logits_layers: Optional[Tensor] = None,
    mode: str = "batched",
) -> Dict[str, object]:
    """DeepFNet.forward with either the two estimator callables (logits [B,1,N] from [B,C,N])
    or fixed per-layer logits ``logits_layers`` [depth,B,N] ("solver-only" variant).
    Returns the same dict keys as DeepFNet.py:534-548.
    """
    pts1, pts2, T = normalize_hw(matches, image_size)
    net_in0 = estimator_input(pts1, pts2)

    def logits_for(layer: int, net_in: Tensor) -> Tensor:
        if logits_layers is not None:
            return logits_layers[layer].unsqueeze(1)
        return (input_weights if layer == 0 else update_weights)(net_in)

    logits = logits_for(0, net_in0)
    w = torch.softmax(logits, dim=2)
    out_layers, epi_res_layers, residual_layers = [], [], []
    weights_layers, logits_list = [w], [logits]
    for it in range(depth - 1):
        out, residual, _ = fit_forward(pts1, pts2, w, mode)
        out_layers.append(out)
        residual_layers.append(residual)
        epi = compute_epi_residual(pts1, pts2, out).unsqueeze(1)  # default clamp 0.5 (:479)
        epi_res_layers.append(epi)
        net_in = torch.cat((net_in0, w, epi, residual.unsqueeze(1)), 1)  # (:487)
        logits = logits_for(it + 1, net_in)
        w = torch.softmax(logits, dim=2)
        weights_layers.append(w)
        logits_list.append(logits)
    out, residual, _ = fit_forward(pts1, pts2, w, mode)
    residual_layers.append(residual)
    out_layers.append(out)
    return {
        "logits": logits.squeeze(1),
        "logits_layers": logits_list,
        "F_est": out,
        "epi_res_layers": epi_res_layers,
        "T1": T,
        "T2": T,
        "out_layers": out_layers,
        "pts1": pts1,
        "pts2": pts2,
        "weights": w,
        "residual_layers": residual_layers,
        "weights_layers": weights_layers,
    }


# --------------------------------------------------------------------------------------
# a8: F-loss and E-from-F  (train_good_utils.py:298-520)
# --------------------------------------------------------------------------------------
def f_loss(outs: Dict[str, object], pts1_virt: Tensor, pts2_virt: Tensor, Ks: Tensor, depth: int, clamp_at: float):
    T1, T2 = outs["T1"], outs["T2"]
    p1 = pts1_virt @ T1.transpose(1, 2)  # (:325)
    p2 = pts2_virt @ T2.transpose(1, 2)  # (:326)
    loss_layers, losses_layers, E_layers, per_pair = [], [], [], []
    for i in range(depth):
        losses = compute_epi_residual(p1, p2, outs["out_layers"][i], clamp_at)  # (:340-342)
        per_pair.append(losses.mean(dim=1))
        losses_layers.append(losses)
        loss_layers.append(losses.mean())
        E_layers.append(Ks.transpose(1, 2) @ T2.transpose(1, 2) @ outs["out_layers"][i] @ T1 @ Ks)  # (:356-358)
    per_pair = torch.stack(per_pair)  # [L,B]
    F_ests = T2.transpose(1, 2) @ outs["F_est"] @ T1  # (:366-368)
    E_ests = Ks.transpose(1, 2) @ F_ests @ Ks  # (:369)
    losses = {
        "loss_layers": loss_layers,
        "loss_F": sum(loss_layers) / len(loss_layers),  # (:364)
        "loss_min_layers": per_pair.min(dim=1)[0],  # (:375)
        "loss_min_batch": per_pair.min(dim=0)[0],  # (:376)
        "loss_per_pair": per_pair,
    }
    if depth > 1 and len(outs.get("epi_res_layers", [])) > 0:  # (:429-438) logged only
        l = [(e * w).mean() for e, w in zip(outs["epi_res_layers"], outs["weights_layers"])]
        losses["loss_epi_res_layers"] = l
        losses["loss_epi_res"] = sum(l) / len(l)
    return losses, E_ests, F_ests, E_layers


# --------------------------------------------------------------------------------------
# a9/a10: pose candidates and quaternion  (utils_F.py:478-498, utils_geo.py:58-86,165-167)
# --------------------------------------------------------------------------------------
def get_M2s(E: Tensor):
    """E [3,3] -> ([R1,R2], [t,-t]).  W flips (not U) when det(U W V^T) < 0 (:482-486)."""
    U, S, Vh = torch.linalg.svd(E)
    W = torch.tensor([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]], dtype=E.dtype)
    if torch.det(U @ W @ Vh) < 0:
        W = -W
    t = U[:, 2:3] / U[:, 2:3].norm()
    return [U @ W @ Vh, U @ W.T @ Vh], [t, -t]


def R_to_q(R: Tensor) -> Tensor:
    """Trace-method quaternion [4,1] with q0 >= 0 (utils_geo.py:58-86)."""
    m = R.T
    if m[2, 2] < 0:
        if m[0, 0] > m[1, 1]:
            t = 1 + m[0, 0] - m[1, 1] - m[2, 2]
            q = torch.stack((m[1, 2] - m[2, 1], t, m[0, 1] + m[1, 0], m[2, 0] + m[0, 2]))
        else:
            t = 1 - m[0, 0] + m[1, 1] - m[2, 2]
            q = torch.stack((m[2, 0] - m[0, 2], m[0, 1] + m[1, 0], t, m[1, 2] + m[2, 1]))
    else:
        if m[0, 0] < -m[1, 1]:
            t = 1 - m[0, 0] - m[1, 1] + m[2, 2]
            q = torch.stack((m[0, 1] - m[1, 0], m[2, 0] + m[0, 2], m[1, 2] + m[2, 1], t))
        else:
            t = 1 + m[0, 0] + m[1, 1] + m[2, 2]
            q = torch.stack((t, m[1, 2] - m[2, 1], m[2, 0] - m[0, 2], m[0, 1] - m[1, 0]))
    q = q * (0.5 / torch.sqrt(t))
    if q[0] < 0:
        q = -q
    return q.unsqueeze(-1)


# --------------------------------------------------------------------------------------
# a12: angular metrics  (utils_geo.py:150-155, 175-179)  -- cv2.Rodrigues: parity unpinned
# --------------------------------------------------------------------------------------
def rotation_angle_deg(R0: np.ndarray, R1: np.ndarray) -> float:
    """|log(R0 R1^T)| in degrees.  The reference calls cv2.Rodrigues (:151) and notes the
    equivalent acos((tr-1)/2) form itself (:153); OpenCV is absent here, so this stand-in is
    validated against the generating angle of synthetic rotations, not against OpenCV.
    Uses atan2(|axis|, tr-1) which is accurate near 0 and near pi."""
    R = np.asarray(R0, dtype=np.float64) @ np.asarray(R1, dtype=np.float64).T
    v = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    return float(np.degrees(math.atan2(np.linalg.norm(v), np.trace(R) - 1.0)))


def vector_angle_deg(v1: np.ndarray, v2: np.ndarray) -> float:
    """acos(clip(v1.v2 / ((|v1|+1e-10)(|v2|+1e-10)+1e-10))) in degrees; no abs -> 0..180 (:175-179)."""
    v1 = np.asarray(v1, dtype=np.float64).reshape(-1)
    v2 = np.asarray(v2, dtype=np.float64).reshape(-1)
    l1 = math.sqrt(float(v1 @ v1)) + 1e-10
    l2 = math.sqrt(float(v2 @ v2)) + 1e-10
    return float(np.degrees(math.acos(np.clip(float(v1 @ v2) / (l1 * l2 + 1e-10), -1.0, 1.0))))


# --------------------------------------------------------------------------------------
# a11: pose loss  (train_good_utils.py:64-295)
# --------------------------------------------------------------------------------------
def rt_loss(E_layers: Sequence[Tensor], delta_Rtijs_4_4: Tensor, qs_cam: Tensor, ts_cam: Tensor):
    """Per layer and pair: decompose E^T (:106), compare both R (as quaternions) and both t with
    the ground truth, keep the smaller by strict '<' (:160-168), report angle metrics.
    Returns per-layer tensors q_l2 [L,B], t_l2 [L,B] (differentiable) and numpy R_deg, t_deg [L,B]
    plus the picked candidate indices.  The caller-side means / the 'q_l2_error_list' slip of the
    reference (:276) live in the compat layer, not here.
    """
    delta_inv = torch.linalg.inv(delta_Rtijs_4_4)  # (:134)
    L, B = len(E_layers), E_layers[0].shape[0]
    q_l2 = [[None] * B for _ in range(L)]
    t_l2 = [[None] * B for _ in range(L)]
    R_deg = np.zeros((L, B))
    t_deg = np.zeros((L, B))
    sel = np.zeros((L, B, 2), dtype=np.int64)
    for l, E in enumerate(E_layers):
        for b in range(B):
            Rs, ts = get_M2s(E[b].T)
            q1, q2 = R_to_q(Rs[0]), R_to_q(Rs[1])
            t_gt = ts_cam[b] / ts_cam[b].norm().clamp_min(1e-12)  # F.normalize(p=2, dim=0) (:151)
            qe = [(q1 - qs_cam[b]).norm(), (q2 - qs_cam[b]).norm()]
            te = [(ts[0] - t_gt).norm(), (ts[1] - t_gt).norm()]
            qi = 0 if bool(qe[0] < qe[1]) else 1
            ti = 0 if bool(te[0] < te[1]) else 1
            q_l2[l][b], t_l2[l][b] = qe[qi], te[ti]
            sel[l, b] = (qi, ti)
            R_deg[l, b] = rotation_angle_deg(Rs[qi].detach().numpy(), delta_inv[b, :3, :3].detach().numpy())
            t_deg[l, b] = vector_angle_deg(ts[ti].detach().numpy(), t_gt.detach().numpy())
    q_l2 = torch.stack([torch.stack(r) for r in q_l2])
    t_l2 = torch.stack([torch.stack(r) for r in t_l2])
    return {"q_l2": q_l2, "t_l2": t_l2, "R_deg": R_deg, "t_deg": t_deg, "sel": sel}


def qt_training_loss(q_l2: Tensor, t_l2: Tensor, clamp_q: float, clamp_t: float, balance_q: float, balance_t: float):
    """clamp(stack(q),0,cq).mean()*bq + clamp(stack(t),0,ct).mean()*bt (Train_model_pipeline.py:580-586)."""
    return torch.clamp(q_l2, 0.0, clamp_q).mean() * balance_q + torch.clamp(t_l2, 0.0, clamp_t).mean() * balance_t


# --------------------------------------------------------------------------------------
# a13: cheirality  (utils_F.py:679-763)  -- cv2.triangulatePoints: parity unpinned
# --------------------------------------------------------------------------------------
def triangulate_dlt(P1: np.ndarray, P2: np.ndarray, x1: np.ndarray, x2: np.ndarray) -> np.ndarray:
    """Linear (DLT) triangulation: null vector of [x*P3-P1; y*P3-P2] over both views, [4,N]."""
    N = x1.shape[0]
    X = np.zeros((4, N))
    for i in range(N):
        A = np.stack(
            (
                x1[i, 0] * P1[2] - P1[0],
                x1[i, 1] * P1[2] - P1[1],
                x2[i, 0] * P2[2] - P2[0],
                x2[i, 1] * P2[2] - P2[1],
            )
        )
        X[:, i] = np.linalg.svd(A)[2][-1]
    return X


def cheirality_select(E: Tensor, K: np.ndarray, x1: np.ndarray, x2: np.ndarray, depth_thres: float = 50.0):
    """_E_to_M_train: triangulate with each of the 4 candidates in order (R1,t),(R1,-t),(R2,t),(R2,-t)
    (:495-497), count points with 0<Z<thres in both cameras (:718-725), first argmax wins (:730);
    returns (camera-motion [3,4] = inverse of the winner or None, winner index, counts)."""
    Rs, ts = get_M2s(E)
    K = np.asarray(K, dtype=np.float64)
    P1 = K @ np.hstack((np.eye(3), np.zeros((3, 1))))
    counts, cands = [], []
    for R in Rs:
        for t in ts:
            Rn, tn = R.detach().numpy().astype(np.float64), t.detach().numpy().astype(np.float64)
            Xh = triangulate_dlt(P1, K @ np.hstack((Rn, tn)), x1, x2)
            X = Xh[:3] / Xh[3]
            z1 = X[2]
            z2 = (Rn @ X + tn)[2]
            counts.append(int(np.sum((z1 > 0) & (z1 < depth_thres) & (z2 > 0) & (z2 < depth_thres))))
            cands.append((R, t))
    win = int(np.argmax(counts))  # first maximum, like max(enumerate(...)) (:730)
    if counts[win] == 0:
        return None, win, counts
    R, t = cands[win]
    return torch.cat((R.T, -R.T @ t), 1), win, counts  # utils_misc._inv_Rt (:115-121)


# --------------------------------------------------------------------------------------
# a15: evaluation-time pose  (utils_F.py:909-954 goodCorr_eval_nondecompose, train_good_utils.py:553-646 val_rt)
#      own-source logic pinned by tests/golden/valrt.npz; cv2.recoverPose itself: parity unpinned
# --------------------------------------------------------------------------------------
def recover_pose(E: np.ndarray, p1: np.ndarray, p2: np.ndarray, focal: float, pp, dist: float = 50.0):
    """Restatement of the published algorithm of cv2.recoverPose(E, p1, p2, focal=, pp=) (OpenCV 3.4; call site
    utils_F.py:936): decomposeEssentialMat (U, V^T made proper rotations, W = [[0,1,0],[-1,0,0],[0,0,1]], t = U[:,2]), the
    candidates [R1|t], [R2|t], [R1|-t], [R2|-t], linear triangulation of the (x - pp) / focal points against [I|0], a point
    counts when its depth is in (0, dist) in both cameras, the first candidate with the largest count wins.  Returns
    (count, R, t, counts) in the scene convention x2 ~ R x1 + t."""
    E = np.asarray(E, dtype=np.float64)
    q1 = (np.asarray(p1, dtype=np.float64) - np.asarray(pp, dtype=np.float64)) / focal
    q2 = (np.asarray(p2, dtype=np.float64) - np.asarray(pp, dtype=np.float64)) / focal
    U, _, Vt = np.linalg.svd(E)
    U = -U if np.linalg.det(U) < 0 else U
    Vt = -Vt if np.linalg.det(Vt) < 0 else Vt
    Wm = np.array([[0.0, 1.0, 0.0], [-1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
    R1, R2, t = U @ Wm @ Vt, U @ Wm.T @ Vt, U[:, 2:3]
    P0 = np.hstack((np.eye(3), np.zeros((3, 1))))
    cands = ((R1, t), (R2, t), (R1, -t), (R2, -t))
    counts = []
    for R, tt in cands:
        P = np.hstack((R, tt))
        Q = triangulate_dlt(P0, P, q1, q2)
        ok = (Q[2] * Q[3]) > 0
        Q = Q / Q[3]
        ok &= Q[2] < dist
        z2 = (P @ Q)[2]
        ok &= (z2 > 0) & (z2 < dist)
        counts.append(int(ok.sum()))
    k = int(np.argmax(counts))
    return counts[k], cands[k][0], cands[k][1], counts


def good_corr_eval_nondecompose(p1s, p2s, E_hat, delta_Rtij_inv, K, scores=None):
    """utils_F.goodCorr_eval_nondecompose (:909-954): optional top-10 % score mask (threshold = the (num_top)-th largest score,
    kept with `>=`, :912-919); fewer than 5 correspondences: (180, 90) degrees and the identity pose (:949-952); otherwise
    recoverPose, utils_geo.invert_Rt (utils_geo.py:192-196) and the angles of the camera motion against the ground truth
    (:942-944).  Returns (hstack(R, t) of the scene motion, (err_q_deg, err_t_deg))."""
    p1s, p2s = np.asarray(p1s), np.asarray(p2s)
    if scores is not None:
        scores = np.asarray(scores)
        num_top = max(1, len(scores) // 10)
        th = np.sort(scores)[::-1][num_top]
        mask = scores >= th
        p1s, p2s = p1s[mask], p2s[mask]
    if p1s.shape[0] < 5:
        return np.hstack((np.eye(3, dtype=np.float32), np.zeros((3, 1), np.float32))), (180.0, 90.0)
    K = np.asarray(K)
    _, R, t, _ = recover_pose(E_hat, p1s, p2s, focal=float(K[0, 0]), pp=(float(K[0, 2]), float(K[1, 2])))
    R_cam, t_cam = R.T, -R.T @ t  # invert_Rt
    gt = np.asarray(delta_Rtij_inv, dtype=np.float64)
    err_q = rotation_angle_deg(R_cam, gt[:3, :3])
    err_t = vector_angle_deg(t_cam.reshape(3), gt[:3, 3].reshape(3))
    return np.hstack((R, t)), (err_q, err_t)


def epi_distance_np(F: np.ndarray, X: np.ndarray, Y: np.ndarray):
    """utils_F.epi_distance_np (:363-385), if_homo=False, 2-D form: |y^T F x| (1/|Fx|_xy + 1/|F^T y|_xy); returns (d1 + d2, d1, d2)."""
    F = np.asarray(F, dtype=np.float64)  # the yard-stick evaluates in float64 whatever the inputs are
    Xh = np.hstack((np.asarray(X, dtype=np.float64), np.ones((len(X), 1))))
    Yh = np.hstack((np.asarray(Y, dtype=np.float64), np.ones((len(Y), 1))))
    num = np.abs(np.einsum("ni,ij,nj->n", Yh, F, Xh))
    Fx1, Fx2 = F @ Xh.T, F.T @ Yh.T
    r1, r2 = 1.0 / np.sqrt(Fx1[0] ** 2 + Fx1[1] ** 2), 1.0 / np.sqrt(Fx2[0] ** 2 + Fx2[1] ** 2)
    return num * (r1 + r2), num * r1, num * r2


def val_rt(K, x1, x2, E_est, E_gt, F_est, F_gt, delta_Rtij_4_4):
    """train_good_utils.val_rt (:553-646) without its OpenCV-baseline leg: pose errors of the estimated and of the ground-truth E
    against inv(delta)[:3] (:581,586-601), epi_distance_np of both F over the pair's correspondences (:602-607).
    Returns dict(err_est, epi_est, err_gt, epi_gt, M_est)."""
    dinv = np.linalg.inv(np.asarray(delta_Rtij_4_4))[:3]
    M_est, err_est = good_corr_eval_nondecompose(x1, x2, np.asarray(E_est).astype(np.float64), dinv, K, None)
    _, err_gt = good_corr_eval_nondecompose(x1, x2, np.asarray(E_gt).astype(np.float64), dinv, K, None)
    return {"err_est": np.array(err_est), "epi_est": epi_distance_np(F_est, x1, x2)[0], "err_gt": np.array(err_gt),
            "epi_gt": epi_distance_np(F_gt, x1, x2)[0], "M_est": M_est}


# --------------------------------------------------------------------------------------
# a14: E projection  (utils_F.py:455-462, Train_model_pipeline.py:954-964)
# --------------------------------------------------------------------------------------
def F_to_E(F: Tensor, K: Tensor) -> Tensor:
    E = K.T @ F @ K
    U, S, Vh = torch.linalg.svd(E)
    return U @ torch.diag(torch.tensor([1.0, 1.0, 0.0], dtype=F.dtype)) @ Vh


def E_to_F(E: Tensor, K: Tensor) -> Tensor:
    Ki = torch.linalg.inv(K)
    return Ki.transpose(-1, -2) @ E @ Ki


# --------------------------------------------------------------------------------------
# a16: epipolar metrics  (utils_F.py:291-361)
# --------------------------------------------------------------------------------------
def _homo(x: Tensor) -> Tensor:
    return torch.cat((x, torch.ones(*x.shape[:-1], 1, dtype=x.dtype)), -1)


def _epi_terms(F: Tensor, X: Tensor, Y: Tensor, if_homo: bool):
    if not if_homo:
        X, Y = _homo(X), _homo(Y)
    Fx1 = X @ F.transpose(-1, -2)  # rows F x
    Fx2 = Y @ F  # rows F^T y
    num = (Y * Fx1).sum(-1)  # y^T F x
    return num, Fx1, Fx2


def sampson_dist(F, X, Y, if_homo=False):
    num, a, b = _epi_terms(F, X, Y, if_homo)
    return num**2 / (a[..., 0] ** 2 + a[..., 1] ** 2 + b[..., 0] ** 2 + b[..., 1] ** 2)


def sym_epi_dist(F, X, Y, if_homo=False, clamp_at=None):
    """Squared symmetric epipolar distance; the 1e-10 guard exists only in the batched branch (:329)."""
    num, a, b = _epi_terms(F, X, Y, if_homo)
    eps = 1e-10 if X.dim() == 3 else 0.0
    e = num**2 * (1.0 / (a[..., 0] ** 2 + a[..., 1] ** 2 + eps) + 1.0 / (b[..., 0] ** 2 + b[..., 1] ** 2 + eps))
    return e if clamp_at is None else torch.clamp(e, max=clamp_at)


def epi_distance(F, X, Y, if_homo=False):
    num, a, b = _epi_terms(F, X, Y, if_homo)
    d1 = num.abs() / torch.sqrt(a[..., 0] ** 2 + a[..., 1] ** 2)
    d2 = num.abs() / torch.sqrt(b[..., 0] ** 2 + b[..., 1] ** 2)
    return (d1 + d2) / 2.0, d1, d2


# --------------------------------------------------------------------------------------
# a17: textbook normalised 8-point on K^-1 points  (utils_F.py:15-37, 104-155, 223-275)
# --------------------------------------------------------------------------------------
def _normalize_xy(X: Tensor) -> Tuple[Tensor, Tensor]:
    """Hartley with sqrt(2) (np.sqrt(2), :23) on [N,2]; returns ([N,2], T)."""
    m = X.mean(0)
    s = math.sqrt(2.0) / (X - m).norm(dim=1).mean()
    T = torch.tensor([[s, 0, -s * m[0]], [0, s, -s * m[1]], [0, 0, 1]], dtype=X.dtype)
    Xn = _homo(X) @ T.T
    return Xn[:, :2] / (Xn[:, 2:3] + 1e-10), T  # _de_homo adds 1e-10 (utils_misc.py:73-77)


def _eight_point_rows(X: Tensor, Y: Tensor) -> Tensor:
    return torch.stack(
        (Y[:, 0] * X[:, 0], Y[:, 0] * X[:, 1], Y[:, 0], Y[:, 1] * X[:, 0], Y[:, 1] * X[:, 1], Y[:, 1], X[:, 0], X[:, 1], torch.ones_like(X[:, 0])),
        1,
    )


def F_from_XY(X: Tensor, Y: Tensor, W: Optional[Tensor] = None, normalize: bool = True) -> Tensor:
    if normalize:
        X, T1 = _normalize_xy(X)
        Y, T2 = _normalize_xy(Y)
    XX = _eight_point_rows(X, Y)
    if W is not None:
        XX = W @ XX
    f = torch.linalg.svd(XX, full_matrices=False)[2][-1]
    U, S, Vh = torch.linalg.svd(f.reshape(3, 3))
    S = S.clone()
    S[2] = 0
    Fm = U @ torch.diag(S) @ Vh
    return T2.T @ Fm @ T1 if normalize else Fm


def E_from_XY(X: Tensor, Y: Tensor, K: Tensor, W: Optional[Tensor] = None, if_normzliedK: bool = False, normalize: bool = True) -> Tensor:
    if not if_normzliedK:
        Ki = torch.linalg.inv(K)
        X = _homo(X) @ Ki.T
        X = X[:, :2] / (X[:, 2:3] + 1e-10)
        Y = _homo(Y) @ Ki.T
        Y = Y[:, :2] / (Y[:, 2:3] + 1e-10)
    if normalize:
        X, T1 = _normalize_xy(X)
        Y, T2 = _normalize_xy(Y)
    XX = _eight_point_rows(X, Y)
    if W is not None:
        XX = W @ XX
    f = torch.linalg.svd(XX, full_matrices=False)[2][-1]
    U, S, Vh = torch.linalg.svd(f.reshape(3, 3))
    E = U @ torch.diag(torch.tensor([1.0, 1.0, 0.0], dtype=X.dtype)) @ Vh  # singular values forced (:148-149)
    return T2.T @ E @ T1 if normalize else E


# --------------------------------------------------------------------------------------
# f-2: validation summary  (train_good_utils.py:758-856 write_metrics_summary)
# --------------------------------------------------------------------------------------
METRIC_THS = [0.0, 0.01, 0.03, 0.05, 0.1, 0.3, 0.5, 1.0, 2.0, 5.0, 10.0, 90.0, 180.0]


def metrics_summary_np(epi_est: np.ndarray, epi_gt: np.ndarray, err_q: np.ndarray, err_t: np.ndarray) -> Dict[str, object]:
    """The scalars write_metrics_summary logs for one experiment tag: inlier ratios of the epipolar distances at 0.1 / 1.0
    (:776-777), F1 of (est < th) against (gt < th) (:788-797; sklearn's binary f1 = 2TP / (2TP + FP + FN)), medians (:799-805)
    and maxima (:811-816) of the pose errors, cumulative np.histogram ratios over METRIC_THS (:829-853)."""
    epi_est, epi_gt = np.asarray(epi_est).flatten(), np.asarray(epi_gt).flatten()
    err_q, err_t = np.asarray(err_q).flatten(), np.asarray(err_t).flatten()

    def f1(y_true, y_pred):
        tp = np.sum(y_true & y_pred); fp = np.sum(~y_true & y_pred); fn = np.sum(y_true & ~y_pred)
        return float(2 * tp / (2 * tp + fp + fn)) if (2 * tp + fp + fn) > 0 else 0.0

    n = float(err_q.shape[0])
    return {"ratio_0.1": float(np.sum(epi_est < 0.1) / epi_est.shape[0]), "ratio_1": float(np.sum(epi_est < 1.0) / epi_est.shape[0]),
            "F1_0.1": f1(epi_gt < 0.1, epi_est < 0.1), "F1_1": f1(epi_gt < 1.0, epi_est < 1.0),
            "median_err_q": float(np.median(err_q)), "median_err_t": float(np.median(err_t)),
            "max_err_q": float(np.amax(err_q)), "max_err_t": float(np.amax(err_t)),
            "ratio_q": np.cumsum(np.histogram(err_q, METRIC_THS)[0].astype(float) / n).tolist(),
            "ratio_t": np.cumsum(np.histogram(err_t, METRIC_THS)[0].astype(float) / n).tolist()}


# --------------------------------------------------------------------------------------
# f-4: DSAC hypothesis loop  (dsac_tools/dsac.py:94-197)
# --------------------------------------------------------------------------------------
def dsac_scores(X: Tensor, Y: Tensor, K: Tensor, hyps: int, inlier_thresh: float, inlier_beta: float, idx_list):
    """The reference's loop, one hypothesis at a time (:138-176): E from the 10 sampled correspondences (_E_from_XY, :49),
    Sampson distances of F = K^-T E K^-1 in pixel space (:68; the reference names an undefined `E_to_F` there), soft inlier
    count 1 - sigmoid(beta (d - thresh)) (:74), refinement by _E_from_XY with W = diag(sqrt(dists)) (:92), and the
    per-correspondence average score (:164-165, :197).  `idx_list` = the minimal sets (random.sample draws, :45), passed in
    so that both sides use the same ones.  **Parity unpinned**: the reference class is not runnable as is (undefined name);
    this restates its evident algorithm.  Returns (N_scores / (N_counts + 1e-10) [N,1], per-hypothesis scores, refined Es)."""
    N = X.shape[0]
    Ki = torch.linalg.inv(K)
    n_scores, n_counts = torch.zeros(N, 1, dtype=X.dtype), torch.zeros(N, 1, dtype=X.dtype)
    scores, refined = [], []
    for h in range(hyps):
        idx = idx_list[h]
        E = E_from_XY(X[idx], Y[idx], K)
        d = sampson_dist(Ki.T @ E @ Ki, X, Y)
        dists = 1 - torch.sigmoid(inlier_beta * (d - inlier_thresh))
        score = dists.sum()
        refined.append(E_from_XY(X, Y, K, W=torch.diag(torch.sqrt(dists))))
        scores.append(score)
        n_scores[idx] += score
        n_counts[idx] += 1.0
    return n_scores / (n_counts + 1e-10), torch.stack(scores), torch.stack(refined)


# --------------------------------------------------------------------------------------
# helpers for tests / bench (not reference functions)
# --------------------------------------------------------------------------------------
def align_sign(A: Tensor, ref: Tensor) -> Tensor:
    """Flip each [.., 3,3] / [.., n] item of A so that <A, ref> >= 0 (SVD sign gauge)."""
    dims = tuple(range(1, A.dim()))
    s = torch.sign((A * ref).sum(dim=dims, keepdim=True))
    s = torch.where(s == 0, torch.ones_like(s), s)
    return A * s


def unit_frobenius(A: Tensor) -> Tensor:
    return A / A.flatten(1).norm(dim=1).reshape(-1, *([1] * (A.dim() - 1)))


def hot_path_step(scene: Dict[str, Tensor], image_size, depth: int, clamp_at: float, qt: bool, mode: str,
                  clamp_q: float = 0.1, clamp_t: float = 0.5, balance_q: float = 1.0, balance_t: float = 0.1,
                  backward: bool = True, balance_F: float = 1.0):
    """One pass of the hot path the way bench.py times it: depth fits with fixed per-layer logits,
    F-loss, E-from-F, pose loss, and (optionally) backward to the logits.  ``mode='loop'`` is the
    reference-shaped per-sample structure (DeepFNet.py:232-240, train_good_utils.py:106-239)."""
    logits = scene["logits_layers"][:depth].clone().requires_grad_(backward)
    outs = deepf_forward(scene["matches_xy_ori"], image_size, depth, logits_layers=logits, mode=mode)
    losses, E_ests, F_ests, E_layers = f_loss(outs, scene["pts1_virt_ori"], scene["pts2_virt_ori"], scene["Ks"], depth, clamp_at)
    # Train_model_pipeline.py:580-587 drops the F-loss from the objective when if_qt_loss (`loss += loss_F*balance_F` is
    # commented out): that is balance_F = 0; the solver-only benchmark step keeps both terms (balance_F = 1)
    loss = losses["loss_F"] * balance_F
    pose = None
    if qt:
        pose = rt_loss(E_layers, scene["delta_Rtijs_4_4"], scene["qs_cam"], scene["ts_cam"])
        loss = loss + qt_training_loss(pose["q_l2"], pose["t_l2"], clamp_q, clamp_t, balance_q, balance_t)
    if backward:
        loss.backward()
    return {"loss": loss.detach(), "outs": outs, "losses": losses, "E_layers": E_layers, "pose": pose,
            "grad_logits": logits.grad if backward else None}


# --------------------------------------------------------------------------------------
# f-3: match construction  (train_good_utils.py:649-724 get_matches_from_SP, second half)
# --------------------------------------------------------------------------------------
# nn_match_two_way is NOT in /root/reference: it is ``PointTracker.nn_match_two_way`` of the ``superpoint`` package
# (https://github.com/eric-yyjau/pytorch-superpoint, models/model_wrap.py; installed unpinned from git,
# README.md:37-40), itself the published routine of Magic Leap's SuperPoint demo.  "Parity unpinned" for this one
# function: the restatement below follows the published algorithm and is anchored on the reference's call site
# (train_good_utils.py:687-691: descriptors transposed to [D,N], ``nn_thresh`` from the tracker) and on the
# reference's own consumer of its [3,n] result (:693-716), which tests/golden/make_golden_matching.py runs unmodified.
def nn_match_two_way(desc1: np.ndarray, desc2: np.ndarray, nn_thresh: float) -> np.ndarray:
    """desc1 [D,N1], desc2 [D,N2] unit-norm columns -> [3,n]: rows = index in 1, index in 2, L2 distance."""
    assert desc1.shape[0] == desc2.shape[0]
    if desc1.shape[1] == 0 or desc2.shape[1] == 0:
        return np.zeros((3, 0))
    if nn_thresh < 0.0:
        raise ValueError("'nn_thresh' should be non-negative")
    dmat = np.dot(desc1.T, desc2)
    dmat = np.sqrt(2 - 2 * np.clip(dmat, -1, 1))       # L2 distance of unit vectors
    idx = np.argmin(dmat, axis=1)                      # nearest neighbour in 2 of every point of 1 (first on ties)
    scores = dmat[np.arange(dmat.shape[0]), idx]
    keep = scores < nn_thresh
    idx2 = np.argmin(dmat, axis=0)                     # nearest neighbour in 1 of every point of 2
    keep = np.logical_and(keep, np.arange(len(idx)) == idx2[idx])
    idx, scores = idx[keep], scores[keep]
    matches = np.zeros((3, int(keep.sum())))
    matches[0, :] = np.arange(desc1.shape[1])[keep]
    matches[1, :] = idx
    matches[2, :] = scores
    return matches


def crop_or_pad_choice(in_num_points: int, out_num_points: int, shuffle: bool = False) -> np.ndarray:
    """Indices that crop or pad a set to a fixed size (dsac_tools/utils_misc.py:139-161); draws from np.random's
    global state exactly like the reference (one permutation when shuffling, one choice() when padding)."""
    choice = np.random.permutation(in_num_points) if shuffle else np.arange(in_num_points)
    assert out_num_points > 0
    if in_num_points >= out_num_points:
        return choice[:out_num_points]
    pad = np.random.choice(choice, out_num_points - in_num_points, replace=True)
    return np.concatenate([choice, pad])


def matches_from_sp_outputs(xs_SP, deses_SP, reses_SP, nn_thresh: float, out_num_points: int = 1000):
    """The per-pair loop of get_matches_from_SP after the SuperPoint front-end (train_good_utils.py:679-724):
    xs_SP / reses_SP: two tensors [B,N,2] (integer keypoints, sub-pixel offsets), deses_SP: two tensors [B,N,D]."""
    f = lambda t: t.detach().cpu().numpy()
    B = xs_SP[0].shape[0]
    xs_list, off_list, q_list, n_list = [], [], [], []
    for b in range(B):
        m = nn_match_two_way(f(deses_SP[0][b]).transpose(), f(deses_SP[1][b]).transpose(), nn_thresh)
        choice = crop_or_pad_choice(m.shape[1], out_num_points, shuffle=True)
        n_list.append(m.shape[1])
        m = m[:, choice]
        i1, i2 = m[0].astype(int), m[1].astype(int)
        xs_list.append(torch.cat((xs_SP[0][b][i1], xs_SP[1][b][i2]), dim=1))
        off_list.append(torch.cat((reses_SP[0][b][i1], reses_SP[1][b][i2]), dim=1))
        q_list.append(m[2:3].transpose())
    return {"xs": torch.stack(xs_list), "offsets": torch.stack(off_list),
            "quality": torch.from_numpy(np.stack(q_list)).float(), "num_matches": torch.tensor(n_list),
            "xs_SP": [x + r for x, r in zip(xs_SP, reses_SP)]}
