"""The hot path as one callable: depth x (softmax -> weighted 8-point fit [+ in-loop epipolar residual]),
F-loss on the virtual points, E-from-F, pose loss — forward and backward to the logits.

This is the "solver-only" step of SURVEY.md §8d (C3/C4): the per-layer logits are given tensors (in the
full model they come from the stock-PyTorch ErrorEstimator, which is outside the hot path).  The same
function is what bench.py times and what the parity tests compare with the CPU oracle's hot_path_step.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import torch

from . import ops

Tensor = torch.Tensor


def hot_path_forward(matches: Tensor, logits_layers: Tensor, Ks: Tensor, virt1: Tensor, virt2: Tensor,
                     q_gt: Tensor, t_gt: Tensor, R_gt: Tensor, image_size: Sequence[int], clamp_at: float = 0.02,
                     qt: bool = True, clamp_q: float = 0.1, clamp_t: float = 0.5, balance_q: float = 1.0,
                     balance_t: float = 0.1, hw_T: Optional[Tensor] = None) -> Dict[str, Tensor]:
    """matches [B,N,4] pixels, logits_layers [L,B,N]; returns dict with the loss (local batch mean) and
    every intermediate the reference exposes (F per layer, residuals, in-loop epipolar residuals, E per layer,
    per-pair F-loss sums, pose errors and angular metrics)."""
    L, B, N = logits_layers.shape
    H, W = float(image_size[0]), float(image_size[1])
    Fs, residuals, epis, weights = [], [], [], []
    for l in range(L):
        w = torch.softmax(logits_layers[l], dim=1)
        F, res, epi = ops.w8pt_raw(matches, w, W, H, clamp_at=0.5, want_epi=True)
        Fs.append(F)
        residuals.append(res)
        epis.append(epi)
        weights.append(w)
    F_layers = torch.stack(Fs)  # [L,B,3,3]
    if hw_T is None:
        hw_T = torch.tensor([[2.0 / W, 0.0, -1.0], [0.0, 2.0 / H, -1.0], [0.0, 0.0, 1.0]], device=matches.device)
    loss_sum, E_layers = ops.floss(F_layers, hw_T, hw_T, Ks, virt1, virt2, clamp_at)
    M = virt1.shape[1]
    loss_layers = loss_sum.sum(dim=1) / float(B * M)  # losses.mean() per layer
    loss_F = loss_layers.mean()
    out = {"F_layers": F_layers, "residual_layers": residuals, "epi_res_layers": epis, "weights_layers": weights,
           "E_layers": E_layers, "loss_sum": loss_sum, "loss_layers": loss_layers, "loss_F": loss_F}
    loss = loss_F
    if qt:
        q_l2, t_l2, R_deg, t_deg, sel = ops.pose_errors(E_layers, q_gt, t_gt, R_gt)
        loss_qt = torch.clamp(q_l2, 0.0, clamp_q).mean() * balance_q + torch.clamp(t_l2, 0.0, clamp_t).mean() * balance_t
        out.update({"q_l2": q_l2, "t_l2": t_l2, "R_deg": R_deg, "t_deg": t_deg, "sel": sel, "loss_qt": loss_qt})
        loss = loss + loss_qt
    out["loss"] = loss
    return out


def scene_to_device(scene: Dict[str, Tensor], device) -> Dict[str, Tensor]:
    """Move a synth.make_scene() batch to the GPU and derive the camera-motion rotation the pose loss needs
    (R_gt = inv(delta)[:3,:3] = R^T, train_good_utils.py:134,170)."""
    d = {k: v.to(device) for k, v in scene.items()}
    d["R_gt"] = d["delta_Rtijs_4_4"][:, :3, :3].transpose(1, 2).contiguous()
    return d


def hot_path_step(scene: Dict[str, Tensor], image_size: Sequence[int], depth: int, clamp_at: float = 0.02,
                  qt: bool = True, backward: bool = True, **kw) -> Dict[str, Tensor]:
    logits = scene["logits_layers"][:depth].detach().clone().requires_grad_(backward)
    out = hot_path_forward(scene["matches_xy_ori"], logits, scene["Ks"], scene["pts1_virt_ori"], scene["pts2_virt_ori"],
                           scene["qs_cam"], scene["ts_cam"], scene["R_gt"], image_size, clamp_at, qt, **kw)
    if backward:
        out["loss"].backward()
        out["grad_logits"] = logits.grad
    return out
