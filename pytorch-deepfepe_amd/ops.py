"""torch.autograd wrappers over the C ABI (device memory and streams come from PyTorch-ROCm; the
arithmetic is entirely in libdfepe_hip.so)."""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import torch

from . import _lib

Tensor = torch.Tensor


def _ptr(t: Optional[Tensor]):
    return None if t is None else t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _prep(t: Tensor, name: str) -> Tensor:
    if not t.is_cuda:
        raise _lib.DfepeError(f"{name} must live on the GPU (this package has no CPU path)")
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def save_floats() -> int:
    return _lib.lib().dfepe_save_floats()


# ------------------------------------------------------------------------------------------------
# weighted 8-point fit
# ------------------------------------------------------------------------------------------------
def w8pt_forward(pts1: Tensor, pts2: Optional[Tensor], weights: Tensor, raw: bool, image_w: float, image_h: float,
                 clamp_at: float, want_epi: bool, want_save: bool):
    """Raw (non-differentiable) launch.  weights [B,N]; returns F [B,3,3], residual [B,N], epi [B,N]|None, save|None."""
    L = _lib.lib()
    B, N = weights.shape
    F = torch.empty(B, 3, 3, device=weights.device, dtype=torch.float32)
    residual = torch.empty(B, N, device=weights.device, dtype=torch.float32)
    epi = torch.empty(B, N, device=weights.device, dtype=torch.float32) if want_epi else None
    save = torch.empty(B, _lib.lib().dfepe_save_floats(), device=weights.device, dtype=torch.float32) if want_save else None
    with torch.cuda.device(weights.device):
        rc = L.dfepe_w8pt_fwd(_ptr(pts1), _ptr(pts2), _ptr(weights), B, N, _lib.W8PT_RAW_MATCHES if raw else 0,
                              float(image_w), float(image_h), float(clamp_at), _ptr(F), _ptr(residual), _ptr(epi),
                              _ptr(save), _stream())
    _lib.check(rc, "dfepe_w8pt_fwd")
    return F, residual, epi, save


class _W8ptFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pts1, pts2, weights, raw, image_w, image_h, clamp_at, want_epi):
        F, residual, epi, save = w8pt_forward(pts1, pts2, weights, raw, image_w, image_h, clamp_at, want_epi,
                                              want_save=True)
        ctx.save_for_backward(pts1, pts2 if pts2 is not None else pts1.new_empty(0), weights, save, F)
        ctx.cfg = (raw, image_w, image_h, clamp_at, want_epi)
        if want_epi:
            return F, residual, epi
        return F, residual

    @staticmethod
    def backward(ctx, gF, gRes, gEpi=None):
        pts1, pts2, weights, save, F = ctx.saved_tensors
        raw, image_w, image_h, clamp_at, want_epi = ctx.cfg
        L = _lib.lib()
        B, N = weights.shape
        gW = torch.empty_like(weights)
        gF = None if gF is None else gF.contiguous().float()
        gRes = None if gRes is None else gRes.contiguous().float()
        gEpi = None if (gEpi is None or not want_epi) else gEpi.contiguous().float()
        with torch.cuda.device(weights.device):
            rc = L.dfepe_w8pt_bwd(_ptr(pts1), _ptr(pts2) if pts2.numel() else None, _ptr(weights), B, N,
                                  _lib.W8PT_RAW_MATCHES if raw else 0, float(image_w), float(image_h), float(clamp_at),
                                  _ptr(save), _ptr(F), _ptr(gF), _ptr(gRes), _ptr(gEpi), _ptr(gW), _stream())
        _lib.check(rc, "dfepe_w8pt_bwd")
        return None, None, gW, None, None, None, None, None


def w8pt(pts1: Tensor, pts2: Tensor, weights: Tensor, clamp_at: float = 0.5, want_epi: bool = False):
    """Differentiable (w.r.t. weights) fit on homogeneous points [B,N,3]; weights [B,N] or [B,1,N]."""
    pts1, pts2 = _prep(pts1, "pts1"), _prep(pts2, "pts2")
    w = _prep(weights.reshape(weights.shape[0], -1), "weights")
    return _W8ptFunction.apply(pts1, pts2, w, False, 0.0, 0.0, clamp_at, want_epi)


def w8pt_raw(matches: Tensor, weights: Tensor, image_w: float, image_h: float, clamp_at: float = 0.5,
             want_epi: bool = True):
    """Differentiable fit straight from pixel matches [B,N,4] (image-size normalisation fused)."""
    m = _prep(matches, "matches")
    w = _prep(weights.reshape(weights.shape[0], -1), "weights")
    return _W8ptFunction.apply(m, None, w, True, float(image_w), float(image_h), clamp_at, want_epi)


# ------------------------------------------------------------------------------------------------
# F-loss (virtual-point epipolar residual sums per layer and pair) and E-from-F
# ------------------------------------------------------------------------------------------------
def _t_arg(T: Tensor, B: int):
    """T given as [3,3] / [1,3,3] (shared) or [B,3,3]; expanded (stride-0) views are collapsed to shared."""
    if T.dim() == 2:
        return _prep(T, "T"), 0
    if T.shape[0] == 1 or (T.stride(0) == 0):
        return _prep(T[0], "T"), 0
    assert T.shape[0] == B
    return _prep(T, "T"), 9


class _FlossFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, F_layers, T1, T2, K, virt1, virt2, clamp_at):
        lib = _lib.lib()
        L, B = F_layers.shape[0], F_layers.shape[1]
        M = virt1.shape[1]
        T1c, st1 = _t_arg(T1, B)
        T2c, st2 = _t_arg(T2, B)
        if st1 != st2:
            T1c, T2c, st1 = T1c.expand(B, 3, 3).contiguous(), T2c.expand(B, 3, 3).contiguous(), 9
        loss_sum = torch.empty(L, B, device=F_layers.device, dtype=torch.float32)
        E_layers = torch.empty(L, B, 3, 3, device=F_layers.device, dtype=torch.float32)
        with torch.cuda.device(F_layers.device):
            rc = lib.dfepe_floss_fwd(_ptr(F_layers), L, B, _ptr(T1c), _ptr(T2c), st1, _ptr(K), _ptr(virt1), _ptr(virt2), M,
                                     float(clamp_at), _ptr(loss_sum), _ptr(E_layers), _stream())
        _lib.check(rc, "dfepe_floss_fwd")
        ctx.save_for_backward(F_layers, T1c, T2c, K, virt1, virt2)
        ctx.cfg = (st1, float(clamp_at))
        return loss_sum, E_layers

    @staticmethod
    def backward(ctx, g_loss_sum, g_E):
        F_layers, T1c, T2c, K, virt1, virt2 = ctx.saved_tensors
        st, clamp_at = ctx.cfg
        lib = _lib.lib()
        L, B = F_layers.shape[0], F_layers.shape[1]
        gF = torch.empty_like(F_layers)
        g_loss_sum = None if g_loss_sum is None else g_loss_sum.contiguous().float()
        g_E = None if g_E is None else g_E.contiguous().float()
        with torch.cuda.device(F_layers.device):
            rc = lib.dfepe_floss_bwd(_ptr(F_layers), L, B, _ptr(T1c), _ptr(T2c), st, _ptr(K), _ptr(virt1), _ptr(virt2),
                                     virt1.shape[1], clamp_at, _ptr(g_loss_sum), _ptr(g_E), _ptr(gF), _stream())
        _lib.check(rc, "dfepe_floss_bwd")
        return gF, None, None, None, None, None, None


def floss(F_layers: Tensor, T1: Tensor, T2: Tensor, K: Tensor, virt1: Tensor, virt2: Tensor, clamp_at: float):
    """F_layers [L,B,3,3] -> (loss_sum [L,B] = sum over virtual points of the clamped residual, E_layers [L,B,3,3])."""
    return _FlossFunction.apply(_prep(F_layers, "F_layers"), T1, T2, _prep(K, "K"), _prep(virt1, "virt1"),
                                _prep(virt2, "virt2"), clamp_at)


# ------------------------------------------------------------------------------------------------
# pose loss
# ------------------------------------------------------------------------------------------------
class _PoseFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, E_layers, q_gt, t_gt, R_gt):
        lib = _lib.lib()
        L, B = E_layers.shape[0], E_layers.shape[1]
        dev = E_layers.device
        q_l2 = torch.empty(L, B, device=dev, dtype=torch.float32)
        t_l2 = torch.empty(L, B, device=dev, dtype=torch.float32)
        R_deg = torch.empty(L, B, device=dev, dtype=torch.float32)
        t_deg = torch.empty(L, B, device=dev, dtype=torch.float32)
        sel = torch.empty(L, B, device=dev, dtype=torch.int32)
        with torch.cuda.device(dev):
            rc = lib.dfepe_pose_fwd(_ptr(E_layers), L, B, _ptr(q_gt), _ptr(t_gt), _ptr(R_gt), _ptr(q_l2), _ptr(t_l2),
                                    _ptr(R_deg), _ptr(t_deg), _ptr(sel), _stream())
        _lib.check(rc, "dfepe_pose_fwd")
        ctx.save_for_backward(E_layers, q_gt, t_gt)
        ctx.mark_non_differentiable(R_deg, t_deg, sel)
        return q_l2, t_l2, R_deg, t_deg, sel

    @staticmethod
    def backward(ctx, g_q, g_t, _a, _b, _c):
        E_layers, q_gt, t_gt = ctx.saved_tensors
        lib = _lib.lib()
        L, B = E_layers.shape[0], E_layers.shape[1]
        gE = torch.empty_like(E_layers)
        g_q = None if g_q is None else g_q.contiguous().float()
        g_t = None if g_t is None else g_t.contiguous().float()
        with torch.cuda.device(E_layers.device):
            rc = lib.dfepe_pose_bwd(_ptr(E_layers), L, B, _ptr(q_gt), _ptr(t_gt), _ptr(g_q), _ptr(g_t), _ptr(gE), _stream())
        _lib.check(rc, "dfepe_pose_bwd")
        return gE, None, None, None


def pose_errors(E_layers: Tensor, q_gt: Tensor, t_gt: Tensor, R_gt: Tensor):
    """E_layers [L,B,3,3]; q_gt [B,4(,1)], t_gt [B,3(,1)], R_gt [B,3,3] (camera motion).
    Returns q_l2, t_l2 (differentiable w.r.t. E), R_deg, t_deg, sel — all [L,B]."""
    B = E_layers.shape[1]
    return _PoseFunction.apply(_prep(E_layers, "E_layers"), _prep(q_gt.reshape(B, 4), "q_gt"),
                               _prep(t_gt.reshape(B, 3), "t_gt"), _prep(R_gt.reshape(B, 3, 3), "R_gt"))
