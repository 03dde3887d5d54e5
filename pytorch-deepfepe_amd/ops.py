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
