"""torch.autograd wrappers over the C ABI (device memory and streams come from PyTorch-ROCm; the
arithmetic is entirely in libdfepe_hip.so)."""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib

Tensor = torch.Tensor


def _ptr(t: Optional[Tensor]):
    return None if t is None else t.data_ptr()


# Host cost of a launch (scripts/eager_profile.py: the reference's eager call sequence is host-bound, ~50 launches per step):
# torch.cuda.current_stream().cuda_stream builds a Stream object per call (~10 us) and torch.cuda.device() is a python context
# manager doing two device queries (~8 us) -- the raw C entry points below do the same in well under a microsecond each.
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream() -> int:
    """The current HIP stream of the current device, as the integer handle the C ABI takes."""
    if _raw_stream is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def _on(dev):
    """`with _on(t.device):` = `with torch.cuda.device(t.device):` without the context-manager cost when that device is already
    the current one (every call of a single-GPU process, and of a rank that called torch.cuda.set_device)."""
    if _cur_device is not None and dev.index is not None and dev.index == _cur_device():
        return _NO_GUARD
    return torch.cuda.device(dev)


def _prep(t: Tensor, name: str) -> Tensor:
    if not t.is_cuda:
        raise _lib.DfepeError(f"{name} must live on the GPU (this package has no CPU path)")
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _shape(t: Tensor, name: str, *dims) -> None:
    """Raise ValueError unless t has the given shape (None = any size): the kernels index raw pointers, so a wrong
    layout would be a silent out-of-bounds read, not an exception."""
    if t.dim() != len(dims) or any(d is not None and t.shape[i] != d for i, d in enumerate(dims)):
        want = "[" + ",".join("?" if d is None else str(d) for d in dims) + "]"
        raise ValueError(f"{name} must have shape {want}, got {tuple(t.shape)}")


def save_floats() -> int:
    return _lib.lib().dfepe_save_floats()


# ------------------------------------------------------------------------------------------------
# per-layer stacks without copies.  The reference's API speaks in python lists of per-layer tensors (out_layers,
# E_ests_layers, q_l2_error_layers_list ...; DeepFNet.py:534-548, train_good_utils.py:272-293) and glues them with
# torch.stack; the kernels want one [L, ...] buffer.  These helpers let both views share the memory: the rows of a stack are
# handed out as separate autograd outputs (no SelectBackward that would zero-fill and copy a full stack per row in the
# backward), and a list whose tensors turn out to be the rows of one buffer is re-assembled without a copy kernel.
# ------------------------------------------------------------------------------------------------
def row_of(stack: Tensor, l: int) -> Tensor:
    """Row l of a contiguous [L, ...] buffer as a tensor of its own over the same memory (not an autograd view of it)."""
    n = stack[0].numel()
    t = torch.empty(0, dtype=stack.dtype, device=stack.device)
    t.set_(stack.untyped_storage(), stack.storage_offset() + l * n, tuple(stack.shape[1:]), None)
    return t


def alias_rows(rows) -> Optional[Tensor]:
    """[len(rows), *shape] tensor over the memory of ``rows`` when they are equally shaped contiguous tensors lying at a uniform
    distance from each other in one buffer, in order (back to back: the result is contiguous; further apart -- e.g. the same
    channel of consecutive per-layer buffers -- it is a strided view); None otherwise (the caller then copies)."""
    if len(rows) == 0 or rows[0] is None:
        return None
    r0 = rows[0]
    n = r0.numel()
    if n == 0:
        return None
    base = r0.untyped_storage().data_ptr()
    step = n if len(rows) == 1 else (rows[1].storage_offset() - r0.storage_offset() if rows[1] is not None else -1)
    if step < n:
        return None
    for l, r in enumerate(rows):
        if (r is None or r.shape != r0.shape or r.dtype != r0.dtype or r.device != r0.device or not r.is_contiguous()
                or r.untyped_storage().data_ptr() != base or r.storage_offset() != r0.storage_offset() + l * step):
            return None
    t = torch.empty(0, dtype=r0.dtype, device=r0.device)
    t.set_(r0.untyped_storage(), r0.storage_offset(), (len(rows),) + tuple(r0.shape), (step,) + tuple(r0.stride()))
    return t


class _StackRowsFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *rows):
        ctx.set_materialize_grads(False)  # an unused stack must not send rows of zeros back (a zero-fill, and kernels that then
        ctx.n = len(rows)                 # run their gradient paths for nothing)
        a = alias_rows(rows)
        if a is None:
            return torch.stack(rows)
        # the result is a tensor of its own over the rows' memory, so autograd's version counters of the rows do not cover it.
        # The rows are therefore saved: an in-place write to any of them between here and the backward makes the saved-tensor
        # check below raise ("modified by an inplace operation") instead of silently corrupting what a kernel saved of the stack
        ctx.save_for_backward(*rows)
        return a

    @staticmethod
    def backward(ctx, g):
        ctx.saved_tensors  # the version check of every aliased row (no-op when the rows were copied)
        return (None,) * ctx.n if g is None else tuple(g.unbind(0))


class _UnstackRowsFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.meta = (x.shape, x.dtype, x.device)
        ctx.set_materialize_grads(False)
        x = x if x.is_contiguous() else x.contiguous()
        # real views of x: they share x's version counter, and autograd forbids in-place writes to the outputs of a function that
        # returns several views ("...is a view and is being modified inplace"), so a caller cannot silently overwrite a row of a
        # stack that some kernel saved for its backward.  Their grad_fn is still THIS node (no SelectBackward per row).
        return tuple(x[l] for l in range(x.shape[0]))

    @staticmethod
    def backward(ctx, *gs):
        shape, dtype, dev = ctx.meta
        if all(g is None for g in gs):
            return None
        a = alias_rows(gs)  # torch.stack's backward hands out the rows of one gradient buffer: taken as it is
        if a is not None:
            return a
        return torch.stack([torch.zeros(shape[1:], dtype=dtype, device=dev) if g is None else g for g in gs])


def stack_rows(rows) -> Tensor:
    """torch.stack(rows) without the copy when the rows already are the consecutive slices of one buffer."""
    rows = list(rows)
    return _StackRowsFunction.apply(*rows)


def unstack_rows(x: Tensor):
    """The rows of x [L, ...] as a tuple of L views of x, each an autograd output of its own (and, like the outputs of
    torch.unbind, not writable in place while gradients are tracked)."""
    return _UnstackRowsFunction.apply(x)


# ------------------------------------------------------------------------------------------------
# weighted 8-point fit
# ------------------------------------------------------------------------------------------------
def _flags(raw: bool, logits: bool, row_per_pair: bool = False, extra: int = 0) -> int:
    return ((_lib.W8PT_RAW_MATCHES if raw else 0) | (_lib.W8PT_LOGITS if logits else 0) |
            (_lib.W8PT_ROW_PER_PAIR if row_per_pair else 0) | int(extra))


def w8pt_forward(pts1: Tensor, pts2: Optional[Tensor], weights: Tensor, raw: bool, image_w: float, image_h: float,
                 clamp_at: float, want_epi: bool, want_save: bool, logits: bool = False, F_out: Optional[Tensor] = None,
                 row_per_pair: bool = False, extra_flags: int = 0, dst: Optional[dict] = None):
    """Raw (non-differentiable) launch.  weights (or logits when ``logits``) [B,N].
    Returns F [B,3,3], residual [B,N], epi [B,N]|None, save|None, weights_out [B,N]|None.  ``F_out`` lets the caller
    provide the destination (e.g. one [B,3,3] slice of a per-layer stack).  ``row_per_pair`` forces one 16-lane row per pair
    where a cooperative workgroup per pair (N > 128 at small batch) would run: same function, same ``save`` record.
    ``dst``: {"F" | "residual" | "epi" | "weights": (stack, l)} writes that output into row l of the caller's per-layer stack."""
    L = _lib.lib()
    B, N = weights.shape
    dev = weights.device
    dst = dst or {}

    def out(name, shape, want=True):
        if not want:
            return None
        if name in dst:
            stack, l = dst[name]
            if tuple(stack.shape[1:]) != tuple(shape) or stack.dtype != torch.float32 or not stack.is_contiguous():
                raise ValueError(f"dst[{name!r}] must be a contiguous float32 stack of {tuple(shape)} rows, got {tuple(stack.shape)}")
            return row_of(stack, l)
        return torch.empty(*shape, device=dev, dtype=torch.float32)

    F = out("F", (B, 3, 3)) if F_out is None else F_out
    residual = out("residual", (B, N))
    epi = out("epi", (B, N), want_epi)
    save = torch.empty(B, L.dfepe_save_floats(), device=dev, dtype=torch.float32) if want_save else None
    w_out = out("weights", (B, N), logits)
    with _on(dev):
        rc = L.dfepe_w8pt_fwd(_ptr(pts1), _ptr(pts2), _ptr(weights), B, N, 1, _flags(raw, logits, row_per_pair, extra_flags), float(image_w),
                              float(image_h), float(clamp_at), _ptr(F), _ptr(residual), _ptr(epi), _ptr(save), _ptr(w_out), _stream())
    _lib.check(rc, "dfepe_w8pt_fwd")
    return F, residual, epi, save, w_out


def w8pt_backward(pts1, pts2, weights, raw, image_w, image_h, clamp_at, save, F, gF, gRes, gEpi, logits=False, gW_extra=None,
                  out: Optional[Tensor] = None, want_pts: bool = False, row_per_pair: bool = False, g_scale: Optional[Tensor] = None,
                  pending_loss_head: Optional[Tensor] = None, extra_flags: int = 0):
    """Raw launch of the adjoint; returns d/d(weights) (or d/d(logits) when ``logits``; then ``weights`` must be the
    forward's weights_out) and, when ``want_pts``, the gradients w.r.t. the points ([B,N,3] x 2, or [B,N,4] for raw matches)."""
    L = _lib.lib()
    B, N = weights.shape
    gW = torch.empty_like(weights) if out is None else out
    gP1 = gP2 = None
    if want_pts:
        gP1 = torch.empty_like(pts1)
        gP2 = None if raw else torch.empty_like(pts2)
    with _on(weights.device):
        rc = L.dfepe_w8pt_bwd(_ptr(pts1), _ptr(pts2), _ptr(weights), B, N, 1, _flags(raw, logits, row_per_pair, extra_flags), float(image_w), float(image_h),
                              float(clamp_at), _ptr(save), _ptr(F), _ptr(gF), _ptr(gRes), _ptr(gEpi), _ptr(gW_extra), _ptr(g_scale), _ptr(gW),
                              _ptr(gP1), _ptr(gP2), _ptr(pending_loss_head), _stream())
    _lib.check(rc, "dfepe_w8pt_bwd")
    return (gW, gP1, gP2) if want_pts else gW


def eight_point(X: Tensor, Y: Tensor, w: Optional[Tensor], essential: bool, normalize: bool = True) -> Tensor:
    """Textbook normalised 8-point on 2-D points X, Y [B,N,2] with optional per-correspondence weights [B,N]
    (utils_F._F_from_XY / _E_from_XY after the K^-1 step): sqrt(2) Hartley (none when ``normalize`` is False),
    unnormalised rows, S3 -> 0 or (1,1,0)."""
    X, Y = _prep(X, "X"), _prep(Y, "Y")
    B, N = X.shape[0], X.shape[1]
    ones = torch.ones(B, N, 1, device=X.device)
    p1 = torch.cat((X, ones), 2).contiguous()
    p2 = torch.cat((Y, ones), 2).contiguous()
    wt = torch.ones(B, N, device=X.device) if w is None else _prep(w.reshape(B, N), "w")
    L = _lib.lib()
    F = torch.empty(B, 3, 3, device=X.device)
    residual = torch.empty(B, N, device=X.device)
    flags = (_lib.W8PT_SQRT2 | _lib.W8PT_NO_ROWNORM | (_lib.W8PT_FORCE_110 if essential else 0) |
             (0 if normalize else _lib.W8PT_NO_HARTLEY))
    with _on(X.device):
        rc = L.dfepe_w8pt_fwd(_ptr(p1), _ptr(p2), _ptr(wt), B, N, 1, flags, 0.0, 0.0, 0.5, _ptr(F), _ptr(residual), None, None, None,
                              _stream())
    _lib.check(rc, "dfepe_w8pt_fwd")
    return F


def eight_point_rows(rows: Tensor, T1: Optional[Tensor], T2: Optional[Tensor], essential: bool) -> Tensor:
    """Closing steps of the textbook solvers on an explicit design matrix rows [B,N,9] (the dense-W form,
    utils_F.py:129-155,245-275): smallest right singular vector, S3 -> 0 or (1,1,0), T2^T F T1 (T1, T2 [B,3,3] or None)."""
    rows = _prep(rows, "rows")
    _shape(rows, "rows (design matrix)", None, None, 9)
    B, N = rows.shape[0], rows.shape[1]
    if (T1 is None) != (T2 is None):
        raise ValueError("T1 and T2 come together")
    if T1 is not None:
        T1, T2 = _prep(T1, "T1"), _prep(T2, "T2")
        _shape(T1, "T1", B, 3, 3)
        _shape(T2, "T2", B, 3, 3)
    F = torch.empty(B, 3, 3, device=rows.device)
    with _on(rows.device):
        rc = _lib.lib().dfepe_w8pt_rows_fwd(_ptr(rows), B, N, _lib.W8PT_FORCE_110 if essential else 0, _ptr(T1), _ptr(T2), _ptr(F), _stream())
    _lib.check(rc, "dfepe_w8pt_rows_fwd")
    return F


def _cf(t: Optional[Tensor]) -> Optional[Tensor]:
    return None if t is None else t.contiguous().float()


class _W8ptFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pts1, pts2, weights, raw, image_w, image_h, clamp_at, want_epi, logits, extra_flags=0, dst=None):
        ctx.set_materialize_grads(False)  # unused outputs (e.g. the last layer's epi) must not cost a zero-fill
        F, residual, epi, save, w_out = w8pt_forward(pts1, pts2, weights, raw, image_w, image_h, clamp_at, want_epi,
                                                     want_save=True, logits=logits, extra_flags=extra_flags, dst=dst)
        ctx.extra_flags = extra_flags
        ctx.save_for_backward(pts1, pts2 if pts2 is not None else pts1.new_empty(0), w_out if logits else weights, save, F)
        ctx.cfg = (raw, image_w, image_h, clamp_at, want_epi, logits)
        outs = [F, residual]
        if want_epi:
            outs.append(epi)
        if logits:
            outs.append(w_out)
        return tuple(outs)

    @staticmethod
    def backward(ctx, gF, gRes, *rest):
        pts1, pts2, weights, save, F = ctx.saved_tensors
        raw, image_w, image_h, clamp_at, want_epi, logits = ctx.cfg
        rest = list(rest)
        gEpi = rest.pop(0) if want_epi else None
        gWout = rest.pop(0) if logits else None
        if gF is None and gRes is None and gEpi is None and gWout is None:
            return (None,) * 11
        want_pts = ctx.needs_input_grad[0] or (not raw and ctx.needs_input_grad[1])
        res = w8pt_backward(pts1, pts2 if pts2.numel() else None, weights, raw, image_w, image_h, clamp_at, save, F,
                            _cf(gF), _cf(gRes), _cf(gEpi), logits=logits, gW_extra=_cf(gWout), want_pts=want_pts,
                            extra_flags=ctx.extra_flags)
        if want_pts:
            gW, gP1, gP2 = res
            return gP1, gP2, gW, None, None, None, None, None, None, None, None
        return None, None, res, None, None, None, None, None, None, None, None


def w8pt(pts1: Tensor, pts2: Tensor, weights: Tensor, clamp_at: float = 0.5, want_epi: bool = False, normalize_rows: bool = True):
    """Differentiable fit on homogeneous points [B,N,3]; weights [B,N] or [B,1,N].  Gradients flow to the weights and,
    when the point tensors require grad, to both point sets.  ``normalize_rows=False`` = Fit(normalize_SVD=False)
    (DeepFNet.py:211): the design rows enter X un-normalised; gradients then flow to the weights only."""
    pts1, pts2 = _prep(pts1, "pts1"), _prep(pts2, "pts2")
    w = _prep(weights.reshape(weights.shape[0], -1), "weights")
    B, N = w.shape
    _shape(pts1, "pts1 (homogeneous points)", B, N, 3)
    _shape(pts2, "pts2 (homogeneous points)", B, N, 3)
    if not normalize_rows and torch.is_grad_enabled() and (pts1.requires_grad or pts2.requires_grad):
        raise _lib.DfepeError("w8pt(normalize_rows=False): gradients w.r.t. the points are not built for un-normalised rows")
    return _W8ptFunction.apply(pts1, pts2, w, False, 0.0, 0.0, clamp_at, want_epi, False, 0 if normalize_rows else _lib.W8PT_NO_ROWNORM)


def w8pt_raw(matches: Tensor, weights: Tensor, image_w: float, image_h: float, clamp_at: float = 0.5,
             want_epi: bool = True):
    """Differentiable fit straight from pixel matches [B,N,4] (image-size normalisation fused)."""
    m = _prep(matches, "matches")
    w = _prep(weights.reshape(weights.shape[0], -1), "weights")
    _shape(m, "matches (pixel x1,y1,x2,y2)", w.shape[0], w.shape[1], 4)
    return _W8ptFunction.apply(m, None, w, True, float(image_w), float(image_h), clamp_at, want_epi, False)


def w8pt_raw_logits(matches: Tensor, logits: Tensor, image_w: float, image_h: float, clamp_at: float = 0.5,
                    want_epi: bool = True, dst: Optional[dict] = None):
    """As w8pt_raw but takes the estimator's logits and fuses F.softmax(dim=N); additionally returns the weights
    (differentiable: the next estimator layer consumes them).  Returns (F, residual[, epi], weights).
    ``dst`` (see w8pt_forward): rows of the caller's per-layer stacks to write the outputs into."""
    m = _prep(matches, "matches")
    l = _prep(logits.reshape(logits.shape[0], -1), "logits")
    _shape(m, "matches (pixel x1,y1,x2,y2)", l.shape[0], l.shape[1], 4)
    return _W8ptFunction.apply(m, None, l, True, float(image_w), float(image_h), clamp_at, want_epi, True, 0, dst)


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
        with _on(F_layers.device):
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
        with _on(F_layers.device):
            rc = lib.dfepe_floss_bwd(_ptr(F_layers), L, B, _ptr(T1c), _ptr(T2c), st, _ptr(K), _ptr(virt1), _ptr(virt2),
                                     virt1.shape[1], clamp_at, _ptr(g_loss_sum), 0.0, None, _ptr(g_E), _ptr(gF), _stream())
        _lib.check(rc, "dfepe_floss_bwd")
        return gF, None, None, None, None, None, None


def floss(F_layers: Tensor, T1: Tensor, T2: Tensor, K: Tensor, virt1: Tensor, virt2: Tensor, clamp_at: float):
    """F_layers [L,B,3,3] -> (loss_sum [L,B] = sum over virtual points of the clamped residual, E_layers [L,B,3,3])."""
    F_layers, K, virt1, virt2 = _prep(F_layers, "F_layers"), _prep(K, "K"), _prep(virt1, "virt1"), _prep(virt2, "virt2")
    _shape(F_layers, "F_layers", None, None, 3, 3)
    B = F_layers.shape[1]
    _shape(K, "K (one intrinsic matrix per pair)", B, 3, 3)
    _shape(virt1, "virt1 (homogeneous pixel points)", B, None, 3)
    _shape(virt2, "virt2", B, virt1.shape[1], 3)
    for T in (T1, T2):
        if not ((T.dim() == 2 and tuple(T.shape) == (3, 3)) or (T.dim() == 3 and T.shape[0] in (1, B) and tuple(T.shape[1:]) == (3, 3))):
            raise ValueError(f"T1/T2 must be [3,3], [1,3,3] or [{B},3,3], got {tuple(T.shape)}")
    return _FlossFunction.apply(F_layers, T1, T2, K, virt1, virt2, clamp_at)


# ------------------------------------------------------------------------------------------------
# pose loss
# ------------------------------------------------------------------------------------------------
def _sum_opt(a: Optional[Tensor], b: Optional[Tensor]) -> Optional[Tensor]:
    if a is None:
        return b
    return a if b is None else a + b


class _PoseFunction(torch.autograd.Function):
    """Outputs: qt [2,L,B] (q_l2 | t_l2 in one buffer, so that one reduction serves both), its two halves q_l2, t_l2 [L,B] as
    outputs of their own over the same memory, ang [2,L,B] = R_deg | t_deg and sel [L,B] (not differentiable)."""

    @staticmethod
    def forward(ctx, E_layers, q_gt, t_gt, R_gt):
        lib = _lib.lib()
        L, B = E_layers.shape[0], E_layers.shape[1]
        dev = E_layers.device
        ctx.set_materialize_grads(False)
        qt = torch.empty(2, L, B, device=dev, dtype=torch.float32)
        ang = torch.empty(2, L, B, device=dev, dtype=torch.float32)
        sel = torch.empty(L, B, device=dev, dtype=torch.int32)
        q_l2, t_l2 = row_of(qt, 0), row_of(qt, 1)
        with _on(dev):
            rc = lib.dfepe_pose_fwd(_ptr(E_layers), L, B, _ptr(q_gt), _ptr(t_gt), _ptr(R_gt), _ptr(q_l2), _ptr(t_l2),
                                    _ptr(ang[0]), _ptr(ang[1]), _ptr(sel), _stream())
        _lib.check(rc, "dfepe_pose_fwd")
        ctx.save_for_backward(E_layers, q_gt, t_gt)
        ctx.mark_non_differentiable(ang, sel)
        return qt, q_l2, t_l2, ang, sel

    @staticmethod
    def backward(ctx, g_qt, g_q, g_t, _a, _b):
        E_layers, q_gt, t_gt = ctx.saved_tensors
        lib = _lib.lib()
        L, B = E_layers.shape[0], E_layers.shape[1]
        if g_qt is not None:
            g_qt = g_qt.contiguous().float()
            g_q, g_t = _sum_opt(g_q, g_qt[0]), _sum_opt(g_t, g_qt[1])
        if g_q is None and g_t is None:
            return None, None, None, None
        gE = torch.empty_like(E_layers)
        g_q = None if g_q is None else g_q.contiguous().float()
        g_t = None if g_t is None else g_t.contiguous().float()
        with _on(E_layers.device):
            rc = lib.dfepe_pose_bwd(_ptr(E_layers), L, B, _ptr(q_gt), _ptr(t_gt), _ptr(g_q), _ptr(g_t), 0.0, 0.0, 0.0, 0.0, None,
                                    _ptr(gE), _stream())
        _lib.check(rc, "dfepe_pose_bwd")
        return gE, None, None, None


def _pose_args(E_layers: Tensor, q_gt: Tensor, t_gt: Tensor, R_gt: Tensor):
    _shape(E_layers, "E_layers", None, None, 3, 3)
    B = E_layers.shape[1]
    if q_gt.numel() != 4 * B or t_gt.numel() != 3 * B or R_gt.numel() != 9 * B:
        raise ValueError(f"q_gt / t_gt / R_gt must hold {B} quaternions / translations / rotations, got {tuple(q_gt.shape)}, {tuple(t_gt.shape)}, {tuple(R_gt.shape)}")
    return _prep(E_layers, "E_layers"), _prep(q_gt.reshape(B, 4), "q_gt"), _prep(t_gt.reshape(B, 3), "t_gt"), _prep(R_gt.reshape(B, 3, 3), "R_gt")


def pose_errors(E_layers: Tensor, q_gt: Tensor, t_gt: Tensor, R_gt: Tensor):
    """E_layers [L,B,3,3]; q_gt [B,4(,1)], t_gt [B,3(,1)], R_gt [B,3,3] (camera motion).
    Returns q_l2, t_l2 (differentiable w.r.t. E), R_deg, t_deg, sel — all [L,B]."""
    _, q_l2, t_l2, ang, sel = _PoseFunction.apply(*_pose_args(E_layers, q_gt, t_gt, R_gt))
    return q_l2, t_l2, ang[0], ang[1], sel


def pose_errors_packed(E_layers: Tensor, q_gt: Tensor, t_gt: Tensor, R_gt: Tensor):
    """As pose_errors, returning the buffers the kernel filled: (qt [2,L,B] = q_l2 | t_l2, q_l2, t_l2, ang [2,L,B] = R_deg | t_deg,
    sel); qt, q_l2 and t_l2 are differentiable and share memory."""
    return _PoseFunction.apply(*_pose_args(E_layers, q_gt, t_gt, R_gt))


def _slice_of(buf: Tensor, off: int, n: int) -> Tensor:
    """buf[off:off+n] of a contiguous 1-D buffer as a tensor of its own over the same memory (not an autograd view)."""
    t = torch.empty(0, dtype=buf.dtype, device=buf.device)
    t.set_(buf.untyped_storage(), buf.storage_offset() + off, (n,), None)
    return t


def _stats_launch(sets, C: int, want_min: bool):
    """sets: four (tensor [R,C] | None, scale).  One dfepe_loss_stats launch -> (per set: (means [R_k], overall [] scalar) over one
    buffer, or (None, None); row_min; col_min)."""
    dev = next(x for x, _ in sets if x is not None).device
    total = sum(x.shape[0] + 1 for x, _ in sets if x is not None)
    out = torch.empty(total, device=dev, dtype=torch.float32)
    x0 = sets[0][0]
    row_min = torch.empty(x0.shape[0], device=dev, dtype=torch.float32) if want_min else None
    col_min = torch.empty(C, device=dev, dtype=torch.float32) if want_min else None
    args = []
    for x, sc in sets:
        args += [_ptr(x), 0 if x is None else x.shape[0], float(sc)]
    with _on(dev):
        rc = _lib.lib().dfepe_loss_stats(*args, C, _ptr(out), _ptr(row_min), _ptr(col_min), _stream())
    _lib.check(rc, "dfepe_loss_stats")
    blocks, off = [], 0
    for x, _ in sets:
        if x is None:
            blocks.append((None, None))
        else:
            R = x.shape[0]
            blocks.append((_slice_of(out, off, R), _slice_of(out, off + R, 1).view(())))
            off += R + 1
    return blocks, row_min, col_min


def _stat_block_grad(gm: Optional[Tensor], go: Optional[Tensor], R: int, C: int, scale: float) -> Optional[Tensor]:
    """Gradient w.r.t. the [R,C] rows of a dfepe_loss_stats set from the gradients of its R row means and of their mean."""
    if gm is None and go is None:
        return None
    coef = _sum_opt(gm, None if go is None else (go * (1.0 / R)).reshape(1).expand(R)) * (scale / C)
    return coef.unsqueeze(1).expand(R, C)


class _LossStatsFunction(torch.autograd.Function):
    """dfepe_loss_stats as a differentiable op on up to four [R_k, C] row sets (set 0 optionally with its minima, which carry no
    gradient).  Outputs: (means [R_k], overall scalar) per set (None, None for absent sets), then row_min, col_min."""

    @staticmethod
    def forward(ctx, want_min, x0, s0, x1, s1, x2, s2, x3, s3):
        ctx.set_materialize_grads(False)
        sets = [(x0, s0), (x1, s1), (x2, s2), (x3, s3)]
        C = x0.shape[1]
        ctx.meta = [(None if x is None else x.shape[0], float(sc)) for x, sc in sets]
        ctx.C = C
        blocks, row_min, col_min = _stats_launch(sets, C, want_min)
        if want_min:
            ctx.mark_non_differentiable(row_min, col_min)
        return tuple(t for blk in blocks for t in blk) + (row_min, col_min)

    @staticmethod
    def backward(ctx, *gs):
        out = [None]
        for k, (R, sc) in enumerate(ctx.meta):
            out += [None if R is None else _stat_block_grad(gs[2 * k], gs[2 * k + 1], R, ctx.C, sc), None]
        return tuple(out)


def loss_stats(sets, want_min: bool = False):
    """sets: up to four (x [R,C], scale) pairs (None for an absent set), the first one present.  Returns (blocks, row_min, col_min):
    blocks[k] = (means [R_k] = scale_k * mean over C of every row, overall = the mean of those) or (None, None); the minima
    (set 0, times scale_0, no gradient) only with want_min."""
    sets = list(sets) + [None] * (4 - len(sets))
    flat = []
    C = sets[0][0].shape[1]
    for s in sets:
        if s is None:
            flat += [None, 0.0]
        else:
            x = _prep(s[0], "rows")
            _shape(x, "rows", None, C)
            flat += [x, float(s[1])]
    res = _LossStatsFunction.apply(bool(want_min), *flat)
    return [(res[2 * k], res[2 * k + 1]) for k in range(4)], res[8], res[9]


class _TailJacFunction(torch.autograd.Function):
    """get_all_loss_DeepF's per-layer body and get_Rt_loss's loop as ONE launch (dfepe_loss_tail_jac) plus ONE for every batch
    statistic the two functions return (dfepe_loss_stats), with a one-launch adjoint for whatever upstream gradients arrive
    (dfepe_loss_tail_bwd): the reference's callers mix clamps and balances themselves (Train_model_pipeline.py:580-586), so no
    coefficient is baked in.  Outputs (None where absent): loss_sum [L,B], E_layers [L,B,3,3], m_loss [L] / o_loss [] (per-layer
    means of loss_sum / M and loss_F), row_min [L], col_min [B] (loss_min_layers / loss_min_batch), m_x / o_x (statistics of the
    optional fourth row set ``extra`` [n,B]), and -- with ground truth -- qt [2,L,B], q_l2, t_l2, ang [2,L,B], sel as in
    _PoseFunction, m_q / o_q, m_t / o_t."""

    @staticmethod
    def forward(ctx, F_layers, T1c, T2c, t_stride, K, virt1, virt2, clamp_at, q_gt, t_gt, R_gt, want_floss_jac, extra, extra_scale):
        lib = _lib.lib()
        L, B = F_layers.shape[0], F_layers.shape[1]
        M = virt1.shape[1]
        dev = F_layers.device
        ctx.set_materialize_grads(False)
        loss_sum = torch.empty(L, B, device=dev, dtype=torch.float32)
        E_layers = torch.empty(L, B, 3, 3, device=dev, dtype=torch.float32)
        J = torch.empty(L, B, 27, device=dev, dtype=torch.float32)
        pose = q_gt is not None
        qt = torch.empty(2, L, B, device=dev, dtype=torch.float32) if pose else None
        ang = torch.empty(2, L, B, device=dev, dtype=torch.float32) if pose else None
        sel = torch.empty(L, B, device=dev, dtype=torch.int32) if pose else None
        q_l2 = row_of(qt, 0) if pose else None
        t_l2 = row_of(qt, 1) if pose else None
        with _on(dev):
            rc = lib.dfepe_loss_tail_jac(_ptr(F_layers), L, B, _ptr(T1c), _ptr(T2c), t_stride, _ptr(K), _ptr(virt1), _ptr(virt2), M,
                                         float(clamp_at), _ptr(q_gt), _ptr(t_gt), _ptr(R_gt), 1 if want_floss_jac else 0, _ptr(loss_sum),
                                         _ptr(E_layers), _ptr(q_l2), _ptr(t_l2), _ptr(ang[0]) if pose else None,
                                         _ptr(ang[1]) if pose else None, _ptr(sel), _ptr(J), _stream())
        _lib.check(rc, "dfepe_loss_tail_jac")
        ((m_loss, o_loss), (m_q, o_q), (m_t, o_t), (m_x, o_x)), row_min, col_min = _stats_launch(
            [(loss_sum, 1.0 / M), (q_l2, 1.0), (t_l2, 1.0), (extra, extra_scale)], B, True)
        ctx.save_for_backward(J, F_layers, T1c, T2c, K, virt1, virt2)
        ctx.cfg = (t_stride, float(clamp_at), bool(want_floss_jac), 1.0 / M, None if extra is None else (extra.shape[0], float(extra_scale)))
        ctx.mark_non_differentiable(*([row_min, col_min] + ([ang, sel] if pose else [])))  # ONE call: a second one replaces the first
        return loss_sum, E_layers, m_loss, o_loss, row_min, col_min, m_x, o_x, qt, q_l2, t_l2, ang, sel, m_q, o_q, m_t, o_t

    @staticmethod
    def backward(ctx, g_loss_sum, g_E, gm_loss, go_loss, _rm, _cm, gm_x, go_x, g_qt, g_q, g_t, _a, _b, gm_q, go_q, gm_t, go_t):
        J, F_layers, T1c, T2c, K, virt1, virt2 = ctx.saved_tensors
        t_stride, clamp_at, want_floss_jac, loss_scale, extra_meta = ctx.cfg
        lib = _lib.lib()
        L, B = J.shape[0], J.shape[1]
        if g_qt is not None:
            g_qt = g_qt.contiguous().float()
            g_q, g_t = _sum_opt(g_q, g_qt[0]), _sum_opt(g_t, g_qt[1])
        g_extra = None if extra_meta is None else _stat_block_grad(gm_x, go_x, extra_meta[0], B, extra_meta[1])
        if all(g is None for g in (g_loss_sum, g_E, g_q, g_t, gm_loss, go_loss, gm_q, go_q, gm_t, go_t)):
            return (None,) * 12 + (g_extra, None)
        if (g_loss_sum is not None or gm_loss is not None or go_loss is not None) and not want_floss_jac:
            raise _lib.DfepeError("the F-loss Jacobian was switched off for this call (loss_params['floss_grad'] = False) but a gradient "
                                  "arrived on loss_F / loss_layers")
        c = lambda g: None if g is None else g.contiguous().float()
        gF = torch.empty(L, B, 3, 3, device=J.device, dtype=torch.float32)
        with _on(J.device):
            rc = lib.dfepe_loss_tail_bwd(_ptr(J), L, B, _ptr(c(g_loss_sum)), _ptr(c(g_q)), _ptr(c(g_t)), _ptr(c(gm_loss)), _ptr(c(go_loss)),
                                         _ptr(c(gm_q)), _ptr(c(go_q)), _ptr(c(gm_t)), _ptr(c(go_t)), float(loss_scale), _ptr(gF), _stream())
            _lib.check(rc, "dfepe_loss_tail_bwd")
            if g_E is not None:  # a gradient on the E matrices themselves (none of the reference's losses has one): the stand-alone adjoint
                gF2 = torch.empty_like(gF)
                rc = lib.dfepe_floss_bwd(_ptr(F_layers), L, B, _ptr(T1c), _ptr(T2c), t_stride, _ptr(K), _ptr(virt1), _ptr(virt2),
                                         virt1.shape[1], clamp_at, None, 0.0, None, _ptr(c(g_E)), _ptr(gF2), _stream())
                _lib.check(rc, "dfepe_floss_bwd")
                gF = gF + gF2
        return (gF,) + (None,) * 11 + (g_extra, None)


def _floss_args(F_layers, T1, T2, K, virt1, virt2):
    F_layers, K, virt1, virt2 = _prep(F_layers, "F_layers"), _prep(K, "K"), _prep(virt1, "virt1"), _prep(virt2, "virt2")
    _shape(F_layers, "F_layers", None, None, 3, 3)
    B = F_layers.shape[1]
    _shape(K, "K (one intrinsic matrix per pair)", B, 3, 3)
    _shape(virt1, "virt1 (homogeneous pixel points)", B, None, 3)
    _shape(virt2, "virt2", B, virt1.shape[1], 3)
    for T in (T1, T2):
        if not ((T.dim() == 2 and tuple(T.shape) == (3, 3)) or (T.dim() == 3 and T.shape[0] in (1, B) and tuple(T.shape[1:]) == (3, 3))):
            raise ValueError(f"T1/T2 must be [3,3], [1,3,3] or [{B},3,3], got {tuple(T.shape)}")
    T1c, st1 = _t_arg(T1, B)
    T2c, st2 = _t_arg(T2, B)
    if st1 != st2:
        T1c, T2c, st1 = T1c.expand(B, 3, 3).contiguous(), T2c.expand(B, 3, 3).contiguous(), 9
    return F_layers, T1c, T2c, st1, K, virt1, virt2


def loss_tail_jac(F_layers: Tensor, T1: Tensor, T2: Tensor, K: Tensor, virt1: Tensor, virt2: Tensor, clamp_at: float,
                  q_gt: Optional[Tensor] = None, t_gt: Optional[Tensor] = None, R_gt: Optional[Tensor] = None, floss_grad: bool = True,
                  extra: Optional[Tensor] = None, extra_scale: float = 1.0) -> dict:
    """F-loss sums + E-from-F (+ pose errors when the ground truth is given) of every layer in one launch and every batch
    statistic of them in a second one, differentiable for any upstream gradients.  Returns a dict: loss_sum [L,B], E_layers,
    m_loss [L] / o_loss (per-layer means of loss_sum / M, their mean), row_min [L], col_min [B], m_extra / o_extra (statistics of the
    optional row set ``extra`` [n,B], times extra_scale) and, with ground truth, qt, q_l2, t_l2, ang, sel, m_q, o_q, m_t, o_t.
    Needs M <= 112 virtual points (raises DfepeError 'unsupported' otherwise: use floss + pose_errors)."""
    F_layers, T1c, T2c, st, K, virt1, virt2 = _floss_args(F_layers, T1, T2, K, virt1, virt2)
    if q_gt is not None:
        _, q_gt, t_gt, R_gt = _pose_args(F_layers, q_gt, t_gt, R_gt)
    if extra is not None:
        extra = _prep(extra, "extra")
        _shape(extra, "extra rows", None, F_layers.shape[1])
    r = _TailJacFunction.apply(F_layers, T1c, T2c, st, K, virt1, virt2, float(clamp_at), q_gt, t_gt, R_gt, bool(floss_grad), extra,
                               float(extra_scale))
    keys = ("loss_sum", "E_layers", "m_loss", "o_loss", "row_min", "col_min", "m_extra", "o_extra", "qt", "q_l2", "t_l2", "ang", "sel",
            "m_q", "o_q", "m_t", "o_t")
    return dict(zip(keys, r))


# ------------------------------------------------------------------------------------------------
# stand-alone geometry entry points
# ------------------------------------------------------------------------------------------------
class _EpiResidualFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pts1, pts2, F, clamp_at):
        B, N = pts1.shape[0], pts1.shape[1]
        out = torch.empty(B, N, device=pts1.device, dtype=torch.float32)
        with _on(pts1.device):
            rc = _lib.lib().dfepe_epi_residual_fwd(_ptr(pts1), _ptr(pts2), _ptr(F), B, N, float(clamp_at), _ptr(out), _stream())
        _lib.check(rc, "dfepe_epi_residual_fwd")
        ctx.save_for_backward(pts1, pts2, F)
        ctx.clamp_at = float(clamp_at)
        return out

    @staticmethod
    def backward(ctx, g):
        pts1, pts2, F = ctx.saved_tensors
        B, N = pts1.shape[0], pts1.shape[1]
        gF = torch.empty_like(F)
        g = g.contiguous().float()
        with _on(pts1.device):
            rc = _lib.lib().dfepe_epi_residual_bwd(_ptr(pts1), _ptr(pts2), _ptr(F), B, N, ctx.clamp_at, _ptr(g), _ptr(gF), _stream())
        _lib.check(rc, "dfepe_epi_residual_bwd")
        return None, None, gF, None


def epi_residual(pts1: Tensor, pts2: Tensor, F: Tensor, clamp_at: float = 0.5) -> Tensor:
    """pts [B,N,3], F [B,3,3] -> [B,N]; differentiable w.r.t. F."""
    pts1, pts2, F = _prep(pts1, "pts1"), _prep(pts2, "pts2"), _prep(F, "F")
    _shape(pts1, "pts1 (homogeneous points)", None, None, 3)
    _shape(pts2, "pts2", pts1.shape[0], pts1.shape[1], 3)
    _shape(F, "F", pts1.shape[0], 3, 3)
    return _EpiResidualFunction.apply(pts1, pts2, F, clamp_at)


def epi_metrics(kind: int, F: Tensor, X: Tensor, Y: Tensor, clamp_at: Optional[float] = None, eps: float = 0.0) -> Tensor:
    """kind 0 sym-epi (squared), 1 Sampson, 2 epi-distance (3 planes).  F [B,3,3]; X, Y [B,N,2], or [B,N,3] homogeneous points
    that are used as they are; clamp_at None = no clamp."""
    F, X, Y = _prep(F, "F"), _prep(X, "X"), _prep(Y, "Y")
    if X.dim() != 3 or X.shape != Y.shape or X.shape[2] not in (2, 3) or F.shape != (X.shape[0], 3, 3):
        raise ValueError(f"epi_metrics: F [B,3,3], X and Y [B,N,2|3] expected, got {tuple(F.shape)}, {tuple(X.shape)}, {tuple(Y.shape)}")
    B, N = X.shape[0], X.shape[1]
    kind = int(kind) | (_lib.EPI_HOMOGENEOUS if X.shape[2] == 3 else 0)
    clamp_at = -1.0 if clamp_at is None else float(clamp_at)
    out = torch.empty((3, B, N) if (kind & 7) == 2 else (B, N), device=X.device, dtype=torch.float32)
    with _on(X.device):
        rc = _lib.lib().dfepe_epi_metrics(int(kind), _ptr(F), _ptr(X), _ptr(Y), B, N, float(clamp_at), float(eps), _ptr(out), _stream())
    _lib.check(rc, "dfepe_epi_metrics")
    return out


_GEO_OUT = {0: 4, 1: 1, 2: 1, 3: 9, 4: 21, 5: 9, 6: 9}


def geo_misc(kind: int, in0: Tensor, in1: Optional[Tensor] = None) -> Tensor:
    in0 = _prep(in0, "in0")
    in1 = None if in1 is None else _prep(in1, "in1")
    n = in0.shape[0]
    out = torch.empty(n, _GEO_OUT[kind], device=in0.device, dtype=torch.float32)
    with _on(in0.device):
        rc = _lib.lib().dfepe_geo_misc(int(kind), _ptr(in0), _ptr(in1), n, _ptr(out), _stream())
    _lib.check(rc, "dfepe_geo_misc")
    return out


def rot_to_quat(R: Tensor) -> Tensor:
    return geo_misc(0, R.reshape(-1, 9))


def rot_angle_deg(R0: Tensor, R1: Tensor) -> Tensor:
    return geo_misc(1, R0.reshape(-1, 9), R1.reshape(-1, 9))[:, 0]


def vector_angle_deg(v1: Tensor, v2: Tensor) -> Tensor:
    return geo_misc(2, v1.reshape(-1, 3), v2.reshape(-1, 3))[:, 0]


def project_essential(E: Tensor) -> Tensor:
    return geo_misc(3, E.reshape(-1, 9)).reshape(-1, 3, 3)


def congruence(F: Tensor, A: Tensor) -> Tensor:
    """A^T F A for batches of 3x3 matrices: E = K^T T^T F T K with A = T K (train_good_utils.py:356-358), or K^T F K."""
    return geo_misc(5, F.reshape(-1, 9), A.reshape(-1, 9)).reshape(-1, 3, 3)


class _RowDotFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        ctx.set_materialize_grads(False)
        n, B, N = a.shape
        out = torch.empty(n, B, device=a.device, dtype=torch.float32)
        with _on(a.device):
            rc = _lib.lib().dfepe_row_dot(_ptr(a), a.stride(0), _ptr(b), b.stride(0), n, B, N, _ptr(out), _stream())
        _lib.check(rc, "dfepe_row_dot")
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None
        a, b = ctx.saved_tensors
        g = g.unsqueeze(2)
        return (g * b if ctx.needs_input_grad[0] else None), (g * a if ctx.needs_input_grad[1] else None)


def row_dot(a: Tensor, b: Tensor) -> Tensor:
    """out[l,b] = sum_n a[l,b,n] * b[l,b,n] for two [n,B,N] stacks whose layers are contiguous [B,N] blocks (any distance apart:
    the strided stacks of alias_rows are taken as they are).  One launch; differentiable (plain torch products in the backward)."""
    if a.shape != b.shape or a.dim() != 3:
        raise ValueError(f"row_dot: two [n,B,N] stacks expected, got {tuple(a.shape)}, {tuple(b.shape)}")
    fix = lambda t: t if (t.is_cuda and t.dtype == torch.float32 and t.stride(2) == 1 and t.stride(1) == t.shape[2]) else _prep(t, "stack")
    return _RowDotFunction.apply(fix(a), fix(b))


class _CongruenceFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, F, A):
        ctx.save_for_backward(A)
        return geo_misc(5, F.reshape(-1, 9), A.reshape(-1, 9)).reshape(-1, 3, 3)

    @staticmethod
    def backward(ctx, g):
        A, = ctx.saved_tensors
        return geo_misc(5, g.reshape(-1, 9), A.transpose(1, 2).reshape(-1, 9)).reshape(-1, 3, 3), None  # A g A^T


def congruence_diff(F: Tensor, A: Tensor) -> Tensor:
    """A^T F A like congruence, differentiable w.r.t. F (F_ests = T^T F_est T of get_all_loss_DeepF, train_good_utils.py:366)."""
    return _CongruenceFunction.apply(_prep(F, "F"), _prep(A, "A"))


def camera_rotation(delta_4x4: Tensor) -> Tensor:
    """inv(delta)[:, :3, :3] for scene motions delta [B,4,4]: the ground-truth camera rotation of get_Rt_loss
    (train_good_utils.py:134,170), one launch, no LAPACK round trip / host synchronisation."""
    d = _prep(delta_4x4, "delta_Rtijs_4_4")
    _shape(d, "delta_Rtijs_4_4", None, 4, 4)
    return geo_misc(6, d.reshape(-1, 16)).reshape(-1, 3, 3)


def deepf_input(matches: Tensor, image_w: float, image_h: float, quality: Optional[Tensor] = None, want_pts: bool = True,
                recurrent_copies: int = 0, recurrent_channels: int = 3):
    """DeepFNet.get_input in one launch (DeepFNet.py:362-391): matches [B,N,4] pixels (+ quality [B,N,Q]) ->
    (weight_in [B,4+Q,N], pts1 [B,N,3], pts2 [B,N,3], stores); not differentiable (the matches are data).
    ``recurrent_copies`` = n > 0: the same launch also fills the point (and quality) channels of n channel-major buffers
    ``stores`` [n, 4+Q+recurrent_channels, B, N] -- the inputs of the later estimator calls (DeepFNet.py:484-489), whose remaining
    channels the fit kernels write as plain [B,N] blocks; weight_in then is a [B,4+Q,N] view of one more such buffer."""
    m = _prep(matches, "matches")
    _shape(m, "matches (pixel x1,y1,x2,y2)", None, None, 4)
    B, N = m.shape[0], m.shape[1]
    Q = 0
    if quality is not None:
        quality = _prep(quality, "quality")
        _shape(quality, "quality", B, N, None)
        Q = quality.shape[2]
    dev = m.device
    p1 = torch.empty(B, N, 3, device=dev, dtype=torch.float32) if want_pts else None
    p2 = torch.empty(B, N, 3, device=dev, dtype=torch.float32) if want_pts else None
    lib = _lib.lib()
    if recurrent_copies > 0:
        C = 4 + Q + recurrent_channels
        buf = torch.empty(recurrent_copies + 1, C, B, N, device=dev, dtype=torch.float32)  # copy 0 serves the first estimator call
        with _on(dev):
            rc = lib.dfepe_deepf_input(_ptr(m), _ptr(quality), B, N, Q, float(image_w), float(image_h), _ptr(buf), B * N, N,
                                       recurrent_copies + 1, C * B * N, _ptr(p1), _ptr(p2), _stream())
        _lib.check(rc, "dfepe_deepf_input")
        return buf[0, :4 + Q].permute(1, 0, 2), p1, p2, buf[1:]
    w_in = torch.empty(B, 4 + Q, N, device=dev, dtype=torch.float32)
    with _on(dev):
        rc = lib.dfepe_deepf_input(_ptr(m), _ptr(quality), B, N, Q, float(image_w), float(image_h), _ptr(w_in), N, (4 + Q) * N, 1, 0,
                                   _ptr(p1), _ptr(p2), _stream())
    _lib.check(rc, "dfepe_deepf_input")
    return w_in, p1, p2, None


class _EstimatorInputFunction(torch.autograd.Function):
    """The [B,C,N] input of an update_weights call (DeepFNet.py:484-489: cat of the point channels, weights, epipolar residual,
    residual) WITHOUT the cat: ``store`` [C,B,N] already holds the point channels (deepf_input) and the three recurrent channels
    (the fit kernel wrote its outputs there: ``rec`` are those very rows); the result is the channel-major buffer seen as [B,C,N].
    Backward: the gradient's three recurrent channels go to the rows."""

    @staticmethod
    def forward(ctx, store, first, *rec):
        ctx.first = first
        out = torch.empty(0, dtype=store.dtype, device=store.device)
        C, B, N = store.shape
        out.set_(store.untyped_storage(), store.storage_offset(), (B, C, N), (N, B * N, 1))
        return out

    @staticmethod
    def backward(ctx, g):
        return (None, None) + tuple(g[:, ctx.first + k, :] for k in range(g.shape[1] - ctx.first))


def estimator_input(store: Tensor, first: int, rec_rows) -> Tensor:
    """See _EstimatorInputFunction: store [C,B,N] channel-major, rec_rows = the [B,N] tensors living in store[first:], in order."""
    for k, r in enumerate(rec_rows):
        if r.data_ptr() != store[first + k].data_ptr() or r.shape != store.shape[1:]:
            raise ValueError("estimator_input: the recurrent rows must be the channels of the store they were written into")
    return _EstimatorInputFunction.apply(store, first, *rec_rows)


def decompose_essential(E: Tensor):
    o = geo_misc(4, E.reshape(-1, 9))
    return o[:, 0:9].reshape(-1, 3, 3), o[:, 9:18].reshape(-1, 3, 3), o[:, 18:21]


def _cheirality_workspace(B: int, dev) -> Tensor:
    """Per-call scratch of dfepe_cheirality_ex (the per-pair constants its preparation launch leaves for the main kernel)."""
    return torch.empty((_lib.lib().dfepe_cheirality_workspace_bytes(B) + 7) // 8, dtype=torch.float64, device=dev)


def cheirality(E: Tensor, K: Tensor, matches: Tensor, depth_thres: float = 50.0, pre: Optional[Tensor] = None, fp64_only: bool = False,
               prepared: Optional[bool] = None):
    """E, K [B,3,3], matches [B,N,4] pixels -> (Rt_cam [B,3,4], winner [B] int32, counts [B,4] int32).
    ``pre`` [B,3,3]: decompose pre^T E pre instead (E = F and pre = T K fuses E-from-F into the launch).
    ``fp64_only``: every correspondence through the fp64 route (DFEPE_CHEIR_FP64_ONLY: the reference the adaptive default is
    tested against for exact equality of the counts).  ``prepared``: form the per-pair constants in a preparation launch (one lane
    per pair) instead of in the main kernel; same outputs bit for bit.  Default: from 2048 pairs on (measured 93.8 vs 98.7 us at
    4096 x 1000; below, where a pair's workgroup forms them once and shares them through LDS, the second launch costs more than it
    saves: 22.9 vs 21.1 us at 512 x 1000)."""
    E, K, m = _prep(E, "E"), _prep(K, "K"), _prep(matches, "matches")
    pre = None if pre is None else _prep(pre, "pre")
    _shape(m, "matches (pixel x1,y1,x2,y2)", None, None, 4)
    B, N = m.shape[0], m.shape[1]
    _shape(E, "E", B, 3, 3)
    _shape(K, "K (one intrinsic matrix per pair)", B, 3, 3)
    if pre is not None:
        _shape(pre, "pre", B, 3, 3)
    Rt = torch.empty(B, 3, 4, device=m.device, dtype=torch.float32)
    win = torch.empty(B, device=m.device, dtype=torch.int32)
    cnt = torch.empty(B, 4, device=m.device, dtype=torch.int32)
    ws = _cheirality_workspace(B, m.device) if (prepared if prepared is not None else B >= 2048) else None
    with _on(m.device):
        rc = _lib.lib().dfepe_cheirality_ex(_ptr(E), _ptr(pre), _ptr(K), _ptr(m), B, N, float(depth_thres),
                                            _lib.CHEIR_FP64_ONLY if fp64_only else 0, _ptr(ws), _ptr(Rt), _ptr(win), _ptr(cnt), _stream())
    _lib.check(rc, "dfepe_cheirality_ex")
    return Rt, win, cnt


def fit_pose(matches: Tensor, weights: Tensor, K: Tensor, image_w: float, image_h: float, depth_thres: float = 50.0,
             pre: Optional[Tensor] = None, clamp_at: float = 0.5, want_epi: bool = True, logits: bool = False, row_per_pair: bool = False):
    """One weighted 8-point fit and the cheirality-checked pose of its F (BASELINE config 5): w8pt_forward followed by
    cheirality(F, K, matches, depth_thres, pre=pre), same numbers, ONE launch when a cooperative workgroup serves the pair
    (128 < N <= 2048 below 3072 pairs).  Returns (F, residual, epi | None, weights_out | None, Rt_cam, winner, counts)."""
    m, w, K = _prep(matches, "matches"), _prep(weights, "weights"), _prep(K, "K")
    pre = None if pre is None else _prep(pre, "pre")
    _shape(m, "matches (pixel x1,y1,x2,y2)", None, None, 4)
    B, N = m.shape[0], m.shape[1]
    _shape(w, "weights", B, N)
    _shape(K, "K (one intrinsic matrix per pair)", B, 3, 3)
    if pre is not None:
        _shape(pre, "pre", B, 3, 3)
    dev = m.device
    F = torch.empty(B, 3, 3, device=dev)
    residual = torch.empty(B, N, device=dev)
    epi = torch.empty(B, N, device=dev) if want_epi else None
    w_out = torch.empty(B, N, device=dev) if logits else None
    Rt = torch.empty(B, 3, 4, device=dev)
    win = torch.empty(B, device=dev, dtype=torch.int32)
    cnt = torch.empty(B, 4, device=dev, dtype=torch.int32)
    ws = _cheirality_workspace(B, dev) if B >= 2048 else None
    with _on(dev):
        rc = _lib.lib().dfepe_w8pt_pose_fwd(_ptr(m), _ptr(w), B, N, _flags(True, logits, row_per_pair), float(image_w), float(image_h), float(clamp_at),
                                            _ptr(K), _ptr(pre), float(depth_thres), _ptr(F), _ptr(residual), _ptr(epi), _ptr(w_out), _ptr(Rt), _ptr(win),
                                            _ptr(cnt), _ptr(ws), _stream())
    _lib.check(rc, "dfepe_w8pt_pose_fwd")
    return F, residual, epi, w_out, Rt, win, cnt


# ------------------------------------------------------------------------------------------------
# validation summary reductions ("next" row f-2)
# ------------------------------------------------------------------------------------------------
METRIC_THS = (0.0, 0.01, 0.03, 0.05, 0.1, 0.3, 0.5, 1.0, 2.0, 5.0, 10.0, 90.0, 180.0)


def metrics_summary(epi_est: Tensor, epi_gt: Optional[Tensor], err_q: Tensor, err_t: Tensor) -> dict:
    """Device-side reductions of write_metrics_summary (train_good_utils.py:758-856) for one experiment tag: epi_est / epi_gt
    = epipolar distances of every correspondence (any shape, same numel), err_q / err_t [B] pose errors in degrees.
    Two launches and ONE device-to-host copy of 280 bytes.  Returns python floats: ratio_0.1, ratio_1, F1_0.1, F1_1 (None
    without epi_gt), median / max of err_q and err_t, and ratio_q / ratio_t = cumulative fractions below METRIC_THS[1:]."""
    import numpy as np

    e = _prep(epi_est.reshape(-1), "epi_est")
    g = None if epi_gt is None else _prep(epi_gt.reshape(-1), "epi_gt")
    q, t = _prep(err_q.reshape(-1), "err_q"), _prep(err_t.reshape(-1), "err_t")
    if g is not None and g.numel() != e.numel():
        raise ValueError("epi_gt must have as many entries as epi_est")
    if q.numel() != t.numel():
        raise ValueError("err_q and err_t must have the same length")
    L = _lib.lib()
    nbytes = int(L.dfepe_metrics_summary_bytes())
    out = torch.empty((nbytes + 7) // 8, dtype=torch.int64, device=e.device)
    with _on(e.device):
        rc = L.dfepe_metrics_summary(_ptr(e) if e.numel() else None, _ptr(g), e.numel(), _ptr(q), _ptr(t), q.numel(), _ptr(out), _stream())
    _lib.check(rc, "dfepe_metrics_summary")
    raw = out.cpu().numpy().tobytes()[:nbytes]
    counts = np.frombuffer(raw, dtype=np.uint64, count=8).astype(np.float64)
    hist = np.frombuffer(raw, dtype=np.uint64, count=24, offset=64).astype(np.float64).reshape(2, 12)
    mx = np.frombuffer(raw, dtype=np.float32, count=2, offset=64 + 192)
    mids = np.frombuffer(raw, dtype=np.float32, count=4, offset=64 + 192 + 8).reshape(2, 2)
    n, B = max(e.numel(), 1), max(q.numel(), 1)

    def f1(tp, fp, fn):
        return float(2 * tp / (2 * tp + fp + fn)) if (2 * tp + fp + fn) > 0 else 0.0

    res = {"ratio_0.1": float(counts[0] / n), "ratio_1": float(counts[1] / n),
           "F1_0.1": f1(*counts[2:5]) if g is not None else None, "F1_1": f1(*counts[5:8]) if g is not None else None,
           "median_err_q": float(0.5 * (float(mids[0, 0]) + float(mids[0, 1]))), "median_err_t": float(0.5 * (float(mids[1, 0]) + float(mids[1, 1]))),
           "max_err_q": float(mx[0]), "max_err_t": float(mx[1]),
           "ratio_q": (np.cumsum(hist[0]) / B).tolist(), "ratio_t": (np.cumsum(hist[1]) / B).tolist()}
    return res


# ------------------------------------------------------------------------------------------------
# InstanceNorm1d(affine) + LeakyReLU on channel-major activations (weight-estimator fusion, "next" row f-1)
# ------------------------------------------------------------------------------------------------
class _InormLReLUFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, Y, gamma, beta, eps, slope, skipped_bias):
        C, R, N = Y.shape
        A = torch.empty_like(Y)
        stats = torch.empty(C * R, 2, device=Y.device, dtype=torch.float32)
        with _on(Y.device):
            rc = _lib.lib().dfepe_inorm_lrelu_fwd(_ptr(Y), _ptr(gamma), _ptr(beta), C, R, N, float(eps), float(slope), _ptr(A),
                                                  _ptr(stats), _stream())
        _lib.check(rc, "dfepe_inorm_lrelu_fwd")
        ctx.save_for_backward(Y, gamma, beta, stats)
        ctx.slope = float(slope)
        ctx.bias_like = skipped_bias
        return A

    @staticmethod
    def backward(ctx, gA):
        Y, gamma, beta, stats = ctx.saved_tensors
        C, R, N = Y.shape
        gA = gA.contiguous()
        gY = torch.empty_like(Y)
        rg = torch.empty(C, R, device=Y.device, dtype=torch.float32)
        rb = torch.empty(C, R, device=Y.device, dtype=torch.float32)
        with _on(Y.device):
            rc = _lib.lib().dfepe_inorm_lrelu_bwd(_ptr(Y), _ptr(gA), _ptr(gamma), _ptr(beta), _ptr(stats), C, R, N, ctx.slope,
                                                  _ptr(gY), _ptr(rg), _ptr(rb), _stream())
        _lib.check(rc, "dfepe_inorm_lrelu_bwd")
        # the bias of the convolution that feeds this normalisation cancels in it: its gradient is exactly zero, and it is
        # returned as such so that the parameter stays in the graph (DistributedDataParallel, optimizer state, weight decay)
        gb = None if ctx.bias_like is None else torch.zeros_like(ctx.bias_like)
        return gY, rg.sum(dim=1), rb.sum(dim=1), None, None, gb


def inorm_lrelu(Y: Tensor, gamma: Tensor, beta: Tensor, eps: float = 1e-5, slope: float = 0.01,
                skipped_bias: Optional[Tensor] = None) -> Tensor:
    """Y [C,R,N] channel-major (contiguous) -> LeakyReLU(InstanceNorm over N with affine gamma/beta [C]).
    ``skipped_bias``: the [C] bias of the convolution that produced Y without it (a per-channel constant cancels in the
    instance normalisation); it receives an exact zero gradient instead of none."""
    return _InormLReLUFunction.apply(_prep(Y, "Y"), _prep(gamma, "gamma"), _prep(beta, "beta"), eps, slope, skipped_bias)


# ------------------------------------------------------------------------------------------------
# match construction ("next" row f-3): two-way nearest-neighbour matching of descriptors + crop/pad gather
# ------------------------------------------------------------------------------------------------
def nn_match_two_way(desc1: Tensor, desc2: Tensor, nn_thresh: float):
    """desc1 [B,N1,D], desc2 [B,N2,D] unit-norm descriptors -> (m_idx1 [B,N1] int32, m_idx2 [B,N1] int32, score [B,N1],
    count [B] int32): the first count[b] entries of row b are pair b's mutual nearest neighbours closer than ``nn_thresh``
    in increasing m_idx1 order (PointTracker.nn_match_two_way as called at train_good_utils.py:687-691, for all pairs)."""
    if nn_thresh < 0.0:
        raise ValueError("'nn_thresh' should be non-negative")
    d1, d2 = _prep(desc1, "desc1"), _prep(desc2, "desc2")
    if d1.dim() != 3 or d2.dim() != 3 or d1.shape[0] != d2.shape[0] or d1.shape[2] != d2.shape[2]:
        raise ValueError("descriptors must be [B,N1,D] and [B,N2,D]")
    B, N1, D = d1.shape
    N2 = d2.shape[1]
    L = _lib.lib()
    dev = d1.device
    m1 = torch.empty(B, max(N1, 1), device=dev, dtype=torch.int32)
    m2 = torch.empty(B, max(N1, 1), device=dev, dtype=torch.int32)
    sc = torch.empty(B, max(N1, 1), device=dev, dtype=torch.float32)
    cnt = torch.empty(B, device=dev, dtype=torch.int32)
    ws = torch.empty(max(int(L.dfepe_nn_match_workspace_bytes(B, N1, N2)) // 8, 1), device=dev, dtype=torch.int64)
    with _on(dev):
        rc = L.dfepe_nn_match_two_way(_ptr(d1), _ptr(d2), B, N1, N2, D, float(nn_thresh), _ptr(ws), _ptr(m1), _ptr(m2), _ptr(sc),
                                      _ptr(cnt), _stream())
    _lib.check(rc, "dfepe_nn_match_two_way")
    return m1, m2, sc, cnt


def gather_matches(pts1: Tensor, pts2: Tensor, off1: Optional[Tensor], off2: Optional[Tensor], m_idx1: Tensor, m_idx2: Tensor,
                   score: Tensor, choice: Tensor):
    """choice [B,n_out] int32 positions into each pair's match list -> xs [B,n_out,4], offsets [B,n_out,4] | None,
    quality [B,n_out,1]  (train_good_utils.py:698-716)."""
    p1, p2 = _prep(pts1, "pts1"), _prep(pts2, "pts2")
    B, N1, N2 = p1.shape[0], p1.shape[1], p2.shape[1]
    o1 = None if off1 is None else _prep(off1, "off1")
    o2 = None if off2 is None else _prep(off2, "off2")
    if m_idx1.shape != (B, N1) or m_idx1.dtype != torch.int32 or choice.dtype != torch.int32 or not choice.is_contiguous():
        raise ValueError("m_idx1/m_idx2 must be the [B,N1] int32 outputs of nn_match_two_way and choice a contiguous int32 [B,n_out]")
    n_out = choice.shape[1]
    dev = p1.device
    xs = torch.empty(B, n_out, 4, device=dev)
    offs = torch.empty(B, n_out, 4, device=dev) if o1 is not None else None
    q = torch.empty(B, n_out, 1, device=dev)
    with _on(dev):
        rc = _lib.lib().dfepe_gather_matches(_ptr(p1), _ptr(p2), _ptr(o1), _ptr(o2), B, N1, N2, _ptr(m_idx1), _ptr(m_idx2), _ptr(score),
                                             _ptr(choice), n_out, _ptr(xs), _ptr(offs), _ptr(q), _stream())
    _lib.check(rc, "dfepe_gather_matches")
    return xs, offs, q
