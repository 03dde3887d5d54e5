"""The hot path as one callable: depth x (softmax -> weighted 8-point fit [+ in-loop epipolar residual]),
F-loss on the virtual points, E-from-F, pose loss — forward and backward to the logits.

This is the "solver-only" step of SURVEY.md §8d (C3/C4): the per-layer logits are given tensors (in the
full model they come from the stock-PyTorch ErrorEstimator, which is outside the hot path).  The same
function is what bench.py times and what the parity tests compare with the CPU oracle's hot_path_step.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import torch

from . import _lib, ops

Tensor = torch.Tensor


def hot_path_forward(matches: Tensor, logits_layers: Tensor, Ks: Tensor, virt1: Tensor, virt2: Tensor,
                     q_gt: Tensor, t_gt: Tensor, R_gt: Tensor, image_size: Sequence[int], clamp_at: float = 0.02,
                     qt: bool = True, clamp_q: float = 0.1, clamp_t: float = 0.5, balance_q: float = 1.0,
                     balance_t: float = 0.1, hw_T: Optional[Tensor] = None, balance_F: float = 1.0) -> Dict[str, Tensor]:
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
    loss = loss_F * balance_F
    if qt:
        q_l2, t_l2, R_deg, t_deg, sel = ops.pose_errors(E_layers, q_gt, t_gt, R_gt)
        loss_qt = torch.clamp(q_l2, 0.0, clamp_q).mean() * balance_q + torch.clamp(t_l2, 0.0, clamp_t).mean() * balance_t
        out.update({"q_l2": q_l2, "t_l2": t_l2, "R_deg": R_deg, "t_deg": t_deg, "sel": sel, "loss_qt": loss_qt})
        loss = loss + loss_qt
    out["loss"] = loss
    return out


_exchange_streams = {}


def _exchange_stream(dev):
    """One side stream per device for the loss head + exchange branch."""
    key = str(dev)
    if key not in _exchange_streams:
        _exchange_streams[key] = torch.cuda.Stream(device=dev)
    return _exchange_streams[key]


def _tail_workspace(dev, B: int) -> Tensor:
    """Scratch of dfepe_loss_tail (per-workgroup partial sums, and the descriptor of a deferred head): ~100 KB from the caching
    allocator per call, so that it belongs to the stream / graph-capture pool the call runs in (a cached buffer shared between
    an eager call and a captured graph, or two graphs replayed on different streams, would race on the partial sums)."""
    return torch.empty((_lib.lib().dfepe_loss_tail_workspace_bytes(B) + 7) // 8, dtype=torch.float64, device=dev)


class _HotPathFunction(torch.autograd.Function):
    """The whole solver-only step as ONE autograd node.  Forward: L x w8pt_fwd (softmax fused) + ONE loss-tail launch that
    also forms d loss / d F of every layer (dfepe_loss_tail); backward: L x w8pt_bwd, which apply the upstream gradient of
    the loss as their g_scale.  12 launches for L = 5 (5 fits, tail, head, 5 adjoints), no intermediate torch ops, no per-op autograd bookkeeping.
    ``fused_tail=False`` keeps the round-1 structure (floss_fwd, pose_fwd, loss_head | pose_bwd, floss_bwd: 15 launches)."""

    @staticmethod
    def forward(ctx, matches, logits_layers, Ks, virt1, virt2, q_gt, t_gt, R_gt, hw_T, cfg):
        (H, W, clamp_at, qt, clamp_q, clamp_t, balance_q, balance_t, batched, balance_F, fused_tail, grad_pairs, defer_head, exchange) = cfg
        ctx.set_materialize_grads(False)  # 13 auxiliary outputs: do not let autograd zero-fill [L,B,N] gradients for them
        lib = _lib.lib()
        L, B, N = logits_layers.shape
        dev = matches.device
        M = virt1.shape[1]
        if L > _lib.TAIL_MAX_LAYERS:  # before any launch: every loss kernel (fused tail, dfepe_floss_*, dfepe_loss_head) stacks <= 16 layers
            raise _lib.DfepeError(f"depth {L} > {_lib.TAIL_MAX_LAYERS} layers per loss launch (the reference's configs use depth 5)")
        fused_tail = fused_tail and M <= 128  # more virtual points than a row holds: the five-kernel tail serves any M
        F_layers = torch.empty(L, B, 3, 3, device=dev)
        residuals = torch.empty(L, B, N, device=dev)
        epis = torch.empty(L, B, N, device=dev)
        weights = torch.empty(L, B, N, device=dev)
        saves = torch.empty(L, B, lib.dfepe_save_floats(), device=dev)
        flags = _lib.W8PT_RAW_MATCHES | _lib.W8PT_LOGITS
        st = ops._stream()
        gF = None
        with ops._on(dev):
            if batched:  # the L weightings of the same pairs in ONE launch (n_weight_sets = L): no per-layer launch tails
                rc = lib.dfepe_w8pt_fwd(matches.data_ptr(), None, logits_layers.data_ptr(), B, N, L, flags, W, H, 0.5,
                                        F_layers.data_ptr(), residuals.data_ptr(), epis.data_ptr(), saves.data_ptr(),
                                        weights.data_ptr(), st)
                _lib.check(rc, "dfepe_w8pt_fwd")
            else:
                for l in range(L):
                    rc = lib.dfepe_w8pt_fwd(matches.data_ptr(), None, logits_layers[l].data_ptr(), B, N, 1, flags, W, H, 0.5,
                                            F_layers[l].data_ptr(), residuals[l].data_ptr(), epis[l].data_ptr(),
                                            saves[l].data_ptr(), weights[l].data_ptr(), st)
                    _lib.check(rc, "dfepe_w8pt_fwd")
            loss_sum = torch.empty(L, B, device=dev)
            E_layers = torch.empty(L, B, 3, 3, device=dev)
            q_l2 = t_l2 = R_deg = t_deg = sel = None
            if qt:
                q_l2 = torch.empty(L, B, device=dev)
                t_l2 = torch.empty(L, B, device=dev)
                R_deg = torch.empty(L, B, device=dev)
                t_deg = torch.empty(L, B, device=dev)
                sel = torch.empty(L, B, device=dev, dtype=torch.int32)
            packed = torch.empty(L + 4, device=dev, dtype=torch.float64)
            scalars = torch.empty(4 + L, device=dev)
            if fused_tail:
                gF = torch.empty(L, B, 3, 3, device=dev)
                # deferred loss head: packed / scalars are finished by the first backward launch (off the critical path); the
                # workspace then carries the head's descriptor from here to there, so it belongs to this call alone
                will_backward = bool(ctx.needs_input_grad[1])
                branch = exchange is not None and exchange[1] is not None and will_backward  # head + exchange on a side stream
                defer = bool((defer_head and will_backward) or branch)
                ws = _tail_workspace(dev, B)
                ctx.pending_ws = ws if (defer and not branch) else None
                rc = lib.dfepe_loss_tail(F_layers.data_ptr(), L, B, hw_T.data_ptr(), hw_T.data_ptr(), 0, Ks.data_ptr(), virt1.data_ptr(),
                                         virt2.data_ptr(), M, clamp_at, ops._ptr(q_gt if qt else None), ops._ptr(t_gt if qt else None),
                                         ops._ptr(R_gt if qt else None), clamp_q, clamp_t, balance_F, balance_q, balance_t,
                                         float(grad_pairs if grad_pairs else B), loss_sum.data_ptr(), E_layers.data_ptr(), ops._ptr(q_l2),
                                         ops._ptr(t_l2), ops._ptr(R_deg), ops._ptr(t_deg), ops._ptr(sel), gF.data_ptr(),
                                         packed.data_ptr(), scalars.data_ptr(), ws.data_ptr(), 1 if defer else 0, st)
                _lib.check(rc, "dfepe_loss_tail")
                ctx.join_stream = None
                if branch:
                    # fork: tail -> [head -> exchange(packed)] on the exchange stream  ||  [L x w8pt_bwd] on this one; the backward
                    # joins.  Nothing in the backward needs the batch scalars, so the ~5 us head and the latency of the collective
                    # leave the step's critical path; inside a hipGraph capture the two become parallel branches of the graph.
                    fn, side = exchange
                    main = torch.cuda.current_stream()
                    side.wait_stream(main)
                    with torch.cuda.stream(side):
                        _lib.check(lib.dfepe_loss_head_pending(ws.data_ptr(), side.cuda_stream), "dfepe_loss_head_pending")
                        fn(packed)
                    ctx.join_stream = side
                    ctx.keep = (ws, packed, scalars)  # used on the side stream: alive until the join
                    if not torch.cuda.is_current_stream_capturing():
                        # should the backward (= the join) never run -- loss discarded, an exception in between -- ctx dies and
                        # the caching allocator must not hand these blocks out again on the main stream while the head or the
                        # collective still uses them on the side stream.  (Inside a capture the graph's pool owns the memory, and
                        # a forward captured without its backward leaves the fork unjoined: the capture fails loudly.)
                        for t in ctx.keep:
                            t.record_stream(side)
                elif exchange is not None and will_backward:  # in stream order behind the last backward fit (the head may ride in the first)
                    ctx.exchange_after = (exchange[0], packed)
                elif exchange is not None:  # forward only: head already ran in this stream (defer = 0); exchange in stream order
                    exchange[0](packed)
            else:
                rc = lib.dfepe_floss_fwd(F_layers.data_ptr(), L, B, hw_T.data_ptr(), hw_T.data_ptr(), 0, Ks.data_ptr(), virt1.data_ptr(),
                                         virt2.data_ptr(), M, clamp_at, loss_sum.data_ptr(), E_layers.data_ptr(), st)
                _lib.check(rc, "dfepe_floss_fwd")
                if qt:
                    rc = lib.dfepe_pose_fwd(E_layers.data_ptr(), L, B, q_gt.data_ptr(), t_gt.data_ptr(), R_gt.data_ptr(), q_l2.data_ptr(),
                                            t_l2.data_ptr(), R_deg.data_ptr(), t_deg.data_ptr(), sel.data_ptr(), st)
                    _lib.check(rc, "dfepe_pose_fwd")
                rc = lib.dfepe_loss_head(loss_sum.data_ptr(), ops._ptr(q_l2), ops._ptr(t_l2), L, B, M, clamp_q, clamp_t, balance_q,
                                         balance_t, packed.data_ptr(), scalars.data_ptr(), st)
                _lib.check(rc, "dfepe_loss_head")
                if balance_F != 1.0:  # dfepe_loss_head mixes loss_F + loss_qt; any other balance is one more tiny op here
                    scalars[0:1].copy_(scalars[1:2] * balance_F + (scalars[2:3] if qt else 0.0))
                if exchange is not None:
                    exchange[0](packed)
        saved = [matches, weights, Ks, virt1, virt2, q_gt, t_gt, hw_T, F_layers, E_layers, saves]
        if gF is not None:
            saved.append(gF)
        ctx.save_for_backward(*saved)
        ctx.cfg = cfg
        ctx.fused_tail = fused_tail
        extras = (F_layers, residuals, epis, weights, E_layers, loss_sum, packed, scalars)
        pose = (q_l2, t_l2, R_deg, t_deg, sel) if qt else ()
        ctx.mark_non_differentiable(*extras, *pose)
        return (scalars[0:1].view(()),) + extras + pose

    @staticmethod
    def backward(ctx, g_loss, *unused):
        matches, weights, Ks, virt1, virt2, q_gt, t_gt, hw_T, F_layers, E_layers, saves = ctx.saved_tensors[:11]
        (H, W, clamp_at, qt, clamp_q, clamp_t, balance_q, balance_t, batched, balance_F, _ft, grad_pairs, _dh, _ex) = ctx.cfg
        lib = _lib.lib()
        L, B, N = weights.shape
        M = virt1.shape[1]
        dev = matches.device
        if g_loss is None:
            return (None,) * 10
        g_scale = g_loss.reshape(1).contiguous().float()
        st = ops._stream()
        flags = _lib.W8PT_RAW_MATCHES | _lib.W8PT_LOGITS
        g_logits = torch.empty(L, B, N, device=dev)
        n = float(grad_pairs if grad_pairs else B)
        with ops._on(dev):
            if ctx.fused_tail:
                gF = ctx.saved_tensors[11]  # d loss / d F with unit upstream; w8pt_bwd applies g_scale
                gs_ptr = g_scale.data_ptr()
            else:
                gF = torch.empty(L, B, 3, 3, device=dev)
                gs_ptr = None
                gE_ptr = None
                if qt:
                    gE = torch.empty(L, B, 3, 3, device=dev)
                    rc = lib.dfepe_pose_bwd(E_layers.data_ptr(), L, B, q_gt.data_ptr(), t_gt.data_ptr(), None, None,
                                            balance_q / (L * n), clamp_q, balance_t / (L * n), clamp_t, g_scale.data_ptr(),
                                            gE.data_ptr(), st)
                    _lib.check(rc, "dfepe_pose_bwd")
                    gE_ptr = gE.data_ptr()
                rc = lib.dfepe_floss_bwd(F_layers.data_ptr(), L, B, hw_T.data_ptr(), hw_T.data_ptr(), 0, Ks.data_ptr(), virt1.data_ptr(),
                                         virt2.data_ptr(), M, clamp_at, None, balance_F / (L * n * M), g_scale.data_ptr(), gE_ptr,
                                         gF.data_ptr(), st)
                _lib.check(rc, "dfepe_floss_bwd")
            pend = getattr(ctx, "pending_ws", None)
            pend_ptr = pend.data_ptr() if pend is not None else None  # handed to exactly one backward launch
            if batched:
                rc = lib.dfepe_w8pt_bwd(matches.data_ptr(), None, weights.data_ptr(), B, N, L, flags, W, H, 0.5, saves.data_ptr(),
                                        F_layers.data_ptr(), gF.data_ptr(), None, None, None, gs_ptr, g_logits.data_ptr(), None, None,
                                        pend_ptr, st)
                _lib.check(rc, "dfepe_w8pt_bwd")
            else:
                for l in range(L):
                    rc = lib.dfepe_w8pt_bwd(matches.data_ptr(), None, weights[l].data_ptr(), B, N, 1, flags, W, H, 0.5,
                                            saves[l].data_ptr(), F_layers[l].data_ptr(), gF[l].data_ptr(), None, None, None, gs_ptr,
                                            g_logits[l].data_ptr(), None, None, pend_ptr if l == 0 else None, st)
                    _lib.check(rc, "dfepe_w8pt_bwd")
            join = getattr(ctx, "join_stream", None)
            if join is not None:  # the head / exchange branch meets the backward here
                torch.cuda.current_stream().wait_stream(join)
                ctx.join_stream = ctx.keep = None
            after = getattr(ctx, "exchange_after", None)
            if after is not None:  # the step's last enqueue: by now the head (riding in the first backward launch) has finished packed
                after[0](after[1])
                ctx.exchange_after = None
        return None, g_logits, None, None, None, None, None, None, None, None


def hot_path_fused(matches: Tensor, logits_layers: Tensor, Ks: Tensor, virt1: Tensor, virt2: Tensor, q_gt: Tensor,
                   t_gt: Tensor, R_gt: Tensor, image_size: Sequence[int], clamp_at: float = 0.02, qt: bool = True,
                   clamp_q: float = 0.1, clamp_t: float = 0.5, balance_q: float = 1.0, balance_t: float = 0.1,
                   hw_T: Optional[Tensor] = None, layers_batched: bool = False, balance_F: float = 1.0, fused_tail: bool = True,
                   grad_pairs: Optional[int] = None, defer_loss_head: bool = False, loss_exchange=None,
                   exchange_branch: bool = False, exchange_stream: Optional["torch.cuda.Stream"] = None) -> Dict[str, Tensor]:
    """Same contract and same numbers as hot_path_forward, 12 kernel launches instead of ~120:
    loss = balance_F * loss_F + loss_qt.  The reference's pipeline drops the F-loss from the objective when if_qt_loss
    (Train_model_pipeline.py:580-587, `loss += loss_F * balance_F` commented out): that is balance_F = 0; the solver-only
    benchmark step keeps both terms (BASELINE metric "F+E+pose+loss") with balance_F = 1.
    ``layers_batched`` fits all L weightings in one launch (legal only because the per-layer logits are given; in the
    real recurrent model each layer's logits depend on the previous fit).  ``grad_pairs``: the number of pairs the batch
    means run over in the gradient (the global batch under data parallelism; default B).
    ``defer_loss_head``: the batch sums behind ``loss`` / ``loss_F`` / ``loss_qt`` / ``loss_layers`` / ``packed`` are finished
    by the first backward launch instead of a launch of their own (11 launches, the ~7 us head off the critical path).  Those
    tensors are then valid only AFTER ``loss.backward()`` -- for steps that run forward and backward back to back (a captured
    graph); everything per pair (F, E, loss_sum, pose errors) and the gradients do not depend on it.
    ``loss_exchange``: a callable applied to ``packed`` -- the data-parallel all-reduce of the L+4 loss sums,
    ``lambda p: torch.distributed.all_reduce(p)`` -- as PART of the step, so that a captured step carries its collective in its
    hipGraph (RCCL collectives are capturable) and the host enqueues nothing per step but the replay.  Default placement: in
    stream order behind the last backward fit (forward only: behind the loss head).  ``exchange_branch=True`` instead forks it
    onto ``exchange_stream`` right behind the loss head, parallel to the backward fits, which join it at their end:
    tail -> [head -> all_reduce] || [L x w8pt_bwd].  Measured on this stack (scripts/exchange_probe.py, one-rank RCCL, B = 4096):
    the in-order node costs nothing measurable, the branch +33 us per step (cross-stream edges of a hipGraph, like stream-event
    waits outside one, cost 18-30 us each here -- more than the 72-byte collective they would hide), so the branch is the option,
    not the default.  Either way ``packed`` holds the reduced sums after the backward."""
    L, B, N = logits_layers.shape
    H, W = float(image_size[0]), float(image_size[1])
    dev = matches.device
    if hw_T is None:
        hw_T = torch.tensor([[2.0 / W, 0.0, -1.0], [0.0, 2.0 / H, -1.0], [0.0, 0.0, 1.0]], device=dev)
    f32 = lambda t: ops._prep(t, "input")
    exchange = None
    if loss_exchange is not None:
        if not fused_tail:
            raise _lib.DfepeError("loss_exchange rides behind the fused loss tail (fused_tail=True)")
        exchange = (loss_exchange, (exchange_stream if exchange_stream is not None else _exchange_stream(dev)) if exchange_branch else None)
    cfg = (H, W, float(clamp_at), bool(qt), float(clamp_q), float(clamp_t), float(balance_q), float(balance_t), bool(layers_batched),
           float(balance_F), bool(fused_tail), grad_pairs, bool(defer_loss_head), exchange)
    res = _HotPathFunction.apply(f32(matches), f32(logits_layers), f32(Ks), f32(virt1), f32(virt2), f32(q_gt.reshape(B, 4)),
                                 f32(t_gt.reshape(B, 3)), f32(R_gt.reshape(B, 3, 3)), f32(hw_T), cfg)
    loss, F_layers, residuals, epis, weights, E_layers, loss_sum, packed, scalars = res[:9]
    M = virt1.shape[1]
    out = {"loss": loss, "F_layers": F_layers, "residual_layers": [residuals[l] for l in range(L)],
           "epi_res_layers": [epis[l] for l in range(L)], "weights_layers": [weights[l] for l in range(L)],
           "E_layers": E_layers, "loss_sum": loss_sum, "loss_layers": scalars[4:4 + L], "loss_F": scalars[1],
           "packed": packed}
    if qt:
        q_l2, t_l2, R_deg, t_deg, sel = res[9:]
        out.update({"q_l2": q_l2, "t_l2": t_l2, "R_deg": R_deg, "t_deg": t_deg, "sel": sel, "loss_qt": scalars[2]})
    return out


def scene_to_device(scene: Dict[str, Tensor], device) -> Dict[str, Tensor]:
    """Move a synth.make_scene() batch to the GPU and derive the camera-motion rotation the pose loss needs
    (R_gt = inv(delta)[:3,:3] = R^T, train_good_utils.py:134,170)."""
    d = {k: v.to(device) for k, v in scene.items()}
    d["R_gt"] = d["delta_Rtijs_4_4"][:, :3, :3].transpose(1, 2).contiguous()
    return d


def hot_path_step(scene: Dict[str, Tensor], image_size: Sequence[int], depth: int, clamp_at: float = 0.02,
                  qt: bool = True, backward: bool = True, fused: bool = True, **kw) -> Dict[str, Tensor]:
    logits = scene["logits_layers"][:depth].detach().clone().requires_grad_(backward)
    out = (hot_path_fused if fused else hot_path_forward)(scene["matches_xy_ori"], logits, scene["Ks"], scene["pts1_virt_ori"], scene["pts2_virt_ori"],
                           scene["qs_cam"], scene["ts_cam"], scene["R_gt"], image_size, clamp_at, qt, **kw)
    if backward:
        out["loss"].backward()
        out["grad_logits"] = logits.grad
    return out


# ------------------------------------------------------------------------------------------------------------------------
# the same step through the reference's OWN call sequence (what train_good.py's agent runs, Train_model_pipeline.py:495-595):
#     outs = net(data_batch); get_all_loss_DeepF(outs, ...); get_Rt_loss(E_ests_layers, ...); caller-side clamp / balance; backward
# with the weight estimator replaced by given per-layer logits, so that what is measured / compared is the solver path behind
# that API and nothing else.  bench.py times it next to hot_path_fused ("api_path"), tests/test_api_path_gpu.py requires the two
# to agree.
# ------------------------------------------------------------------------------------------------------------------------
class FixedLogitsEstimator(torch.nn.Module):
    """Stand-in for compat.ErrorEstimators.ErrorEstimator: call k returns rows[k mod len(rows)] ([B,1,N] logits) whatever its input."""

    def __init__(self, rows):
        super().__init__()
        self.rows = list(rows)
        self.k = 0

    def forward(self, data):
        r = self.rows[self.k % len(self.rows)]
        self.k += 1
        return r


class LinearProbeEstimator(FixedLogitsEstimator):
    """Given logits plus a fixed linear read-out of the three recurrent input channels (weights, in-loop epipolar residual,
    residual: channels 4..6 of update_weights' input, DeepFNet.py:484-489).  Not a model of anything -- the cheapest estimator
    through which gradients reach residual / epi_res / weights of the previous fit, i.e. one that makes every layer but the last
    run the backward the real recurrent model runs (w8pt_bwd with g_residual, g_epi and the weights gradient)."""

    def __init__(self, rows, coef=(0.5, -2.0, 40.0), first_channel=4):
        super().__init__(rows)
        self.coef = tuple(float(c) for c in coef)
        self.first = int(first_channel)  # 4 + quality channels

    def forward(self, data):
        base = super().forward(data)
        if getattr(self, "_c", None) is None or self._c.device != data.device:  # one host copy, on the first (eager) call
            self._c = data.new_tensor(self.coef).view(1, 3, 1)
        return base + (data[:, self.first:self.first + 3, :] * self._c).sum(dim=1, keepdim=True)


def make_api_net(depth: int, image_size: Sequence[int], logits_rows: Sequence[Tensor], recurrent_probe: bool = False):
    """compat.DeepFNet whose two estimators hand out ``logits_rows`` (depth tensors [B,1,N]; leaves that require grad give
    d loss / d logits): layer 0 from input_weights, layers 1.. from the successive update_weights calls (DeepFNet.py:441,510).
    ``recurrent_probe``: update_weights is a LinearProbeEstimator, so the step has the recurrent model's backward shape."""
    from .compat.DeepFNet import DeepFNet

    net = DeepFNet(depth=depth, image_size=image_size, if_quality=False)
    net.input_weights = FixedLogitsEstimator(logits_rows[:1])
    net.update_weights = (LinearProbeEstimator if recurrent_probe else FixedLogitsEstimator)(logits_rows[1:] if depth > 1 else logits_rows[:1])
    return net


def reference_call_sequence(net, scene: Dict[str, Tensor], depth: int, clamp_at: float = 0.02, clamp_q: float = 0.1, clamp_t: float = 0.5,
                            balance_q: float = 1.0, balance_t: float = 0.1, balance_F: float = 1.0, pose_gt_in_loss_params: bool = False,
                            get_residual_summaries: bool = False):
    """One forward of the reference's training step (Train_model_pipeline.py:495-586) on a synth.make_scene batch that lives on
    the GPU.  Returns (loss, outs, losses_dict, geo_errors_dict).  ``balance_F = 0`` is the reference's if_qt_loss objective
    (the F-loss evaluated, not added, :587).  ``pose_gt_in_loss_params``: hand the ground truth to get_all_loss_DeepF as well
    (loss_params["pose_gt"]), which fuses the pose errors into its launch."""
    from .compat import train_good_utils as tgu

    batch = {"matches_xy_ori": scene["matches_xy_ori"], "matches_good_unique_nums": None, "t_scene_scale": None}
    loss_params = {"model": "GoodCorresNet_layers_deepF", "clamp_at": clamp_at, "depth": depth, "good_num": scene["matches_xy_ori"].shape[1],
                   "if_img_feat": False, "matches_good_unique_nums": None, "topK": 8, "if_sample_loss": False, "if_tri_depth": False}
    if pose_gt_in_loss_params:
        loss_params["pose_gt"] = (scene["qs_cam"], scene["ts_cam"], scene["delta_Rtijs_4_4"])
        if balance_F == 0.0:
            loss_params["floss_grad"] = False
    for est in (net.input_weights, net.update_weights):
        if isinstance(est, FixedLogitsEstimator):
            est.k = 0
    outs = net(batch)
    losses_dict, E_ests, F_ests, logits_weights, _, _, E_ests_layers = tgu.get_all_loss_DeepF(
        outs, scene["pts1_virt_ori"], scene["pts2_virt_ori"], scene["Ks"], loss_params, get_residual_summaries=get_residual_summaries)
    geo = tgu.get_Rt_loss(E_ests_layers, scene["Ks"], None, None, scene["delta_Rtijs_4_4"], scene["qs_cam"], scene["ts_cam"],
                          device=scene["matches_xy_ori"].device)
    # the caller's own lines (Train_model_pipeline.py:580-586)
    loss_q = torch.clamp(torch.stack(geo["q_l2_error_layers_list"]), 0.0, clamp_q).mean()
    loss_t = torch.clamp(torch.stack(geo["t_l2_error_layers_list"]), 0.0, clamp_t).mean()
    loss = loss_q * balance_q + loss_t * balance_t
    if balance_F != 0.0:
        loss = loss + losses_dict["loss_F"] * balance_F
    return loss, outs, losses_dict, geo
