"""CapturedStep -- the reference's EAGER training step, replayed as one hipGraph.

The reference's agent runs, per batch (deepFEPE/Train_model_pipeline.py:495-595):

    outs = self.net(data_batch)
    losses_dict, ..., E_ests_layers = get_all_loss_DeepF(outs, pts1_virt_ori, pts2_virt_ori, Ks, loss_params, ...)
    geo_errors_dict = get_Rt_loss(E_ests_layers, ..., delta_Rtijs_4_4, qs_cam, ts_cam, ...)
    loss = clamp(stack(q_l2_error_layers_list)).mean() * balance_q + ...        # the caller's own mixing lines
    self.optimizer.zero_grad(); loss.backward(); self.optimizer.step()

Behind this package's mirrors that sequence is ~50 kernel launches of 5-13 us each; issued one by one from Python it is bound by
the host (~1.7 ms at B = 4096 against 0.24 ms of GPU work).  It is capturable -- no host synchronisation, no allocation outside
the caching allocator -- but train_good.py never builds a graph.  This helper does it for the caller, in three lines:

    step = compat.CapturedStep(forward_and_loss, net)                # forward_and_loss(batch) -> (loss, aux): the lines above
    for batch in loader:                                             # up to, NOT including, loss.backward()
        optimizer.zero_grad()
        loss, aux = step(batch)      # forward + loss + backward; .grad of every parameter is (over)written
        optimizer.step()

The first ``warmup`` calls of a batch signature (shapes, dtypes, non-tensor values) run eagerly -- they are real steps, their
results are returned like any other -- the next one captures forward + loss + backward into a hipGraph over STATIC copies of the
batch's tensors (device copies; host tensors and numpy arrays are uploaded into them before every replay), and from then on a call
is: copy the batch in, replay, hand out the same (static) ``loss`` / ``aux`` objects with their new contents.  A batch with another
signature gets a graph of its own (all graphs share one memory pool).  Contract (the usual one of whole-step graph capture):
  * ``forward_and_loss`` must be a pure function of the batch and the parameters' CURRENT VALUES: no host-side branching on tensor
    values, no .item() / .cpu(); python-side state it mutates is not replayed.
  * the optimizer updates parameters in place (every torch.optim one does); gradients are OVERWRITTEN by every replay, not
    accumulated, and ``optimizer.zero_grad(set_to_none=True)`` is harmless: the step re-attaches its static .grad tensors.
  * every new graph VERIFIES ITSELF before it is trusted (round 6): the capture call runs the step eagerly on the static batch, then
    replays the fresh graph twice and requires loss and every parameter gradient of both replays to equal the eager ones (ROCm 7.2's
    graph packet capture replays a full-model step correctly once and wrongly ever after, see the package __init__); a graph that
    fails is dropped, its signature stays eager and ONE error is logged.  Without the runtime workaround in effect
    (``dfepe.HIP_GRAPH_PACKET_CAPTURE_OFF`` False) the helper does not capture at all unless ``allow_unsafe_graph=True``.
  * ``loss`` / ``aux`` are rewritten by the next call: copy what must outlive it.  Host-side metrics of get_Rt_loss found in
    ``aux`` are lazy while captured (no device synchronisation inside the step); ``step.realise(aux)`` returns a copy with the
    reference's numpy arrays / floats, read from the buffers as they are after the latest replay.
"""
from __future__ import annotations

from typing import Any, Callable, Dict, Iterable, List, Optional, Tuple

import numpy as np
import torch

from .. import _lib
from .. import dist as _dist


def _flatten(obj, path=()):
    """(path, leaf) pairs of a nest of dicts / lists / tuples, in a deterministic order."""
    if isinstance(obj, dict):
        for k in sorted(obj, key=repr):
            yield from _flatten(obj[k], path + (("k", k),))
    elif isinstance(obj, (list, tuple)):
        for i, v in enumerate(obj):
            yield from _flatten(v, path + (("i", i),))
    else:
        yield path, obj


def _rebuild(obj, leaves: Dict[tuple, Any], path=()):
    if isinstance(obj, dict):
        return {k: _rebuild(v, leaves, path + (("k", k),)) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        seq = [_rebuild(v, leaves, path + (("i", i),)) for i, v in enumerate(obj)]
        return seq if isinstance(obj, list) else (type(obj)(*seq) if hasattr(obj, "_fields") else tuple(seq))
    return leaves.get(path, obj)


def _is_array(x) -> bool:
    return torch.is_tensor(x) or isinstance(x, np.ndarray)


def _leaf_key(x):
    if torch.is_tensor(x):
        return ("T", tuple(x.shape), str(x.dtype), bool(x.requires_grad))
    if isinstance(x, np.ndarray):
        return ("N", tuple(x.shape), str(x.dtype))
    try:
        hash(x)
        return ("V", x)
    except TypeError:
        return ("O", id(x))


class _Entry:
    """One captured signature: static inputs, graph, static outputs, static gradients."""

    __slots__ = ("seen", "static", "graph", "out", "grads", "batch", "eager_only")

    def __init__(self):
        self.seen = 0
        self.eager_only = False  # its graph failed the self-check (or graphs are not allowed): every call of this signature runs eagerly
        self.static: Dict[tuple, torch.Tensor] = {}
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.out = None
        self.grads: List[Optional[torch.Tensor]] = []
        self.batch = None


class CapturedStep:
    def __init__(self, forward_and_loss: Callable[[Any], Tuple[torch.Tensor, Any]], parameters=(),
                 device: Optional[torch.device] = None, warmup: int = 2, max_graphs: int = 8, enabled: bool = True,
                 allow_unsafe_graph: bool = False, verify: bool = True):
        """``parameters``: the nn.Module whose parameters the step trains (preferred), or an iterable of leaf tensors.
        With a module the step runs its forward over fresh VIEWS of the parameters (torch.nn.utils.stateless): gradients are then
        taken at nodes created on the capture stream.  With bare leaves they are taken at the leaves' AccumulateGrad nodes, which
        live on the stream of the model's first backward -- if that was the default stream (a step run outside this helper whose
        graph is still alive) the autograd engine's hop to it pulls the legacy stream into the capture and hipStreamEndCapture
        crashes (scripts/capture_probe2.py): pass the module."""
        self.fn = forward_and_loss
        self.module = parameters if isinstance(parameters, torch.nn.Module) else None
        self.params = [p for p in (parameters.parameters() if self.module is not None else parameters)]
        self._names = [n for n, _ in self.module.named_parameters()] if self.module is not None else None
        self.device = torch.device(device) if device is not None else (self.params[0].device if self.params else torch.device("cuda", torch.cuda.current_device()))
        if self.device.type != "cuda":
            raise _lib.DfepeError("CapturedStep: the step runs on the GPU (this package has no CPU path)")
        self.warmup = max(1, int(warmup))  # at least one eager step: it sizes the allocator and runs every lazy initialisation
        self.max_graphs = int(max_graphs)
        self.enabled = bool(enabled)
        self.allow_unsafe_graph = bool(allow_unsafe_graph)
        self.verify = bool(verify)
        self.n_rejected = 0  # graphs dropped by the self-check
        self._complained = False
        self._entries: Dict[tuple, _Entry] = {}
        self._pool = None
        self._stream = torch.cuda.Stream(device=self.device)
        self.n_eager = self.n_captures = self.n_replays = 0

    # ------------------------------------------------------------------------------------------------------------------
    def _signature(self, batch) -> tuple:
        return tuple((path, _leaf_key(x)) for path, x in _flatten(batch))

    def _to_device(self, x):
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(x)
        return x.to(self.device, non_blocking=True)

    def _zero_grads(self):
        for p in self.params:
            if p.requires_grad:  # a frozen parameter's .grad is none of this helper's business
                p.grad = None

    def _run(self, batch):
        """forward + loss + backward.  The parameter gradients are taken with torch.autograd.grad and ASSIGNED to .grad, not
        accumulated by loss.backward(): an AccumulateGrad node runs on the stream it was created on -- the default stream if the
        model ever ran a step outside this helper -- and the engine's hop to that stream and back, harmless in eager mode, pulls the
        legacy default stream into a capture: hipStreamEndCapture then segfaults (scripts/capture_probe2.py).  autograd.grad stays on
        the stream of the forward."""
        if self.module is not None:
            from torch.nn.utils import stateless

            views = [p.view_as(p) for p in self.params]  # non-leaf aliases: their grad_fn is created here, on this stream
            with stateless._reparametrize_module(self.module, dict(zip(self._names, views))):
                loss, aux = self.fn(batch)
            targets = views
        else:
            loss, aux = self.fn(batch)
            targets = self.params
        # only what trains: autograd.grad raises for a target that does not require grad (allow_unused covers unused ones only)
        idx = [i for i, p in enumerate(self.params) if p.requires_grad]
        if idx:
            grads = torch.autograd.grad(loss, [targets[i] for i in idx], allow_unused=True)
            for i, g in zip(idx, grads):
                self.params[i].grad = g
        return loss, aux

    def _eager(self, batch):
        """One ordinary step on the helper's stream (the stream the capture will use: the caching allocator then recycles the
        warm-up's blocks for the capture, and the parameters' gradient accumulators are bound to one stream throughout)."""
        cur = torch.cuda.current_stream(self.device)
        self._stream.wait_stream(cur)
        with torch.cuda.stream(self._stream):
            dev_batch = _rebuild(batch, {path: self._to_device(x) for path, x in _flatten(batch) if _is_array(x)})
            out = self._run(dev_batch)
        cur.wait_stream(self._stream)
        self.n_eager += 1
        return out

    def _complain(self, msg: str):
        """One error-level message per helper: a training loop must not scroll it, and must not miss it."""
        if not self._complained:
            import logging

            logging.getLogger("dfepe.CapturedStep").error(msg)
            self._complained = True

    def _graphs_allowed(self) -> bool:
        import sys

        pkg = sys.modules[__name__.rsplit(".", 2)[0]]
        if getattr(pkg, "HIP_GRAPH_PACKET_CAPTURE_OFF", False) or self.allow_unsafe_graph:
            return True
        self._complain("CapturedStep: DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 is not in effect (the HIP runtime was initialised before this package "
                       "was imported, or the environment sets another value); on ROCm 7.2 a captured step of the full model then replays "
                       "with wrong parameter gradients from its second launch on -- staying on the eager path "
                       "(allow_unsafe_graph=True captures anyway, behind the self-check)")
        return False

    @staticmethod
    def _same(a: Optional[torch.Tensor], b: Optional[torch.Tensor]) -> bool:
        """Equal as a replay of the same kernels on the same inputs must be: bit for bit, or -- a caller's own atomics may reorder
        sums -- to 1e-5 of the largest entry.  What the runtime bug leaves behind is stale or unwritten memory: nowhere near."""
        if a is None or b is None:
            return a is None and b is None
        if a.shape != b.shape:
            return False
        if torch.equal(a, b):
            return True
        a64, b64 = a.detach().double(), b.detach().double()
        return bool(torch.isfinite(a64).all()) and float((a64 - b64).abs().max()) <= 1e-5 * float(b64.abs().max()) + 1e-30

    def _capture(self, ent: _Entry, batch) -> bool:
        """Capture the step for this entry.  Returns False (entry marked eager-only, ent.out = the eager result on this batch) when
        graphs are not allowed or the new graph fails its self-check."""
        cur = torch.cuda.current_stream(self.device)
        self._stream.wait_stream(cur)
        with torch.cuda.stream(self._stream):
            for path, x in _flatten(batch):
                if _is_array(x):
                    src = self._to_device(x)
                    buf = torch.empty_like(src).copy_(src)
                    if torch.is_tensor(x) and x.requires_grad:
                        buf.requires_grad_(True)
                    ent.static[path] = buf
            ent.batch = _rebuild(batch, ent.static)
            ref = None
            if self.verify:  # the yardstick: this very step, eagerly, on the static batch
                self._zero_grads()
                ref_loss, _ = self._run(ent.batch)
                ref = (ref_loss.detach().clone(), [None if p.grad is None else p.grad.detach().clone() for p in self.params])
            self._zero_grads()  # the captured backward then WRITES each gradient into memory of the graph's pool
        torch.cuda.synchronize(self.device)
        ent.graph = torch.cuda.CUDAGraph()
        # (get_Rt_loss hands out lazy host metrics by itself while its stream is being captured: no module-global is flipped here --
        # round 5 toggled train_good_utils.LAZY_HOST_METRICS around the capture, under the feet of other threads)
        with torch.cuda.graph(ent.graph, pool=self._pool, stream=self._stream, capture_error_mode=_dist.graph_capture_mode()):
            ent.out = self._run(ent.batch)
        if self._pool is None:
            self._pool = ent.graph.pool()
        ent.grads = [p.grad for p in self.params]
        cur.wait_stream(self._stream)  # the static inputs were filled on the helper's stream; the replays run on the caller's
        self.n_captures += 1
        if ref is not None:
            ok = True
            for _ in range(3 if self.allow_unsafe_graph else 2):  # the packet-capture bug shows from the SECOND launch of a graph on
                ent.graph.replay()
                torch.cuda.synchronize(self.device)
                ok = ok and self._same(ent.out[0], ref[0]) and all(self._same(g, r) for g, r in zip(ent.grads, ref[1]))
            if not ok:
                self.n_rejected += 1
                self._complain("CapturedStep: the captured step does not reproduce the eager step on its own batch (loss or parameter "
                               "gradients differ after a replay) -- the graph is dropped and this batch signature stays on the eager path")
                ent.graph, ent.grads, ent.out = None, [], None
                ent.eager_only = True
                return False
        return True

    def _replay(self, ent: _Entry, batch, fresh: bool):
        # on the CALLER's current stream: a captured graph replays on any stream, and a hop to the helper's stream and back would
        # cost two cross-stream edges per step (18-30 us each on this stack) for nothing
        if not fresh:  # the capture call has just filled the static inputs with this very batch
            dsts, srcs = [], []
            for path, x in _flatten(batch):
                if _is_array(x):
                    dst = ent.static[path].detach()
                    if torch.is_tensor(x) and x.data_ptr() == dst.data_ptr():
                        continue  # the caller filled the static buffer itself (step.static_inputs(batch))
                    dsts.append(dst)
                    srcs.append(self._to_device(x))
            if dsts:
                try:
                    torch._foreach_copy_(dsts, srcs, non_blocking=True)  # one multi-tensor launch instead of one per tensor
                except (AttributeError, RuntimeError, TypeError):
                    for dst, src in zip(dsts, srcs):
                        dst.copy_(src, non_blocking=True)
        ent.graph.replay()
        for p, g in zip(self.params, ent.grads):
            if p.requires_grad:
                p.grad = g  # survives the caller's zero_grad(set_to_none=True)
        self._refresh(ent.out[1])
        self.n_replays += 1
        return ent.out

    @staticmethod
    def _refresh(aux):
        """Forget the cached host copies of every get_Rt_loss dict inside ``aux``: the replay rewrote the device buffers."""
        stack = [aux]
        while stack:
            x = stack.pop()
            hm = getattr(x, "host_metrics", None)
            if hm is not None and hasattr(hm, "refresh"):
                hm.refresh()
            if isinstance(x, dict):
                stack.extend(x.values())
            elif isinstance(x, (list, tuple)):
                stack.extend(x)

    # ------------------------------------------------------------------------------------------------------------------
    def __call__(self, batch):
        if not self.enabled:
            return self._eager(batch)
        key = self._signature(batch)
        ent = self._entries.get(key)
        if ent is None:
            if len(self._entries) >= self.max_graphs:  # a stream of ever-changing shapes: stay eager instead of hoarding graphs
                return self._eager(batch)
            ent = self._entries[key] = _Entry()
        if ent.graph is None:
            if ent.eager_only or ent.seen < self.warmup:
                ent.seen += 1
                return self._eager(batch)
            if not self._graphs_allowed():
                ent.eager_only = True
                return self._eager(batch)
            if not self._capture(ent, batch):
                return self._eager(batch)
            return self._replay(ent, batch, fresh=True)
        return self._replay(ent, batch, fresh=False)

    def static_inputs(self, batch):
        """The static input nest of the graph that serves ``batch``'s signature (None before its capture): a loader that writes the
        next batch straight into these tensors and passes them back saves the copy-in."""
        ent = self._entries.get(self._signature(batch))
        return None if ent is None or ent.graph is None else ent.batch

    @staticmethod
    def realise(aux):
        """A copy of ``aux`` in which every get_Rt_loss dict (``aux`` itself may be one) carries the reference's host types -- numpy
        arrays / python floats read from the device buffers as they are NOW (one device synchronisation).  The step's own ``aux``
        stays lazy: it is static, the next replay rewrites the buffers behind it."""
        def conv(x):
            if hasattr(x, "realised") and hasattr(x, "host_metrics"):
                return x.realised()
            if isinstance(x, dict):
                return type(x)((k, conv(v)) for k, v in x.items())
            if isinstance(x, (list, tuple)) and not hasattr(x, "_fields"):
                return type(x)(conv(v) for v in x)
            return x
        return conv(aux)
