"""Data-parallel host logic: one process per GPU, the batch of pairs is split into contiguous shards, every pair is
solved locally (no data-path collective: all reductions of the hot path are over correspondences *inside* a pair),
and one small all-reduce (RCCL over xGMI; gloo in the CPU tests) turns the per-rank loss sums into global means.

The reference has no distributed path (only nn.DataParallel, deepFEPE/train_good.py:311-312); this replaces it with
the `torch.distributed` equivalent for the loss bookkeeping of get_all_loss_DeepF (train_good_utils.py:340-364) and of
the qt loss mixing (Train_model_pipeline.py:580-586).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
import torch.distributed as dist

Tensor = torch.Tensor


def graph_capture_mode() -> str:
    """``capture_error_mode`` for ``torch.cuda.graph`` in a process that holds an RCCL process group: "thread_local".  The group's
    watchdog thread polls the events of earlier collectives with hipEventQuery; in the default "global" mode that call is illegal
    while ANY thread captures, the watchdog dies of "operation not permitted when stream is capturing" and takes the process with
    it (seen in round 6 as `Fatal Python error: Aborted` in the middle of a capture that held an all-reduce -- a race: the poll has
    to fall inside the capture).  Without a process group: "global", torch's default."""
    return "thread_local" if (dist.is_available() and dist.is_initialized()) else "global"


def shard_range(B: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [start, stop) of rank's pairs; the first B % world ranks get one extra pair."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    base, rem = divmod(B, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


_PER_LAYER_KEYS = ("logits_layers",)


def shard_scene(scene: Dict[str, Tensor], rank: int, world: int) -> Dict[str, Tensor]:
    """Slice every per-pair tensor of a synth.make_scene() batch ([B,...], or [L,B,...] for the per-layer logits)."""
    B = scene["matches_xy_ori"].shape[0]
    a, b = shard_range(B, rank, world)
    out = {}
    for k, v in scene.items():
        out[k] = (v[:, a:b] if k in _PER_LAYER_KEYS else v[a:b]).contiguous()
    return out


def pack_loss_sums(loss_sum: Tensor, M: int, q_l2: Optional[Tensor] = None, t_l2: Optional[Tensor] = None,
                   clamp_q: float = 0.1, clamp_t: float = 0.5) -> Tensor:
    """Local sums that make the global means: [sum_b loss_sum[l,b] for l] + [sum clamp(q), sum clamp(t), n_pairs, M]."""
    L, B = loss_sum.shape
    parts = [loss_sum.double().sum(dim=1)]
    zero = loss_sum.new_zeros(1, dtype=torch.float64)
    parts.append(torch.clamp(q_l2, 0.0, clamp_q).double().sum().reshape(1) if q_l2 is not None else zero)
    parts.append(torch.clamp(t_l2, 0.0, clamp_t).double().sum().reshape(1) if t_l2 is not None else zero)
    # torch.full (a fill kernel) rather than torch.tensor (a host copy) so that this is hipGraph-capturable
    parts.append(torch.full((1,), float(B), dtype=torch.float64, device=loss_sum.device))
    parts.append(torch.full((1,), float(M), dtype=torch.float64, device=loss_sum.device))
    return torch.cat(parts)


def reduce_losses(packed: Tensor, L: int, balance_q: float = 1.0, balance_t: float = 0.1, group=None) -> Dict[str, Tensor]:
    """All-reduce(SUM) the packed vector (L+4 doubles: latency-bound, any algorithm) and derive the global-batch
    quantities: per-layer F-loss means, loss_F, and the clamped qt loss."""
    world_M = packed[L + 3].clone()  # M is identical on every rank; keep it out of the sum
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
    n = packed[L + 2]
    loss_layers = packed[:L] / (n * world_M)
    loss_q = packed[L] / (n * L)
    loss_t = packed[L + 1] / (n * L)
    return {"loss_layers": loss_layers, "loss_F": loss_layers.mean(), "loss_q": loss_q, "loss_t": loss_t,
            "loss_qt": loss_q * balance_q + loss_t * balance_t, "n_pairs": n}


class OverlappedLossExchange:
    """Per-step all-reduce of the packed loss vector that never stalls the solver stream.

    The reduced losses are bookkeeping (logging, metrics); nothing of the next step depends on them.  So the vector is
    copied into one of ``depth`` staging buffers and all-reduced with ``async_op=True``: with RCCL the collective runs on
    the communicator's own stream behind an event recorded after the copy, and the next step's kernels (a hipGraph replay
    in bench.py) start immediately instead of waiting ~30 us of small-message latency over xGMI.  A staging buffer is
    reused only after its previous collective has been waited on (a stream-level dependency with RCCL, not a host block).
    ``drain()`` waits for everything outstanding and returns the most recent reduced vector."""

    def __init__(self, numel: int, device, depth: int = 2, dtype=torch.float64, group=None):
        self.bufs = [torch.zeros(numel, device=device, dtype=dtype) for _ in range(max(1, depth))]
        self.works = [None] * len(self.bufs)
        self.group = group
        self.step = 0
        self.last = None

    def exchange(self, packed: Tensor) -> None:
        i = self.step % len(self.bufs)
        if self.works[i] is not None:
            self.works[i].wait()
            self.works[i] = None
        self.bufs[i].copy_(packed)
        if dist.is_available() and dist.is_initialized():
            self.works[i] = dist.all_reduce(self.bufs[i], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.last = i
        self.step += 1

    def drain(self) -> Optional[Tensor]:
        for i, w in enumerate(self.works):
            if w is not None:
                w.wait()
                self.works[i] = None
        return None if self.last is None else self.bufs[self.last]
