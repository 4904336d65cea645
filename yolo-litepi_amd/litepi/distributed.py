"""Multi-GPU partitioning of the pipeline: one process per GPU, images sharded across ranks,
every rank holds a full weight replica, and ONE collective per batch moves the final
fixed-capacity detection records to rank 0.  The reference has no counterpart
(single process, CPU only); there is no data-path collective besides this gather.

``torch.distributed`` is the transport: backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the
CPU tests.  The payload is tiny (max_det x 32 B per image), so the gather is latency-bound.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

from ._ffi import DET_DTYPE

RECORD_BYTES = 32  # sizeof(lp_det)


def env_rank_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment (defaults: single process)."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block partition of n_items over world ranks (first n%world ranks get one extra)."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


COUNT_WORDS = 3  # per image: kept after the area filter, kept before it, float bits of the mean detector score


class ResultBuffers:
    """What ``lp_run_batch_device`` writes for one batch, laid out as ONE contiguous message:
    ``[batch * max_det * 32 B of lp_det records | 3 * batch int32 counts]``.  ``dets`` and ``counts``
    are views into ``payload``, so the multi-GPU gather sends the buffer as it stands -- no
    per-step concatenation or allocation."""

    def __init__(self, batch: int, max_det: int, device):
        self.batch, self.max_det = batch, max_det
        nd = batch * max_det * RECORD_BYTES
        self.payload = torch.zeros((nd + COUNT_WORDS * batch * 4,), dtype=torch.uint8, device=device)
        self.dets = self.payload[:nd].view(batch, max_det, RECORD_BYTES)
        self.counts = self.payload[nd:].view(torch.int32)

    def __iter__(self):  # (dets, counts) unpacking, as the round-1 helper returned
        return iter((self.dets, self.counts))


def alloc_result_buffers(batch: int, max_det: int, device) -> ResultBuffers:
    """Device buffers lp_run_batch_device writes: records [batch, max_det, 32] uint8 and
    counts [3*batch] int32 (kept, pre-filter, mean-score bits), views of one payload tensor."""
    return ResultBuffers(batch, max_det, device)


class Gatherer:
    """The one exchange step of the multi-GPU path, with everything allocated once: rank ``dst``
    owns ``world`` receive slots of payload size; ``gather(buf)`` is one ``dist.gather`` (RCCL
    ``ncclGather``-style send/recv over xGMI on GPUs, gloo in the CPU tests) and returns views."""

    def __init__(self, like: ResultBuffers, dst: int = 0, group: Optional[dist.ProcessGroup] = None, force: bool = False):
        """force: run the collective even in a world of one (a self-gather: how the RCCL path is exercised on a
        single-GPU box, ``bench.py --rccl-self`` and tests/test_gpu_device_path.py)."""
        self.dst, self.group = dst, group
        self.active = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force)
        self.world = dist.get_world_size(group) if self.active else 1
        self.rank = dist.get_rank(group) if self.active else 0
        self.batch, self.max_det = like.batch, like.max_det
        self.recv: Optional[torch.Tensor] = None
        self.slots: Optional[List[torch.Tensor]] = None
        if self.active and self.rank == dst:
            self.recv = torch.empty((self.world, like.payload.numel()), dtype=torch.uint8, device=like.payload.device)
            self.slots = [self.recv[r] for r in range(self.world)]

    def gather(self, buf: ResultBuffers) -> Optional[Tuple[torch.Tensor, torch.Tensor]]:
        """rank dst: ([world, batch, max_det, 32] uint8, [world, 3*batch] int32) in rank order (views of
        the receive buffer, valid until the next gather: nothing is copied or allocated); other ranks: None."""
        if not self.active:
            return buf.dets.unsqueeze(0), buf.counts.view(1, -1)
        dist.gather(buf.payload, gather_list=self.slots, dst=self.dst, group=self.group)
        if self.rank != self.dst:
            return None
        nd = self.batch * self.max_det * RECORD_BYTES
        all_dets = self.recv[:, :nd].view(self.world, self.batch, self.max_det, RECORD_BYTES)
        all_counts = self.recv[:, nd:].view(torch.int32)
        return all_dets, all_counts


def gather_detections(dets, counts=None, dst: int = 0, group: Optional[dist.ProcessGroup] = None):
    """Convenience form (allocates its receive slots per call; steady-state loops keep a ``Gatherer``)."""
    buf = dets if isinstance(dets, ResultBuffers) else None
    if buf is None:
        if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
            return dets.unsqueeze(0), counts.view(1, -1)
        buf = ResultBuffers(dets.shape[0], dets.shape[1], dets.device)
        buf.dets.copy_(dets)
        buf.counts.copy_(counts)
    return Gatherer(buf, dst, group).gather(buf)


def records_to_numpy(dets: torch.Tensor) -> np.ndarray:
    """[..., max_det, 32] uint8 tensor -> structured lp_det array."""
    a = dets.detach().cpu().numpy()
    return np.ascontiguousarray(a).view(DET_DTYPE).reshape(a.shape[:-1])
