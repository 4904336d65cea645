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


def alloc_result_buffers(batch: int, max_det: int, device) -> Tuple[torch.Tensor, torch.Tensor]:
    """Device buffers lp_run_batch_device writes: records [batch, max_det, 32] uint8 and
    counts [2*batch] int32 (kept, pre-filter)."""
    dets = torch.zeros((batch, max_det, RECORD_BYTES), dtype=torch.uint8, device=device)
    counts = torch.zeros((2 * batch,), dtype=torch.int32, device=device)
    return dets, counts


def gather_detections(dets: torch.Tensor, counts: torch.Tensor, dst: int = 0,
                      group: Optional[dist.ProcessGroup] = None) -> Optional[Tuple[torch.Tensor, torch.Tensor]]:
    """The one exchange step: every rank contributes its padded records + counts; rank ``dst``
    returns ([world*batch, max_det, 32] uint8, [world, 2*batch] int32) in rank order, others None.
    Records and counts travel in ONE message (counts are appended as bytes)."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return dets, counts.view(1, -1)
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    payload = torch.cat([dets.reshape(-1), counts.view(torch.uint8).reshape(-1)])
    out: Optional[List[torch.Tensor]] = None
    if rank == dst:
        out = [torch.empty_like(payload) for _ in range(world)]
    dist.gather(payload, gather_list=out, dst=dst, group=group)
    if rank != dst:
        return None
    nd = dets.numel()
    all_dets = torch.stack([o[:nd].view(dets.shape) for o in out]).reshape(world * dets.shape[0], *dets.shape[1:])
    all_counts = torch.stack([o[nd:].view(torch.int32) for o in out])
    return all_dets, all_counts


def records_to_numpy(dets: torch.Tensor) -> np.ndarray:
    """[..., max_det, 32] uint8 tensor -> structured lp_det array."""
    a = dets.detach().cpu().numpy()
    return np.ascontiguousarray(a).view(DET_DTYPE).reshape(a.shape[:-1])
