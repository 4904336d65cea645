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


class NativeGatherer:
    """The same exchange step through the LIBRARY's own collective (include/litepi.h, ABI 310: ``lp_comm_init`` + ``lp_gather`` =
    ``ncclGather`` from RCCL, bound lazily with dlopen): no torch.distributed on the data path, and the gather is enqueued on the
    handle's own stream, i.e. ordered behind the ``lp_run_batch_device`` that wrote the payload without any event.

    The 128-byte ``ncclUniqueId`` has to reach every rank out of band.  ``id_bytes`` may be passed explicitly (drawn on rank 0
    with ``NativeGatherer.unique_id()`` and shipped by the launcher); if it is None and a torch.distributed process group of any
    backend (gloo is enough) exists, rank 0's id is broadcast over it -- that group then carries 128 bytes once, nothing else."""

    @staticmethod
    def unique_id() -> bytes:
        import ctypes as C
        from . import _ffi
        lib = _ffi.load_library()
        buf = (C.c_char * 128)()
        _ffi.check(lib, lib.lp_comm_unique_id(buf))
        return bytes(buf.raw)

    def __init__(self, engine, like: ResultBuffers, rank: int = 0, world: int = 1, dst: int = 0, id_bytes: Optional[bytes] = None):
        import ctypes as C
        from . import _ffi
        self.engine, self.rank, self.world, self.dst = engine, rank, world, dst
        self.batch, self.max_det = like.batch, like.max_det
        if id_bytes is None:
            if world > 1:
                if not (dist.is_available() and dist.is_initialized()):
                    raise RuntimeError("NativeGatherer: pass id_bytes (NativeGatherer.unique_id() on rank 0) or initialise a process group to broadcast it")
                box = [self.unique_id() if rank == 0 else None]
                dist.broadcast_object_list(box, src=0)
                id_bytes = box[0]
            else:
                id_bytes = self.unique_id()
        idbuf = (C.c_char * 128).from_buffer_copy(id_bytes)
        _ffi.check(engine.lib, engine.lib.lp_comm_init(engine._h, idbuf, rank, world))
        self.recv = torch.empty((world, like.payload.numel()), dtype=torch.uint8, device=like.payload.device) if rank == dst else None

    def gather(self, buf: ResultBuffers):
        """rank dst: ([world, batch, max_det, 32] uint8, [world, 3*batch] int32) views of the receive buffer (valid once the
        handle's stream has reached this point: synchronise it, or keep consuming on it); other ranks: None."""
        from . import _ffi
        recv_ptr = self.recv.data_ptr() if self.recv is not None else None
        _ffi.check(self.engine.lib, self.engine.lib.lp_gather(self.engine._h, buf.payload.data_ptr(), buf.payload.numel(), recv_ptr, self.dst))
        if self.rank != self.dst:
            return None
        nd = self.batch * self.max_det * RECORD_BYTES
        return self.recv[:, :nd].view(self.world, self.batch, self.max_det, RECORD_BYTES), self.recv[:, nd:].view(torch.int32)

    def close(self) -> None:
        from . import _ffi
        _ffi.check(self.engine.lib, self.engine.lib.lp_comm_destroy(self.engine._h))


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


# ---- the evaluation harness across ranks (litepi/e2e.py --gpus N) -------------------------------------------------------
# Each rank evaluates a contiguous shard of the sorted image list (shard_range) and keeps, per image, the predictions and
# the ground truths the metric consumes; ONE variable-length gather at the end moves them to rank 0, which scores the whole
# set (evaluate_predictions needs every prediction: it ranks them globally by confidence).  Rows are packed as float64
# [image index in the shard, kind (0 = prediction, 1 = ground truth), class, x1, y1, x2, y2, conf] -- exact for the int
# pixel boxes and for the fp32 confidences.
PRED_COLS = 8


def pack_eval_rows(all_preds: Sequence[Sequence[dict]], all_gts: Sequence[Sequence[tuple]]) -> np.ndarray:
    rows = []
    for i, (preds, gts) in enumerate(zip(all_preds, all_gts)):
        for p in preds:
            x1, y1, x2, y2 = p["bbox"]
            rows.append((i, 0, p["cls_class"], x1, y1, x2, y2, p["conf"]))
        for g in gts:
            rows.append((i, 1, g[0], g[1], g[2], g[3], g[4], 0.0))
    return np.asarray(rows, dtype=np.float64).reshape(-1, PRED_COLS)


def unpack_eval_rows(rows: np.ndarray, n_images: int):
    preds: List[List[dict]] = [[] for _ in range(n_images)]
    gts: List[List[tuple]] = [[] for _ in range(n_images)]
    for r in rows:   # row order within an image is the packing order, i.e. the pipeline's result order
        i = int(r[0])
        if int(r[1]) == 0:
            preds[i].append({"bbox": (int(r[3]), int(r[4]), int(r[5]), int(r[6])), "conf": float(np.float32(r[7])), "cls_class": int(r[2])})
        else:
            gts[i].append((int(r[2]), int(r[3]), int(r[4]), int(r[5]), int(r[6])))
    return preds, gts


def gather_eval_shards(all_preds, all_gts, bench_time: float, device="cpu", dst: int = 0, group: Optional[dist.ProcessGroup] = None):
    """Shards (in rank order = image order, shard_range is contiguous) -> rank dst: (all_preds, all_gts, [bench_time per rank]);
    other ranks: None.  Two collectives for the whole evaluation: an all-gather of (rows, images, time) per rank and one padded
    gather of the rows (backend nccl = RCCL with `device` the rank's GPU, gloo with "cpu")."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return list(all_preds), list(all_gts), [bench_time]
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    rows = pack_eval_rows(all_preds, all_gts)
    meta = torch.tensor([float(len(rows)), float(len(all_preds)), float(bench_time)], dtype=torch.float64, device=device)
    metas = [torch.empty_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    metas = [m.cpu().numpy() for m in metas]
    cap = max(1, int(max(m[0] for m in metas)))
    send = torch.zeros((cap, PRED_COLS), dtype=torch.float64, device=device)
    if len(rows):
        send[:len(rows)] = torch.from_numpy(rows).to(device)
    slots = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    dist.gather(send, gather_list=slots, dst=dst, group=group)
    if rank != dst:
        return None
    preds_all, gts_all = [], []
    for m, slot in zip(metas, slots):
        p, g = unpack_eval_rows(slot[:int(m[0])].cpu().numpy(), int(m[1]))
        preds_all.extend(p)
        gts_all.extend(g)
    return preds_all, gts_all, [float(m[2]) for m in metas]
