"""GPU tests of the entry point bench.py times: ``lp_run_batch_device`` (device-resident images in, device-resident
records out, hipGraph capture / replay keyed on the raw pointers) against the host path ``lp_run_batch`` that the oracle
tests pin, and one pass of the records through the RCCL gather at world size 1.

The device path differs from the host path in everything around the kernels -- 3*B count words, graph capture on the
second sight of a key and replay afterwards, thresholds and pointers baked into the captured arguments, the ROI list
built inside the NMS kernel -- so the same batch must give the same records, counts, pre-filter counts and mean-score
bits on the first (eager), second (capturing) and later (replayed) call, with two alternating input buffers as
bench.py uses them, and with LITEPI_NO_GRAPH=1."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _models(tmp_path, n_cal=8):
    from litepi import Engine, ncnn_export
    from litepi.backend import random_shufflenet_state
    from oracle import ncnn_ref
    p, b = str(tmp_path / "m.param"), str(tmp_path / "m.bin")
    ncnn_export.export_detector(p, b, "v1", seed=4321, cls_bias=0.0)
    rng = np.random.default_rng(99)
    imgs = rng.integers(0, 256, (2, 16, 640, 640, 3), dtype=np.uint8)
    # class bias so that ~8 anchors per image pass conf 0.25 (as bench.py calibrates), using the device's own scores
    e = Engine(precision="fp16", max_batch=16, max_det=300, num_classes=91)
    try:
        e.load_detector(p, b)
        s = np.sort(e.detect_raw(imgs[0][:n_cal])[:, 4].astype(np.float64).ravel())[::-1]
    finally:
        e.close()
    k = 8 * n_cal
    mid = 0.5 * (np.log(s[k - 1] / (1 - s[k - 1])) + np.log(s[k] / (1 - s[k])))
    ncnn_export.shift_cls_bias(p, b, float(np.log(0.25 / 0.75) - mid))
    return p, b, random_shufflenet_state(91, seed=3), imgs


def _host_reference(eng, batch):
    dets, counts, num_det, _ = eng.run_batch(list(batch), 0.25, 0.45, 50)
    return dets, counts, num_det, eng.last_det_conf_avg.copy()


def _check_device_call(eng, dev_imgs, res, ref, tag):
    from litepi._ffi import DET_DTYPE
    dets_ref, counts_ref, num_det_ref, avg_ref = ref
    B = dev_imgs.shape[0]
    eng.run_batch_device(dev_imgs.data_ptr(), B, 640, 640, 0.25, 0.45, 50, res.dets.data_ptr(), res.counts.data_ptr())
    eng.synchronize()
    torch.cuda.synchronize()
    counts = res.counts.cpu().numpy()
    kept, pre = counts[:B].astype(np.int64), counts[B:2 * B].astype(np.int64)
    avg_bits = counts[2 * B:3 * B].view(np.float32)
    assert np.array_equal(kept, counts_ref), f"{tag}: kept counts differ"
    assert np.array_equal(pre, num_det_ref), f"{tag}: pre-filter counts differ"
    assert np.array_equal(avg_bits.view(np.uint32), avg_ref.view(np.uint32)), f"{tag}: mean detector score bits differ"
    recs = res.dets.cpu().numpy().reshape(B, -1).view(DET_DTYPE).reshape(B, -1)
    for i in range(B):
        got, want = recs[i, :kept[i]], dets_ref[i, :kept[i]]
        assert got.tobytes() == want.tobytes(), f"{tag}: records of image {i} differ"
    return int(kept.sum())


@pytest.mark.parametrize("no_graph", [False, True], ids=["graph", "no_graph"])
def test_run_batch_device_matches_host_path(tmp_path, monkeypatch, no_graph):
    from litepi import Engine
    from litepi.distributed import alloc_result_buffers
    if no_graph:
        monkeypatch.setenv("LITEPI_NO_GRAPH", "1")
    else:
        monkeypatch.delenv("LITEPI_NO_GRAPH", raising=False)
    p, b, cls_state, imgs = _models(tmp_path)
    dev = torch.device("cuda", 0)
    B = imgs.shape[1]
    ref_eng = Engine(precision="fp16", max_batch=B, max_det=300, num_classes=91)
    eng = Engine(precision="fp16", max_batch=B, max_det=300, num_classes=91)
    try:
        for e in (ref_eng, eng):
            e.load_detector(p, b)
            e.load_classifier(cls_state)
        refs = [_host_reference(ref_eng, imgs[j]) for j in range(2)]
        assert sum(int(r[1].sum()) for r in refs) >= 16, "the calibrated batches must produce detections"
        dev_imgs = [torch.from_numpy(imgs[j]).to(dev) for j in range(2)]
        res = alloc_result_buffers(B, 300, dev)
        st = torch.cuda.Stream(device=dev)
        eng.set_stream(st.cuda_stream)
        total = 0
        # calls 0/1: first sight of each input pointer (eager); 2/3: capture; 4..7: replay -- alternating buffers as bench.py
        for call in range(8):
            j = call % 2
            with torch.cuda.stream(st):
                total += _check_device_call(eng, dev_imgs[j], res, refs[j], f"call {call} (buffer {j}, {'no graph' if no_graph else 'graph'})")
        print(f"device path == host path on 8 calls, {total} records compared")
    finally:
        eng.close()
        ref_eng.close()


def test_rccl_self_gather_world_size_1(tmp_path):
    """RCCL executes once on hardware: init_process_group('nccl', world_size=1), the per-step payload of
    lp_run_batch_device through Gatherer.gather on the handle's stream order, receive-slot views intact."""
    import torch.distributed as dist
    from litepi import Engine
    from litepi.distributed import Gatherer, alloc_result_buffers
    p, b, cls_state, imgs = _models(tmp_path)
    dev = torch.device("cuda", 0)
    B = imgs.shape[1]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29611")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        created = True
    eng = Engine(precision="fp16", max_batch=B, max_det=300, num_classes=91)
    try:
        eng.load_detector(p, b)
        eng.load_classifier(cls_state)
        st = torch.cuda.Stream(device=dev)
        eng.set_stream(st.cuda_stream)
        res = alloc_result_buffers(B, 300, dev)
        g = Gatherer(res, dst=0, force=True)
        assert g.active and g.world == 1
        dimg = torch.from_numpy(imgs[0]).to(dev)
        for _ in range(3):   # eager, capture, replay -- the gather follows each on the same stream
            with torch.cuda.stream(st):
                eng.run_batch_device(dimg.data_ptr(), B, 640, 640, 0.25, 0.45, 50, res.dets.data_ptr(), res.counts.data_ptr())
                gd, gc = g.gather(res)
        torch.cuda.synchronize()
        assert gd.shape[0] == 1 and gd.shape[1] == B and gc.shape[0] == 1
        assert torch.equal(gc.reshape(-1).cpu(), res.counts.cpu()), "gathered counts differ from the rank's counts"
        assert torch.equal(gd.reshape(-1).cpu(), res.dets.reshape(-1).cpu()), "gathered records differ from the rank's records"
        assert int(res.counts[:B].sum().item()) >= 8
    finally:
        eng.close()
        if created:
            dist.destroy_process_group()


def test_lp_gather_native_collective_world_size_1(tmp_path):
    """The library's OWN collective (include/litepi.h ABI 310: lp_comm_unique_id / lp_comm_init / lp_gather / lp_comm_destroy;
    RCCL bound lazily with dlopen, ncclGather on the handle's stream) executes on hardware at the only world size a one-GPU box
    allows: the per-step payload of lp_run_batch_device gathered on the handle's own stream with no event and no
    torch.distributed -- eager, capture and replay -- equals the rank's payload byte for byte; misuse is an error."""
    from litepi import Engine, _ffi
    from litepi.distributed import NativeGatherer, alloc_result_buffers
    p, b, cls_state, imgs = _models(tmp_path)
    dev = torch.device("cuda", 0)
    B = imgs.shape[1]
    eng = Engine(precision="fp16", max_batch=B, max_det=300, num_classes=91)
    try:
        eng.load_detector(p, b)
        eng.load_classifier(cls_state)
        res = alloc_result_buffers(B, 300, dev)
        with pytest.raises(_ffi.LitepiError):     # no communicator yet
            _ffi.check(eng.lib, eng.lib.lp_gather(eng._h, res.payload.data_ptr(), res.payload.numel(), res.payload.data_ptr(), 0))
        g = NativeGatherer(eng, res, rank=0, world=1, dst=0)
        with pytest.raises(_ffi.LitepiError):     # a handle has one communicator
            NativeGatherer(eng, res, rank=0, world=1, dst=0)
        dimg = torch.from_numpy(imgs[0]).to(dev)
        torch.cuda.synchronize()
        for _ in range(3):   # eager, capture, replay -- the gather follows each on the handle's stream
            eng.run_batch_device(dimg.data_ptr(), B, 640, 640, 0.25, 0.45, 50, res.dets.data_ptr(), res.counts.data_ptr())
            gd, gc = g.gather(res)
        eng.synchronize()
        assert gd.shape[0] == 1 and gd.shape[1] == B and gc.shape[0] == 1
        assert torch.equal(gc.reshape(-1).cpu(), res.counts.cpu()), "gathered counts differ from the rank's counts"
        assert torch.equal(gd.reshape(-1).cpu(), res.dets.reshape(-1).cpu()), "gathered records differ from the rank's records"
        assert int(res.counts[:B].sum().item()) >= 8
        g.close()
        g.close()   # idempotent
    finally:
        eng.close()


def test_dropin_upload_lanes_equal_single_handles(tmp_path, monkeypatch):
    """HybridPipeline.run_batch on 32 host images with the upload lanes (round 4: four further handles of max_batch / 4
    images, a call's images dealt to them in contiguous slices, each slice one lp_run_batch from a worker thread) against
    plain single handles of the SAME capacity (the plan depends on the capacity) fed the same slices one after the other:
    every result dict must be identical -- the lanes change when an image is processed, never how."""
    from litepi import HybridPipeline
    p, b, sd, imgs = _models(tmp_path)
    cls_path = str(tmp_path / "cls.pth")
    torch.save({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, cls_path)
    batch = [imgs[j // 16][j % 16] for j in range(32)]
    monkeypatch.setenv("LITEPI_DROPIN_LANES", "4")   # (off by default: an experiment that measured slower, backend.py)
    pipe = HybridPipeline(p, b, cls_path, "shufflenetv2", num_classes=91, precision="fp16", max_batch=32, max_det=300)
    try:
        assert len(pipe._lanes) == 4 and pipe._lane_cap == 8
        first = pipe.run_batch(batch, 0.25, 0.45, 50)
        again = pipe.run_batch(batch, 0.25, 0.45, 50)   # captured / replayed graphs on every lane
    finally:
        pipe.close()
    monkeypatch.setenv("LITEPI_DROPIN_LANES", "1")
    ref = HybridPipeline(p, b, cls_path, "shufflenetv2", num_classes=91, precision="fp16", max_batch=8, max_det=300)
    try:
        assert not ref._lanes
        want = []
        for k in range(4):
            want += ref.run_batch(batch[8 * k:8 * k + 8], 0.25, 0.45, 50)
    finally:
        ref.close()
    n = 0
    for tag, got in (("first", first), ("again", again)):
        assert len(got) == 32
        for i in range(32):
            (rg, mg), (rw, mw) = got[i], want[i]
            assert mg.num_detections == mw.num_detections and mg.det_confidence_avg == mw.det_confidence_avg, (tag, i)
            assert len(rg) == len(rw), (tag, i)
            for a, c in zip(rg, rw):
                assert all(a[k] == c[k] for k in ("bbox", "det_class", "det_conf", "cls_class", "cls_conf")), (tag, i, a, c)
                n += 1
    assert n >= 2 * 32, f"only {n // 2} results in 32 images"
    print(f"upload lanes: {n // 2} results in 32 images identical to the single-handle path (first call and replay)")


def test_run_batch_chunked_equals_whole_batch(tmp_path, monkeypatch):
    """LITEPI_RUN_CHUNK=16 (an A/B switch, off by default: no gain measured): lp_run_batch on >= 32 frames of one size walks the
    batch in chunks of 16 (api.cpp run_batch_chunked: chunk k's upload on a copy stream under chunk k-1's kernels, every chunk a
    complete detect -> NMS -> ROI -> classifier pass over its slice of the handle's buffers, captured per chunk).  The same
    handle without the switch runs the batch whole: records, counts,
    pre-filter counts and the mean-score bits must be identical, on the first (eager), second (capturing) and third (replaying)
    call; a 40-frame batch has a ragged last chunk."""
    from litepi import Engine
    p, b, sd, imgs = _models(tmp_path)
    frames = [imgs[j // 16][j % 16] for j in range(32)] + [imgs[0][j] for j in range(8)]
    eng = Engine(precision="fp16", max_batch=40, max_det=300, num_classes=91)
    try:
        eng.load_detector(p, b)
        eng.load_classifier(sd)
        monkeypatch.delenv("LITEPI_RUN_CHUNK", raising=False)
        want = {n: _host_reference(eng, frames[:n]) for n in (32, 40)}
        monkeypatch.setenv("LITEPI_RUN_CHUNK", "16")
        total = 0
        for call in range(3):
            for n in (32, 40):
                dets, counts, num_det, avg = _host_reference(eng, frames[:n])
                wd, wc, wn, wa = want[n]
                assert np.array_equal(counts, wc) and np.array_equal(num_det, wn), (call, n)
                assert np.array_equal(avg.view(np.uint32), wa.view(np.uint32)), (call, n)
                for i in range(n):
                    assert dets[i, :counts[i]].tobytes() == wd[i, :wc[i]].tobytes(), (call, n, i)
                total += int(counts.sum())
        assert total >= 3 * 72
        print(f"chunked lp_run_batch: {total // 3} records in 32 + 40 frames identical to the whole-batch pass (eager, capture, replay)")
    finally:
        eng.close()
