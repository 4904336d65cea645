#!/usr/bin/env python3
"""Headline benchmark: images/sec end-to-end (detector + NMS + ROI + ShuffleNetV2), 640x640,
batch 64 per GPU, fp16 storage / fp32 accumulate (BASELINE.json configs[2]; configs[3] at N>1).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole hot path over one batch of B synthetic uint8 BGR images that
are already resident in HBM (lp_run_batch_device), plus -- at N>1 -- the single RCCL gather of the
detection records to rank 0.  Weak scaling: every rank processes its own B images per step.
Steps rotate over --nbatches distinct input batches and --inflight pipeline handles (own stream,
own activation buffers), so consecutive steps overlap on the GPU and no step re-reads the inputs
of the previous one.

Weights: seeded random-init models of the reference's architectures written by
litepi.ncnn_export (YOLO-LitePi v1 widths by default) and a seeded ShuffleNetV2 x1.0; the
class-branch bias is calibrated once, untimed, so that ~8 anchors per image pass conf 0.25 and the
classifier stage has real work.  No dataset/checkpoint is available.

The JSON line (rank 0):
  value / ms_per_step -- the K timed steps between barrier + synchronize pairs (the contract)
  windows             -- --windows further windows of K steps each: median and p95 of ms/step
  h2d_inclusive       -- the same loop with the batch starting in PINNED HOST memory: async H2D copy
                         on the handle's stream, then the pipeline (the reference's t_total starts
                         from a host image, e2e.py:446-506); never `value`
  roofline            -- dominant conv kernel family: algorithmic FLOPs of its launches / their
                         HIP-event durations (profiled passes of the same step, same process) against
                         the dense fp16 MFMA peak (2.5 PFLOP/s, MI355X_MICROARCH.md); the HBM view of the
                         same family and of the whole step as secondary fields
  cpu_baseline        -- the CPU restatement of the reference path (oracle/, torch-CPU + NumPy) timed
                         on this host's cores (all of the GPU's CPU share, and 4 threads = the paper's
                         Pi 5 setting) over a bounded sample of the same images (rank 0, N=1)
"""
import argparse
import json
import os
import sys
import tempfile
import time

_ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (_ROOT, os.path.join(_ROOT, "yolo-litepi_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_FP16_TFLOPS = 2500.0   # dense MFMA fp16/bf16, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_HBM_GBS = 8000.0
CONF, IOU, MIN_AREA = 0.25, 0.45, 50
NUM_CLASSES = 91            # TT100K classifier head
TARGET_CANDIDATES = 8       # anchors per image above conf after calibration
PROFILE_ROUND = "r04"


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=2, choices=[2, 4],
                    help="BASELINE.json configuration: 2 = 640x640 batch 64 (the metric's configuration; configs[3] at N > 1), "
                         "4 = TT100K-shape 2048x2048 frames, letterboxed to 640 ON THE DEVICE inside the timed step, batch 32 per GPU "
                         "(256 over 8 GPUs)")
    ap.add_argument("--batch", type=int, default=None, help="images per GPU per step (default 64; 32 for --config 4)")
    ap.add_argument("--preset", default="v1", choices=["v1", "v2"])
    ap.add_argument("--precision", default="fp16", choices=["fp16", "fp32"])
    ap.add_argument("--max-det", type=int, default=300)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-images", type=int, default=256, help="images in the CPU baseline sample (all-core run)")
    ap.add_argument("--profile-steps", type=int, default=3)
    ap.add_argument("--inflight", type=int, default=3,
                    help="independent pipeline handles (own stream + activation buffers) the steps rotate over, so "
                         "consecutive steps overlap on the GPU")
    ap.add_argument("--nbatches", type=int, default=4, help="distinct input batches the steps rotate over")
    ap.add_argument("--windows", type=int, default=9, help="extra timed windows of --steps steps (median / p95)")
    ap.add_argument("--no-h2d", action="store_true", help="skip the host-resident (PCIe-inclusive) measurement")
    ap.add_argument("--no-dropin", action="store_true", help="skip the HybridPipeline.run_batch (host images -> dicts) measurement")
    ap.add_argument("--conf", type=float, default=None, help="experiment: detector confidence threshold (default 0.25, BASELINE configs[2])")
    ap.add_argument("--dump-profile", default="", help="write the per-launch profile of the roofline pass to this JSON file")
    ap.add_argument("--rccl-self", action="store_true",
                    help="N=1 only: initialise the nccl (RCCL) process group at world size 1 and push every step's payload through "
                         "the gather (a self-gather), so that the multi-GPU step -- RCCL init, stream ordering against the handle's "
                         "stream, the receive-slot views -- executes on a single-GPU box")
    ap.add_argument("--native-gather", action="store_true",
                    help="move the records with the library's own collective (lp_comm_init + lp_gather = ncclGather bound with dlopen, on "
                         "the handle's stream) instead of torch.distributed.gather; with --rccl-self also at N=1")
    return ap.parse_args()


def build_models(args, workdir, engine_factory, cal_imgs):
    """Export the synthetic detector, calibrate its class bias with one untimed GPU pass, and
    return (param, bin, classifier_state, spec)."""
    from litepi import ncnn_export
    from litepi.backend import random_shufflenet_state

    param, binf = os.path.join(workdir, "det.param"), os.path.join(workdir, "det.bin")
    # configs[4]: boxes of ~10-40 px in the letterboxed image = ~30-120 px in the 2048 x 2048 frame (TT100K signs: ~54 px);
    # the default ramp (boxes of ~50 px at 640) would make them 160-640 px crops there
    ramp = {"box_ramp": float(os.environ.get("LITEPI_BENCH_RAMP", "4.0"))} if getattr(args, "config", 2) == 4 else {}
    spec = ncnn_export.export_detector(param, binf, args.preset, seed=1234, cls_bias=0.0, **ramp)
    eng = engine_factory()
    eng.load_detector(param, binf)
    nb = min(8, cal_imgs.shape[0])
    if cal_imgs.shape[1] != 640:   # configs[4]: the calibration sees what the detector sees, the frames letterboxed on the device
        cal_imgs = np.stack([eng.test_letterbox(im)[0] for im in cal_imgs[:nb]])
    out0 = eng.detect_raw(cal_imgs[:nb])
    eng.close()
    s = np.sort(out0[:, 4].astype(np.float64).ravel())[::-1]
    kth = min(max(s[TARGET_CANDIDATES * nb], 1e-6), 1 - 1e-6)
    delta = float(np.log(CONF / (1 - CONF)) - np.log(kth / (1 - kth)))
    ncnn_export.shift_cls_bias(param, binf, delta)
    if cal_imgs.shape[1] != 640 or getattr(args, "config", 2) == 4:
        # configs[4]: the frames are piecewise constant (32 x 32 blocks), so thousands of anchors share a score and the k-th
        # score is a cliff -- the shift above saturates max_det on every image.  Bisect the shift on what the workload is
        # defined by instead: boxes KEPT after NMS, ~TARGET_CANDIDATES per image (TT100K: 2.8 signs per frame), untimed.
        target = TARGET_CANDIDATES * nb
        lo, hi, applied = -80.0, 0.0, 0.0    # extra shift relative to `delta`
        kept = None
        for _ in range(18):
            mid = 0.5 * (lo + hi)
            ncnn_export.shift_cls_bias(param, binf, mid - applied)
            applied = mid
            eng = engine_factory()
            eng.load_detector(param, binf)
            _, counts = eng.detect(list(cal_imgs[:nb]), CONF, IOU)
            eng.close()
            kept = int(counts.sum())
            if 0.6 * target <= kept <= 1.6 * target:
                break
            if kept > target:
                hi = mid
            else:
                lo = mid
        else:
            # a cliff (smooth frames: thousands of anchors within 0.003 logit): settle on its low side rather than saturate
            if kept > 1.6 * target:
                ncnn_export.shift_cls_bias(param, binf, lo - applied)
                applied = lo
                eng = engine_factory()
                eng.load_detector(param, binf)
                _, counts = eng.detect(list(cal_imgs[:nb]), CONF, IOU)
                eng.close()
                kept = int(counts.sum())
        delta += applied
        spec["calibration"] = f"bisected on kept boxes: {kept} on {nb} frames"
    spec["cls_bias_shift"] = delta
    return param, binf, random_shufflenet_state(NUM_CLASSES, seed=0), spec


def cpu_baseline(param, binf, cls_state, imgs_np, n_images):
    """Reference path restated on CPU (oracle/), timed image by image like e2e.py's loop; once on the GPU's whole CPU
    share and once on 4 threads (the paper's Raspberry Pi 5 setting)."""
    from oracle import ncnn_ref, pipeline_ref, shufflenet_ref

    layers = ncnn_ref.load_model(param, binf)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in cls_state.items()}
    model = shufflenet_ref.ShuffleNetV2(NUM_CLASSES)
    missing = model.load_state_dict(sd, strict=False)
    assert not [k for k in missing.missing_keys if "num_batches_tracked" not in k], missing
    model.eval()
    pipe = pipeline_ref.CpuPipeline(layers, model)

    def run(threads, n):
        torch.set_num_threads(threads)
        pipe.run(imgs_np[0], CONF, IOU, MIN_AREA)  # warm-up
        t0 = time.perf_counter()
        ndet = 0
        for i in range(n):
            res, _ = pipe.run(imgs_np[i % len(imgs_np)], CONF, IOU, MIN_AREA)
            ndet += len(res)
        dt = time.perf_counter() - t0
        return n / dt, ndet, dt

    # the GPU box gives one GPU a 16-core CPU share; oneDNN at batch 1 does not scale past that
    cores = min(16, os.cpu_count() or 1)
    v, ndet, dt = run(cores, n_images)
    n4 = max(16, n_images // 4)
    v4, _, dt4 = run(min(4, cores), n4)
    return {"value": v, "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"{n_images} of the bench images, batch 1, fp32 torch-CPU convs + NumPy post-processing "
                      f"(oracle/pipeline_ref.py), {ndet} classified ROIs, {dt:.1f} s",
            "value_4_threads": v4, "sample_4_threads": f"{n4} images, {dt4:.1f} s, torch.set_num_threads(4)"}


def main():
    args = parse_args()
    run_conf = CONF if args.conf is None else args.conf   # (the calibration always targets CONF)
    SRC = 2048 if args.config == 4 else 640               # side of the input frames (uint8 BGR, resident in HBM)
    if args.batch is None:
        args.batch = 32 if args.config == 4 else 64
    if args.config == 4:
        args.nbatches = min(args.nbatches, 2)             # 2 x 32 x 12.6 MB of frames per rank
        args.cpu_images = min(args.cpu_images, 24)
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or args.rccl_self:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    use_gather = world > 1 or args.rccl_self

    from litepi import Engine
    from litepi.distributed import Gatherer, alloc_result_buffers

    B, NB, K = args.batch, max(1, args.nbatches), args.steps
    if args.config == 4:
        from litepi.synth import config4_images
        imgs_np = config4_images(NB * B, seed=2 + rank, grain=12).reshape(NB, B, SRC, SRC, 3)
    else:
        rng = np.random.default_rng(1 + rank)
        imgs_np = rng.integers(0, 256, (NB, B, 640, 640, 3), dtype=np.uint8)
    imgs = torch.from_numpy(imgs_np).to(dev)

    def engine_factory():
        return Engine(precision=args.precision, max_batch=B, max_det=args.max_det, num_classes=NUM_CLASSES,
                      device=local_rank)

    workdir = tempfile.mkdtemp(prefix=f"litepi_bench_r{rank}_")
    # every rank calibrates on rank 0's images so that all replicas are identical
    if args.config == 4:
        from litepi.synth import config4_images
        cal_imgs = config4_images(8, seed=2, grain=12)
    else:
        cal_imgs = np.random.default_rng(1).integers(0, 256, (8, 640, 640, 3), dtype=np.uint8)
    param, binf, cls_state, spec = build_models(args, workdir, engine_factory, cal_imgs)

    # args.inflight pipeline handles, each with its own (non-default torch) stream and buffers:
    # step i runs on handle i % inflight, so the tail of one step overlaps the head of the next.
    # torch streams: torch.cuda.synchronize(), events and the RCCL gather see the library's work.
    engs, streams, outs = [], [], []
    for _ in range(max(1, args.inflight)):
        e = engine_factory()
        e.load_detector(param, binf)
        e.load_classifier(cls_state)
        st = torch.cuda.Stream(device=dev)
        e.set_stream(st.cuda_stream)
        engs.append(e); streams.append(st); outs.append(alloc_result_buffers(B, args.max_det, dev))
    NH = len(engs)
    eng, (dets, counts) = engs[0], outs[0]
    if args.native_gather and use_gather:
        from litepi.distributed import NativeGatherer
        gatherers = [NativeGatherer(e, o, rank=rank, world=world, dst=0) for e, o in zip(engs, outs)]   # one communicator per handle
    else:
        gatherers = [Gatherer(o, dst=0, force=args.rccl_self) for o in outs]   # receive slots allocated once (rank 0), nothing per step
    torch.cuda.synchronize()
    step_no = [0]

    def step():
        k = step_no[0]
        step_no[0] += 1
        i, j = k % NH, k % NB
        d, c = outs[i]
        with torch.cuda.stream(streams[i]):
            engs[i].run_batch_device(imgs[j].data_ptr(), B, SRC, SRC, run_conf, IOU, MIN_AREA, d.data_ptr(), c.data_ptr())
            if use_gather:
                return gatherers[i].gather(outs[i])
        return d, c.view(1, -1)

    def timed(fn, n):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        enq = time.perf_counter() - t0   # host time to enqueue (no GPU wait unless a queue fills)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, enq

    # set-up, untimed: every (handle, input batch) pair twice -- the library runs a launch sequence eagerly the first time
    # it sees it and captures it into a hipGraph the second time; from then on a step is one graph launch
    lcm = NH * NB // int(np.gcd(NH, NB))
    for _ in range(2 * lcm):
        step()
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    elapsed, host_enqueue_s = timed(step, K)
    window_ms = [elapsed / K * 1e3]
    for _ in range(max(0, args.windows)):
        el, _ = timed(step, K)
        window_ms.append(el / K * 1e3)

    # ---- workload facts (untimed) --------------------------------------------------------------
    torch.cuda.synchronize()
    kept = counts[:B].sum().item()
    prefilter = counts[B:2 * B].sum().item()
    # ROI sides (pixels of the source frame) of the last step on handle 0: what the ROI resize and the classifier were fed
    from litepi.distributed import records_to_numpy
    rec = records_to_numpy(dets)
    cnt_np = counts[:B].cpu().numpy()
    sides = np.concatenate([np.maximum(rec[i, :cnt_np[i]]["x2"] - rec[i, :cnt_np[i]]["x1"], rec[i, :cnt_np[i]]["y2"] - rec[i, :cnt_np[i]]["y1"])
                            for i in range(B)] + [np.zeros(0, np.float32)])
    roi_sides = ({"p50": float(np.percentile(sides, 50)), "p90": float(np.percentile(sides, 90)), "p99": float(np.percentile(sides, 99)),
                  "max": float(sides.max())} if len(sides) else None)

    # ---- host-to-host (PCIe-inclusive): the batch starts in PINNED HOST memory and the records end there ------------
    # The reference's t_total runs from a host image to host results (e2e.py:446-506).  Uploads ride a dedicated copy
    # stream into 2 staging buffers per handle (the copy of step k+NH overlaps the kernels of steps k..k+NH-1); the
    # payload (records + counts, 0.61 MB) leaves through an async D2H into pinned memory on the handle's stream.
    h2d = None
    if not args.no_h2d:
        host = [torch.from_numpy(imgs_np[j]).pin_memory() for j in range(NB)]
        NS = 2 * NH
        stage = [torch.empty((B, SRC, SRC, 3), dtype=torch.uint8, device=dev) for _ in range(NS)]
        host_out = [torch.empty_like(outs[i].payload, device="cpu").pin_memory() for i in range(NH)]
        copy_stream = torch.cuda.Stream(device=dev)
        copied = [torch.cuda.Event() for _ in range(NS)]
        freed = [torch.cuda.Event() for _ in range(NS)]
        hstep_no = [0]

        def hstep():
            k = hstep_no[0]
            hstep_no[0] += 1
            i, j, sidx = k % NH, k % NB, k % NS
            d, c = outs[i]
            with torch.cuda.stream(copy_stream):
                if k >= NS:
                    copy_stream.wait_event(freed[sidx])      # the step that last read this staging buffer is done
                stage[sidx].copy_(host[j], non_blocking=True)
                copied[sidx].record(copy_stream)
            with torch.cuda.stream(streams[i]):
                streams[i].wait_event(copied[sidx])
                engs[i].run_batch_device(stage[sidx].data_ptr(), B, SRC, SRC, run_conf, IOU, MIN_AREA, d.data_ptr(), c.data_ptr())
                freed[sidx].record(streams[i])
                if use_gather:
                    r = gatherers[i].gather(outs[i])
                host_out[i].copy_(outs[i].payload, non_blocking=True)   # records + counts back to the host

        lcm_h = NS * NB // int(np.gcd(NS, NB))
        for _ in range(2 * lcm_h + args.warmup):
            hstep()
        hel, _ = timed(hstep, K)
        nbytes = B * SRC * SRC * 3
        h2d = {"value": world * B * K / hel, "unit": "images/sec", "ms_per_step": hel / K * 1e3,
               "upload_gbs": nbytes * K / hel / 1e9,
               "note": f"host to host: the batch starts in pinned host memory (hipMemcpyAsync of {nbytes / 1e6:.1f} MB per step on a dedicated "
                       "copy stream into double-buffered staging, overlapping the kernels of the steps in flight), the same "
                       "pipeline, then the records + counts (0.61 MB) back into pinned host memory by an async D2H"}

    # ---- the drop-in call itself: HybridPipeline.run_batch on B host NumPy images -> per-image result dicts ----------
    dropin = None
    if rank == 0 and world == 1 and not args.no_dropin:   # a single-GPU measurement (as cpu_baseline): the other ranks do not wait for it
        from litepi import HybridPipeline
        cls_path = os.path.join(workdir, "cls.pth")
        torch.save({k: torch.from_numpy(np.asarray(v)) for k, v in cls_state.items()}, cls_path)
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):   # the constructor prints the reference's banner
            pipe = HybridPipeline(param, binf, cls_path, "shufflenetv2", num_classes=NUM_CLASSES, precision=args.precision,
                                  max_batch=B, max_det=args.max_det, device=local_rank)
        try:
            batches = [[imgs_np[j][i] for i in range(B)] for j in range(NB)]
            for j in range(3):
                pipe.run_batch(batches[j % NB], run_conf, IOU, MIN_AREA)
            n_calls = max(4, min(K, 16))
            t0 = time.perf_counter()
            n_res = 0
            for j in range(n_calls):
                res = pipe.run_batch(batches[j % NB], run_conf, IOU, MIN_AREA)
                n_res += sum(len(r[0]) for r in res)
            dt = time.perf_counter() - t0
            dropin = {"value": B * n_calls / dt, "unit": "images/sec", "ms_per_call": dt / n_calls * 1e3, "calls": n_calls,
                      "results_per_call": n_res / n_calls,
                      "upload_lanes": len(pipe._lanes),
                      "note": "HybridPipeline.run_batch(list of 64 host uint8 images) -> per-image result dicts, synchronous: "
                              "pageable-host upload of 78.6 MB (staged through pinned memory by copy workers), the pipeline, D2H of "
                              "the records, Python dict building (the reference's t_total span, e2e.py:446-506, for a batch); "
                              "upload_lanes > 0: the images are dealt to that many further handles of max_batch / lanes images, "
                              "whose uploads and kernels overlap"}
        finally:
            pipe.close()

    # ---- roofline of the dominant conv kernel (one template instantiation): profiled passes of the same step ----
    roofline, families = None, {}
    if rank == 0 and args.profile_steps > 0:
        torch.cuda.synchronize()
        torch.cuda.set_stream(streams[0])
        launches = []
        for _ in range(args.profile_steps):
            eng.profile_next(True)
            eng.run_batch_device(imgs[0].data_ptr(), B, SRC, SRC, run_conf, IOU, MIN_AREA, dets.data_ptr(), counts.data_ptr())
            torch.cuda.synchronize()
            launches.append(eng.profile_read())
        for run in launches:
            for k in run:
                f = families.setdefault(k["name"], {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "launches": 0})
                f["ms"] += k["ms"]; f["flops"] += k["flops"]; f["bytes"] += k["bytes"]; f["launches"] += 1
        n = float(args.profile_steps)
        for f in families.values():
            f["ms"] /= n; f["flops"] /= n; f["bytes"] /= n; f["launches"] = int(f["launches"] / n)
        det_fams = {k: v for k, v in families.items() if v["flops"] > 0 and not k.startswith("cls_")}
        dom = max(det_fams, key=lambda k: det_fams[k]["ms"])
        f = families[dom]
        tfl = f["flops"] / (f["ms"] * 1e-3) / 1e12
        gbs = f["bytes"] / (f["ms"] * 1e-3) / 1e9
        # the conv path is a dense contraction: SURVEY 8(d) prescribes the fp16 MFMA roof for it; HBM view alongside
        roofline = {"bound": "mfma", "kernel": dom, "achieved": tfl, "peak": PEAK_FP16_TFLOPS, "unit": "TFLOP/s",
                    "frac": tfl / PEAK_FP16_TFLOPS, "traffic": None,
                    "launches_per_step": f["launches"], "avg_launch_ms": f["ms"] / max(f["launches"], 1),
                    "algorithmic_flops_per_launch": f["flops"] / max(f["launches"], 1),
                    "algorithmic_bytes_per_launch": f["bytes"] / max(f["launches"], 1),
                    "flop_per_byte": f["flops"] / f["bytes"] if f["bytes"] > 0 else None,
                    "hbm_view": {"achieved_gbs": gbs, "peak_gbs": PEAK_HBM_GBS, "frac": gbs / PEAK_HBM_GBS},
                    "timing": "HIP events attached to every launch (hipExtLaunchKernelGGL start / stop events on the library's stream) "
                              "of profiled (eager) passes, this process"}
        # HBM bytes per launch from the PMC passes (separate rocprofv3 runs, tools/pmc_profile.sh + tools/pmc_traffic.py;
        # committed under profiles/): offline by nature, read here so that the line carries it next to `achieved`
        tname = f"{PROFILE_ROUND}_pmc_traffic.json" if args.preset == "v1" else f"{PROFILE_ROUND}_pmc_traffic_{args.preset}.json"
        tfile = os.path.join(_ROOT, "profiles", tname)
        if os.path.exists(tfile) and args.precision == "fp16" and B == 64 and args.config == 2:
            with open(tfile) as fh:
                fams = json.load(fh).get("families", {})
            if dom in fams:
                roofline["traffic"] = fams[dom]["hbm_bytes_per_launch"]
                roofline["traffic_source"] = (f"profiles/{tname} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, "
                                              "same workload, separate passes)")
                if "rocprof_avg_launch_ms" in fams[dom]:
                    roofline["avg_launch_ms_rocprof"] = fams[dom]["rocprof_avg_launch_ms"]
            roofline["step_traffic_bytes"] = sum(v["hbm_read_bytes"] + v["hbm_write_bytes"] for v in fams.values())
        conv_fl = sum(v["flops"] for k, v in det_fams.items())
        conv_ms = sum(v["ms"] for k, v in det_fams.items())
        step_ms = sum(v["ms"] for v in families.values())
        roofline["detector_conv_tflops_in_conv_kernels"] = conv_fl / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else None
        roofline["profiled_step_ms"] = step_ms
        roofline["profiled_launches"] = sum(v["launches"] for v in families.values())
        # per kernel instantiation (the profiler's names carry the template arguments, as rocprofv3 lists them) and per family
        roofline["kernels_ms"] = {k: round(v["ms"], 4) for k, v in sorted(families.items(), key=lambda kv: -kv[1]["ms"])}
        fam_ms = {}
        for k, v in families.items():
            base = k.split("<")[0] + (k[k.rindex("_f"):] if "<" in k and "_f" in k else "")
            fam_ms[base] = fam_ms.get(base, 0.0) + v["ms"]
        roofline["families_ms"] = {k: round(v, 4) for k, v in sorted(fam_ms.items(), key=lambda kv: -kv[1])}
        # the dominant FAMILY (all instantiations of one kernel template) next to the dominant instantiation above, so that
        # rounds stay comparable whichever way "dominant" is read
        fam_fl, fam_by = {}, {}
        for k, v in det_fams.items():
            base = k.split("<")[0] + (k[k.rindex("_f"):] if "<" in k and "_f" in k else "")
            fam_fl[base] = fam_fl.get(base, 0.0) + v["flops"]
            fam_by[base] = fam_by.get(base, 0.0) + v["bytes"]
        domf = max(fam_fl, key=lambda k: fam_ms[k])
        roofline["dominant_family"] = {
            "family": domf, "ms": fam_ms[domf], "achieved": fam_fl[domf] / (fam_ms[domf] * 1e-3) / 1e12, "unit": "TFLOP/s",
            "frac": fam_fl[domf] / (fam_ms[domf] * 1e-3) / 1e12 / PEAK_FP16_TFLOPS,
            "hbm_view_gbs": fam_by[domf] / (fam_ms[domf] * 1e-3) / 1e9}
        if args.config == 4 and "letterbox_u8" in families:
            # configs[4]: the letterbox (cv2.resize INTER_LINEAR + 114 border, e2e.py:66-86) is a byte mover -> HBM roof.
            # Algorithmic bytes per image: the 2048 x 2048 x 3 source read once + the 640 x 640 x 3 letterboxed image written once.
            lb = families["letterbox_u8"]
            lb_bytes = float(B) * (SRC * SRC * 3 + 640 * 640 * 3)
            lb_gbs = lb_bytes / (lb["ms"] * 1e-3) / 1e9
            conv_view = {k: roofline[k] for k in ("kernel", "achieved", "peak", "unit", "frac", "avg_launch_ms")}
            conv_view["bound"] = "mfma"
            roofline.update({"bound": "hbm", "kernel": "letterbox_u8", "achieved": lb_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                             "frac": lb_gbs / PEAK_HBM_GBS, "traffic": None, "launches_per_step": lb["launches"],
                             "avg_launch_ms": lb["ms"] / max(lb["launches"], 1), "algorithmic_bytes_per_launch": lb_bytes,
                             "algorithmic_flops_per_launch": 0.0, "flop_per_byte": None,
                             "hbm_view": {"achieved_gbs": lb_gbs, "peak_gbs": PEAK_HBM_GBS, "frac": lb_gbs / PEAK_HBM_GBS},
                             "dominant_conv_kernel": conv_view,
                             "note": "configs[4]: the on-device letterbox is the kernel this configuration adds; 13.81 MB algorithmic "
                                     "per image (12.58 MB source read once + 1.23 MB written)"})
            lbt = os.path.join(_ROOT, "profiles", f"{PROFILE_ROUND}_pmc_traffic_config4.json")
            if os.path.exists(lbt):
                with open(lbt) as fh:
                    fam4 = json.load(fh).get("families", {}).get("letterbox_u8")
                if fam4:
                    roofline["traffic"] = fam4["hbm_bytes_per_launch"]
                    roofline["traffic_source"] = f"profiles/{PROFILE_ROUND}_pmc_traffic_config4.json"
        if args.dump_profile:
            with open(args.dump_profile, "w") as fh:
                json.dump({"families": families, "launches": launches[-1]}, fh, indent=1)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(param, binf, cls_state, imgs_np[0], args.cpu_images)

    if rank == 0:
        total_images = world * B * K
        ms_per_step = elapsed / K * 1e3
        flop_img = 2.0 * eng.det_macs
        if roofline is not None and roofline.get("step_traffic_bytes"):
            roofline["step_hbm_gbs"] = roofline["step_traffic_bytes"] / (ms_per_step * 1e-3) / 1e9
            roofline["step_hbm_frac"] = roofline["step_hbm_gbs"] / PEAK_HBM_GBS
        wm = np.array(window_ms)
        line = {
            "metric": "images/sec end-to-end (det+NMS+clf) 640x640 batch64" if args.config == 2 else
                      "images/sec end-to-end (letterbox+det+NMS+clf) 2048x2048 -> 640x640 batch32 (configs[4])",
            "value": total_images / elapsed,
            "unit": "images/sec",
            "n_gpus": world,
            "steps": K,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f16" if args.precision == "fp16" else "f32",
            "data": "synthetic",
            "config": {
                "workload": (f"configs[4]: TT100K-shape {SRC}x{SRC} uint8 frames resident in HBM, letterboxed to 640 on the device "
                             f"inside the step, full det+NMS+ROI+ShuffleNetV2 pipeline, {args.precision}, batch={B}/GPU"
                             if args.config == 4 else
                             f"configs[2]: full det+NMS+ROI+ShuffleNetV2 pipeline, {args.precision}, batch={B}/GPU, 640x640, "
                             f"inputs resident in HBM") + (f"; configs[3]-style sharding over {world} GPUs, one RCCL gather of "
                                                           f"records per step" if world > 1 else ""),
                "detector": f"YOLO-LitePi {args.preset} architecture, seeded random weights (LSUV-scaled), "
                            f"{flop_img / 1e9:.3f} GFLOP/image, class bias calibrated to ~{TARGET_CANDIDATES} candidates/image"
                            + (f" ({spec['calibration']})" if spec.get("calibration") else ""),
                "classifier": f"ShuffleNetV2 x1.0, {NUM_CLASSES} classes, seeded random weights, 64x64 ROIs",
                "conf": run_conf, "iou": IOU, "min_area": MIN_AREA, "max_det": args.max_det, "steps_in_flight": NH,
                "distinct_input_batches": NB,
                "host_enqueue_ms_per_step": round(host_enqueue_s * 1e3 / K, 4),
                "global_batch": world * B,
                "rois_per_step_rank0": int(kept), "boxes_pre_area_filter_rank0": int(prefilter), "roi_longer_side_px": roi_sides,
                "detector_fp16_roofline_frac_e2e": (total_images / elapsed) * flop_img / (world * PEAK_FP16_TFLOPS * 1e12),
            },
            "windows": {"n": int(len(wm)), "steps_each": K, "ms_per_step_median": float(np.median(wm)),
                        "ms_per_step_p95": float(np.percentile(wm, 95)), "ms_per_step_min": float(wm.min()),
                        "images_per_sec_median": world * B / (float(np.median(wm)) * 1e-3)},
            "h2d_inclusive": h2d,
            "dropin": dropin,
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        diag = {k: os.environ[k] for k in ("LITEPI_SKIP_STAGE", "LITEPI_SKIP_OP") if os.environ.get(k)}
        if diag:   # tools/marginal_cost.sh: a launch is left out of the timed passes -- NOT a valid bench line
            line["INVALID_diagnostic_skip"] = diag
        print(json.dumps(line), flush=True)
    for e in engs:
        e.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
