#!/usr/bin/env python3
"""Headline benchmark: images/sec end-to-end (detector + NMS + ROI + ShuffleNetV2), 640x640,
batch 64 per GPU, fp16 storage / fp32 accumulate (BASELINE.json configs[2]; configs[3] at N>1).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole hot path over one batch of B synthetic uint8 BGR images that
are already resident in HBM (lp_run_batch_device), plus -- at N>1 -- the single RCCL gather of the
detection records to rank 0.  Weak scaling: every rank processes its own B images per step.

Weights: seeded random-init models of the reference's architectures written by
litepi.ncnn_export (YOLO-LitePi v1 widths by default) and a seeded ShuffleNetV2 x1.0; the
class-branch bias is calibrated once, untimed, so that ~8 anchors per image pass conf 0.25 and the
classifier stage has real work.  No dataset/checkpoint is available.

Extra objects on the JSON line:
  roofline     -- dominant kernel family: algorithmic FLOPs of its launches / their HIP-event
                  durations (a profiled pass of the same step, same process), vs the dense fp16
                  MFMA peak (2.5 PFLOP/s, MI355X_MICROARCH.md)
  cpu_baseline -- the CPU restatement of the reference path (oracle/, torch-CPU + NumPy) timed
                  on this host's cores over a bounded sample of the same images (rank 0, N=1)
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


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU per step")
    ap.add_argument("--preset", default="v1", choices=["v1", "v2"])
    ap.add_argument("--precision", default="fp16", choices=["fp16", "fp32"])
    ap.add_argument("--max-det", type=int, default=300)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-images", type=int, default=320, help="images in the CPU baseline sample")
    ap.add_argument("--profile-steps", type=int, default=3)
    ap.add_argument("--inflight", type=int, default=3,
                    help="independent pipeline handles (own stream + activation buffers) the steps rotate over, so "
                         "consecutive steps overlap on the GPU")
    ap.add_argument("--dump-profile", default="", help="write the per-launch profile of the roofline pass to this JSON file")
    return ap.parse_args()


def build_models(args, workdir, engine_factory, imgs_dev):
    """Export the synthetic detector, calibrate its class bias with one untimed GPU pass, and
    return (param, bin, classifier_state, spec)."""
    from litepi import ncnn_export
    from litepi.backend import random_shufflenet_state

    param, binf = os.path.join(workdir, "det.param"), os.path.join(workdir, "det.bin")
    spec = ncnn_export.export_detector(param, binf, args.preset, seed=1234, cls_bias=0.0)
    eng = engine_factory()
    eng.load_detector(param, binf)
    nb = min(8, imgs_dev.shape[0])
    out0 = eng.detect_raw(imgs_dev[:nb].cpu().numpy())
    eng.close()
    s = np.sort(out0[:, 4].astype(np.float64).ravel())[::-1]
    kth = min(max(s[TARGET_CANDIDATES * nb], 1e-6), 1 - 1e-6)
    delta = float(np.log(CONF / (1 - CONF)) - np.log(kth / (1 - kth)))
    ncnn_export.shift_cls_bias(param, binf, delta)
    spec["cls_bias_shift"] = delta
    return param, binf, random_shufflenet_state(NUM_CLASSES, seed=0), spec


def cpu_baseline(param, binf, cls_state, imgs_np, n_images):
    """Reference path restated on CPU (oracle/), timed image by image like e2e.py's loop."""
    from oracle import ncnn_ref, pipeline_ref, shufflenet_ref

    layers = ncnn_ref.load_model(param, binf)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in cls_state.items()}
    model = shufflenet_ref.ShuffleNetV2(NUM_CLASSES)
    missing = model.load_state_dict(sd, strict=False)
    assert not [k for k in missing.missing_keys if "num_batches_tracked" not in k], missing
    model.eval()
    pipe = pipeline_ref.CpuPipeline(layers, model)
    # the GPU box gives one GPU a 16-core CPU share; oneDNN at batch 1 does not scale past that
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    cores = torch.get_num_threads()
    pipe.run(imgs_np[0], CONF, IOU, MIN_AREA)  # warm-up
    t0 = time.perf_counter()
    ndet = 0
    for i in range(n_images):
        res, _ = pipe.run(imgs_np[i % len(imgs_np)], CONF, IOU, MIN_AREA)
        ndet += len(res)
    dt = time.perf_counter() - t0
    return {"value": n_images / dt, "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"{n_images} of the bench images, batch 1, fp32 torch-CPU convs + NumPy post-processing "
                      f"(oracle/pipeline_ref.py), {ndet} classified ROIs, {dt:.1f} s"}


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from litepi import Engine
    from litepi.distributed import Gatherer, alloc_result_buffers

    B = args.batch
    rng = np.random.default_rng(1 + rank)
    imgs_np = rng.integers(0, 256, (B, 640, 640, 3), dtype=np.uint8)
    imgs = torch.from_numpy(imgs_np).to(dev)

    def engine_factory():
        return Engine(precision=args.precision, max_batch=B, max_det=args.max_det, num_classes=NUM_CLASSES,
                      device=local_rank)

    workdir = tempfile.mkdtemp(prefix=f"litepi_bench_r{rank}_")
    # every rank calibrates on rank 0's images so that all replicas are identical
    cal_imgs = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (8, 640, 640, 3), dtype=np.uint8)).to(dev)
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
    eng, (dets, counts) = engs[0], outs[0]
    gatherers = [Gatherer(o, dst=0) for o in outs]   # receive slots allocated once (rank 0), nothing per step
    torch.cuda.synchronize()
    step_no = [0]

    def step():
        i = step_no[0] % len(engs)
        step_no[0] += 1
        d, c = outs[i]
        with torch.cuda.stream(streams[i]):
            engs[i].run_batch_device(imgs.data_ptr(), B, 640, 640, CONF, IOU, MIN_AREA, d.data_ptr(), c.data_ptr())
            if world > 1:
                return gatherers[i].gather(outs[i])
        return d, c.view(1, -1)

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        gathered = step()
    host_enqueue_s = time.perf_counter() - t0   # host time to enqueue all timed steps (no GPU wait unless a queue fills)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- workload facts (untimed) --------------------------------------------------------------
    kept = counts[:B].sum().item()
    prefilter = counts[B:2 * B].sum().item()

    # ---- roofline of the dominant kernel family: profiled passes of the same step ---------------
    roofline, families = None, {}
    if rank == 0 and args.profile_steps > 0:
        torch.cuda.set_stream(streams[0])
        launches = []
        for _ in range(args.profile_steps):
            eng.profile_next(True)
            eng.run_batch_device(imgs.data_ptr(), B, 640, 640, CONF, IOU, MIN_AREA, dets.data_ptr(), counts.data_ptr())
            torch.cuda.synchronize()
            launches.append(eng.profile_read())
        for run in launches:
            for k in run:
                f = families.setdefault(k["name"], {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "launches": 0})
                f["ms"] += k["ms"]; f["flops"] += k["flops"]; f["bytes"] += k["bytes"]; f["launches"] += 1
        n = float(args.profile_steps)
        for f in families.values():
            f["ms"] /= n; f["flops"] /= n; f["bytes"] /= n; f["launches"] = int(f["launches"] / n)
        dom = max(families, key=lambda k: families[k]["ms"])
        f = families[dom]
        # which roof bounds the dominant family: its algorithmic intensity against the ridge (2.5 PFLOP/s / 8 TB/s)
        ridge = PEAK_FP16_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9)
        tfl = f["flops"] / (f["ms"] * 1e-3) / 1e12
        gbs = f["bytes"] / (f["ms"] * 1e-3) / 1e9
        if f["flops"] > 0 and f["bytes"] > 0 and f["flops"] / f["bytes"] >= ridge:
            roofline = {"bound": "mfma", "kernel": dom, "achieved": tfl, "peak": PEAK_FP16_TFLOPS, "unit": "TFLOP/s",
                        "frac": tfl / PEAK_FP16_TFLOPS, "traffic": None}
        else:
            roofline = {"bound": "hbm", "kernel": dom, "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": gbs / PEAK_HBM_GBS, "traffic": None}
        roofline["flop_per_byte"] = f["flops"] / f["bytes"] if f["bytes"] > 0 else None
        roofline["achieved_tflops"] = tfl
        roofline["algorithmic_bytes_per_launch"] = f["bytes"] / max(f["launches"], 1)
        # HBM bytes per launch from the PMC passes (separate rocprofv3 runs, tools/pmc_profile.sh + tools/pmc_traffic.py;
        # committed under profiles/): offline by nature, read here so that the line carries it next to `achieved`
        tfile = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_traffic.json")
        if os.path.exists(tfile) and args.preset == "v1" and args.precision == "fp16" and B == 64:
            with open(tfile) as fh:
                fam = json.load(fh).get("families", {}).get(dom)
            if fam:
                roofline["traffic"] = fam["hbm_bytes_per_launch"]
                roofline["traffic_source"] = "profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, same workload)"
            with open(tfile) as fh:
                fams = json.load(fh).get("families", {})
            step_bytes = sum(v["hbm_read_bytes"] + v["hbm_write_bytes"] for v in fams.values())
            # whole-step view: every launch's PMC traffic over the measured step time (set below, once elapsed is known)
            roofline["step_traffic_bytes"] = step_bytes
        roofline["launches_per_step"] = f["launches"]
        roofline["avg_launch_ms"] = f["ms"] / max(f["launches"], 1)
        roofline["algorithmic_per_step"] = f["flops"] if f["flops"] > 0 else f["bytes"]
        conv_fl = sum(v["flops"] for k, v in families.items() if k.startswith(("conv", "stem")))
        step_ms = sum(v["ms"] for v in families.values())
        roofline["detector_conv_tflops_over_whole_step"] = conv_fl / (step_ms * 1e-3) / 1e12 if step_ms > 0 else None
        roofline["profiled_step_ms"] = step_ms
        if args.dump_profile:
            with open(args.dump_profile, "w") as fh:
                json.dump({"families": families, "launches": launches[-1]}, fh, indent=1)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(param, binf, cls_state, imgs_np, args.cpu_images)

    if rank == 0:
        total_images = world * B * args.steps
        ms_per_step = elapsed / args.steps * 1e3
        flop_img = 2.0 * eng.det_macs
        if roofline is not None and roofline.get("step_traffic_bytes"):
            roofline["step_hbm_gbs"] = roofline["step_traffic_bytes"] / (ms_per_step * 1e-3) / 1e9
            roofline["step_hbm_frac"] = roofline["step_hbm_gbs"] / PEAK_HBM_GBS
        line = {
            "metric": "images/sec end-to-end (det+NMS+clf) 640x640 batch64",
            "value": total_images / elapsed,
            "unit": "images/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f16" if args.precision == "fp16" else "f32",
            "data": "synthetic",
            "config": {
                "workload": f"configs[2]: full det+NMS+ROI+ShuffleNetV2 pipeline, {args.precision}, batch={B}/GPU, 640x640, "
                            f"inputs resident in HBM" + (f"; configs[3]-style sharding over {world} GPUs, one RCCL gather of "
                                                         f"records per step" if world > 1 else ""),
                "detector": f"YOLO-LitePi {args.preset} architecture, seeded random weights (LSUV-scaled), "
                            f"{flop_img / 1e9:.3f} GFLOP/image, class bias calibrated to ~{TARGET_CANDIDATES} candidates/image",
                "classifier": f"ShuffleNetV2 x1.0, {NUM_CLASSES} classes, seeded random weights, 64x64 ROIs",
                "conf": CONF, "iou": IOU, "min_area": MIN_AREA, "max_det": args.max_det, "steps_in_flight": len(engs),
                "host_enqueue_ms_per_step": round(host_enqueue_s * 1e3 / args.steps, 4),
                "global_batch": world * B,
                "rois_per_step_rank0": int(kept), "boxes_pre_area_filter_rank0": int(prefilter),
                "detector_fp16_roofline_frac_e2e": (total_images / elapsed) * flop_img / (world * PEAK_FP16_TFLOPS * 1e12),
            },
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    for e in engs:
        e.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
