#!/usr/bin/env python3
"""End-to-end evaluation CLI with the argument surface of the reference's
``src/tt100k/pipeline/e2e.py`` (flags and defaults: e2e.py:1017-1048), running on the HIP
backend.  Same two passes per image (benchmark pass at ``--benchmark_conf`` whose time feeds the
FPS figure, evaluation pass at ``--yolo_conf`` that feeds mAP; e2e.py:955-1011), same
Ultralytics-style metric (e2e.py:656-824) and the same appended ``comparison_summary.csv`` schema
(e2e.py:1166-1184).  New flags: ``--batch_images``, ``--precision``, ``--hip_device``, ``--gpus``.

``--gpus N`` (launched as ``python -m torch.distributed.run --nproc-per-node N -m litepi.e2e ... --gpus N``): one process
per GPU, the sorted image list is sharded contiguously (``distributed.shard_range``), every rank runs both passes on its
shard with a full weight replica, the per-image predictions and ground truths are gathered to rank 0 once
(``distributed.gather_eval_shards``, backend nccl = RCCL), and rank 0 alone scores, prints and writes the CSV.  The FPS
figure of a multi-rank run is images / the slowest rank's summed benchmark time (the wall time of the sharded job).

Differences, on purpose: images are decoded with Pillow (cv2 is not a dependency);
``--detector_threads`` and ``--device`` are accepted and ignored (everything runs on the GPU);
all four ``--clf_arch`` choices run on the GPU; ``--save_viz`` overlays (e2e.py:826-884, 1003-1009) are drawn with Pillow: same
layout and colours as the reference's cv2 drawing (ground truth blue, predictions green, label boxes, summary bar), Pillow's
default font instead of Hershey.
"""
from __future__ import annotations

import argparse
import json
import os
import random
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np


# ------------------------------------------------------------------------------------------------
def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="E2E HIP (MI355X) detector + classifier evaluation pipeline")
    p.add_argument("--detector_param", type=str, default="../convert/model/yolo_plus/yolo_plus_ncnn_model/model.ncnn.param")
    p.add_argument("--detector_bin", type=str, default="../convert/model/yolo_plus/yolo_plus_ncnn_model/model.ncnn.bin")
    p.add_argument("--classifier", type=str, default="../weight/shufflenetv2.pth")
    p.add_argument("--clf_arch", type=str, choices=["resnet18", "efficientnet", "mobilenetv2", "shufflenetv2"], default="shufflenetv2")
    p.add_argument("--input", type=str, default="../Dataset/E2E/data_e2e_tt100k/images/test")
    p.add_argument("--labels", type=str, default="../Dataset/E2E/data_e2e_tt100k/labels/test")
    p.add_argument("--classes", type=str, default="../Dataset/E2E/data_e2e_tt100k/idx2label.json")
    p.add_argument("--num_samples", type=int, default=None)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--yolo_conf", type=float, default=0.001)
    p.add_argument("--benchmark_conf", type=float, default=0.25)
    p.add_argument("--min_area", type=int, default=50)
    p.add_argument("--iou_threshold", type=float, default=0.45)
    p.add_argument("--det_input_size", type=int, default=640)
    p.add_argument("--cls_input_size", type=int, default=64)
    p.add_argument("--detector_threads", type=int, default=4, help="ignored (NCNN threads in the reference)")
    p.add_argument("--batch_size", type=int, default=8, help="kept for compatibility: all ROIs of a call are classified together")
    p.add_argument("--device", type=str, choices=["cpu", "cuda", "hip"], default="hip", help="ignored: always the HIP device")
    p.add_argument("--output", type=str, default="output_eval")
    p.add_argument("--save_viz", default=False, help="write visualizations/vis_<image>.png overlays like the reference (any non-empty value)")
    # new
    p.add_argument("--batch_images", type=int, default=1, help="images per GPU call")
    p.add_argument("--precision", type=str, choices=["fp16", "fp32"], default="fp16")
    p.add_argument("--hip_device", type=int, default=0)
    p.add_argument("--gpus", type=int, default=1,
                   help="GPUs of this node to shard the image list over: one process per GPU under torch.distributed.run "
                        "(WORLD_SIZE must equal --gpus; the rank's GPU is LOCAL_RANK and overrides --hip_device)")
    p.add_argument("--max_det", type=int, default=0,
                   help="detections kept per image; 0 = one slot per anchor, i.e. nothing is ever dropped (what the reference does: "
                        "e2e.py:280-296 keeps every NMS survivor, and the evaluation pass at --yolo_conf 0.001 produces thousands)")
    p.add_argument("--max_rois", type=int, default=0, help="classifier capacity per call; 0 = batch_images x max_det")
    p.add_argument("--numerics", type=str, choices=["e2e", "e2e_optimize"], default="e2e",
                   help="which reference pipeline's ROI stage to follow: e2e.py (PIL antialiased resize) or e2e_optimize.py "
                        "(cv2-linear resize, its own clip rule)")
    # e2e_optimize.py's extras (e2e_optimize.py:882-885)
    p.add_argument("--warmup", type=int, default=None,
                   help="warm-up passes on a random 640x640 frame before the timed loop (e2e_optimize.py:552-570 warmup_pipeline); "
                        "default 10 under --numerics e2e_optimize as there, 0 otherwise (e2e.py has none).  Here they also take "
                        "the hipGraph capture of the benchmark pass out of the first timed image")
    p.add_argument("--no_jit", action="store_true", help="accepted and ignored: there is no TorchScript on this path (e2e_optimize.py:884)")
    return p


def num_anchors(det_input_size: int) -> int:
    """Anchor points of the Detect head on a square input: one per cell of the stride-8/16/32 grids (8400 at 640)."""
    return sum(((det_input_size + s - 1) // s) ** 2 for s in (8, 16, 32))


def load_class_names(path: str) -> List[str]:
    """JSON {"idx": "name"} or one name per line (e2e.py:160-176)."""
    with open(path, "r") as f:
        text = f.read()
    try:
        data = json.loads(text)
    except json.JSONDecodeError:
        return [ln.strip() for ln in text.splitlines() if ln.strip()]
    if not isinstance(data, dict):
        raise ValueError("JSON must be a dictionary")
    names = [""] * (max(int(k) for k in data) + 1)
    for k, v in data.items():
        names[int(k)] = v
    return names


def parse_yolo_label(label_path, img_w: int, img_h: int) -> List[Tuple[int, int, int, int, int]]:
    """``cls xc yc w h`` normalised -> (cls, x1, y1, x2, y2) int pixels (e2e.py:137-157)."""
    if not os.path.exists(label_path):
        return []
    out = []
    with open(label_path) as f:
        for line in f:
            t = line.split()
            if len(t) < 5:
                continue
            xc, yc, w, h = (float(v) for v in t[1:5])
            out.append((int(t[0]), int((xc - w / 2) * img_w), int((yc - h / 2) * img_h), int((xc + w / 2) * img_w),
                        int((yc + h / 2) * img_h)))
    return out


def sample_images(files: Sequence, num_samples: Optional[int], seed: int = 42):
    """Deterministic subset (e2e.py:179-186)."""
    if num_samples is None or num_samples <= 0 or num_samples >= len(files):
        return list(files)
    random.seed(seed)
    return sorted(random.sample(list(files), num_samples))


def visualize_prediction(img_bgr: np.ndarray, predictions: Sequence[Dict], ground_truths, class_names: Sequence[str], output_path) -> None:
    """Ground truths (blue) and predictions (green) drawn on the image, like the reference's visualize_prediction
    (e2e.py:826-884): 3-px rectangles, "GT: <name>" above each ground truth, "PRED: <name>" and "Cls:x Det:y" below each
    prediction on filled label boxes, and the "GT: n | Predictions: m" bar in the top-left corner.  Drawn with Pillow (RGB)."""
    from PIL import Image, ImageDraw, ImageFont
    im = Image.fromarray(np.ascontiguousarray(img_bgr[:, :, ::-1]))
    d = ImageDraw.Draw(im)
    font = ImageFont.load_default()
    h, w = img_bgr.shape[:2]
    BLUE, GREEN, DGREEN, WHITE, BLACK = (0, 0, 255), (0, 255, 0), (0, 200, 0), (255, 255, 255), (0, 0, 0)

    def name_of(i):
        return class_names[i] if 0 <= int(i) < len(class_names) else str(i)

    def label(x, y_base, text, fill):
        l, t, r, b = d.textbbox((0, 0), text, font=font)
        tw, th = r - l, b - t
        d.rectangle([x, y_base - th - 4, x + tw + 2, y_base + 2], fill=fill)
        d.text((x + 1, y_base - th - 2), text, fill=WHITE, font=font)
        return th

    for gt_cls, x1, y1, x2, y2 in ground_truths:
        d.rectangle([x1, y1, x2, y2], outline=BLUE, width=3)
        label(x1, max(y1 - 10, 15), f"GT: {name_of(gt_cls)}", BLUE)
    for pred in predictions:
        x1, y1, x2, y2 = [int(v) for v in pred["bbox"]]
        d.rectangle([x1, y1, x2, y2], outline=GREEN, width=3)
        ty = min(y2 + 25, h - 5)
        th = label(x1, ty, f"PRED: {name_of(pred['cls_class'])}", GREEN)
        label(x1, ty + th + 7, f"Cls:{pred['cls_conf']:.2f} Det:{pred['det_conf']:.2f}", DGREEN)
    d.rectangle([5, 5, 400, 35], fill=BLACK)
    d.text((10, 14), f"GT: {len(ground_truths)} | Predictions: {len(predictions)}", fill=WHITE, font=font)
    im.save(str(output_path))


def read_image_bgr(path) -> Optional[np.ndarray]:
    from PIL import Image
    try:
        with Image.open(path) as im:
            return np.ascontiguousarray(np.asarray(im.convert("RGB"))[:, :, ::-1])
    except Exception:  # noqa: BLE001 - unreadable image -> skipped, like cv2.imread returning None
        return None


# ------------------------------------------------------------------------------------------------
def _pairwise_iou(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt = np.maximum(a[:, None, :2], b[:, :2])
    rb = np.minimum(a[:, None, 2:], b[:, 2:])
    wh = (rb - lt).clip(min=0)
    inter = wh[..., 0] * wh[..., 1]
    return inter / (area_a[:, None] + area_b - inter + 1e-7)


def _ap101(recall: np.ndarray, precision: np.ndarray) -> float:
    mrec = np.concatenate(([0.0], recall, [1.0]))
    mpre = np.concatenate(([1.0], precision, [0.0]))
    mpre = np.flip(np.maximum.accumulate(np.flip(mpre)))
    x = np.linspace(0, 1, 101)
    trapz = getattr(np, "trapezoid", None) or np.trapz
    return float(trapz(np.interp(x, mrec, mpre), x))


def evaluate_predictions(all_preds, all_gts, num_classes, iou_threshold=0.5, iou_thresholds=np.arange(0.5, 1.0, 0.05)) -> Dict:
    """Ultralytics-style global evaluation with the semantics of the reference (e2e.py:656-824):
    per IoU threshold, matches are ranked by IoU, de-duplicated per prediction then per ground
    truth, and count only when the CLASSIFIER class equals the label; predictions are ranked
    globally by detector confidence; AP is the 101-point interpolated area; the means run over the
    classes present in the ground truth.  ``iou_threshold`` is unused, as in the reference."""
    nt = len(iou_thresholds)
    stats = []
    for preds, gts in zip(all_preds, all_gts):
        if len(preds) == 0:
            if len(gts) > 0:
                stats.append((np.zeros((0, nt), bool), np.array([]), np.array([]), np.array(gts)[:, 0]))
            continue
        pb = np.array([p["bbox"] for p in preds])
        pconf = np.array([p["conf"] for p in preds])
        pcls = np.array([p["cls_class"] for p in preds])
        correct = np.zeros((len(preds), nt), bool)
        if len(gts) > 0:
            g = np.array(gts)
            tcls, tb = g[:, 0], g[:, 1:]
            iou = _pairwise_iou(pb, tb)
            for j, thr in enumerate(iou_thresholds):
                pi, gi = np.where(iou >= thr)
                if not len(pi):
                    continue
                m = np.concatenate((np.stack((pi, gi), 1), iou[pi, gi][:, None]), 1)
                if len(pi) > 1:
                    m = m[m[:, 2].argsort()[::-1]]
                    m = m[np.unique(m[:, 0], return_index=True)[1]]
                    m = m[np.unique(m[:, 1], return_index=True)[1]]
                for p_i, g_i, _ in m:
                    if pcls[int(p_i)] == tcls[int(g_i)]:
                        correct[int(p_i), j] = True
        else:
            tcls = np.array([])
        stats.append((correct, pconf, pcls, tcls))

    zeros = lambda: np.zeros(num_classes)  # noqa: E731
    if not stats:
        return {"precision": zeros(), "recall": zeros(), "f1": zeros(), "tp": zeros(), "fp": zeros(), "fn": zeros(),
                "mAP50": 0.0, "mAP50_95": 0.0, "classes_present": np.zeros(num_classes, dtype=bool)}
    tp = np.concatenate([s[0] for s in stats], 0)
    conf = np.concatenate([s[1] for s in stats], 0)
    pcls = np.concatenate([s[2] for s in stats], 0)
    tcls = np.concatenate([s[3] for s in stats], 0)
    order = np.argsort(-conf)
    tp, pcls = tp[order], pcls[order]
    uniq, counts = np.unique(tcls, return_counts=True)
    n_gt_of = dict(zip(uniq, counts))
    ap50, ap5095, pbest, rbest, fbest = zeros(), zeros(), zeros(), zeros(), zeros()
    tpc_, fpc_, fnc_ = zeros(), zeros(), zeros()
    for c in range(num_classes):
        n_gt = n_gt_of.get(c, 0)
        sel = pcls == c
        n_p = sel.sum()
        if n_p == 0 and n_gt == 0:
            continue
        if n_p == 0 or n_gt == 0:
            fnc_[c] = n_gt
            continue
        tpc = tp[sel].cumsum(0)
        fpc = (1 - tp[sel]).cumsum(0)
        rec = tpc / (n_gt + 1e-16)
        prec = tpc / (tpc + fpc + 1e-16)
        aps = [_ap101(rec[:, j], prec[:, j]) for j in range(tp.shape[1])]
        ap50[c], ap5095[c] = aps[0], np.mean(aps)
        f1 = 2 * prec[:, 0] * rec[:, 0] / (prec[:, 0] + rec[:, 0] + 1e-16)
        b = int(np.argmax(f1))
        pbest[c], rbest[c], fbest[c] = prec[b, 0], rec[b, 0], f1[b]
        tpc_[c], fpc_[c] = tpc[b, 0], fpc[b, 0]
        fnc_[c] = n_gt - tpc_[c]
    present = uniq.astype(int)
    return {"precision": pbest, "recall": rbest, "f1": fbest, "tp": tpc_, "fp": fpc_, "fn": fnc_,
            "mAP50": float(np.mean(ap50[present])) if len(present) else 0.0,
            "mAP50_95": float(np.mean(ap5095[present])) if len(present) else 0.0,
            "ap50_per_class": ap50, "classes_present": np.isin(np.arange(num_classes), uniq)}


# ------------------------------------------------------------------------------------------------
def run_evaluation(args) -> Dict:
    """The reference's main loop (e2e.py:1090-1130) over image CHUNKS of --batch_images: per chunk one benchmark pass at
    --benchmark_conf (its wall time feeds the FPS figure) and, unless the two thresholds are equal, one evaluation pass at
    --yolo_conf whose detections feed the metric (process_image, e2e.py:953-1011).  Returns everything main() prints/writes."""
    from .backend import HybridPipeline

    from . import distributed as D

    rank, local_rank, world = D.env_rank_world()
    if args.gpus > 1 or world > 1:
        import torch
        import torch.distributed as dist
        if world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} needs one process per GPU: launch with torch.distributed.run --nproc-per-node "
                             f"{args.gpus} (WORLD_SIZE is {world})")
        backend = os.environ.get("LITEPI_DIST_BACKEND", "nccl")   # "gloo": the CPU tests of the shard / merge logic
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29541")
            if backend == "nccl":
                torch.cuda.set_device(local_rank)
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)
        args.hip_device = local_rank
        gather_device = torch.device("cuda", local_rank) if backend == "nccl" else "cpu"
    else:
        gather_device = "cpu"
    say = print if rank == 0 else (lambda *a, **k: None)

    class_names = load_class_names(args.classes)
    num_classes = len(class_names)
    detector_name = Path(args.detector_param).stem
    combo = f"{detector_name}+{args.clf_arch}"
    say(f"\n{'=' * 60}\nMODEL COMBINATION: {combo}\n{'=' * 60}")
    nb = max(1, args.batch_images)
    max_det = args.max_det if args.max_det > 0 else num_anchors(args.det_input_size)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()) if rank != 0 else contextlib.nullcontext():   # one banner, rank 0's
        pipeline = HybridPipeline(args.detector_param, args.detector_bin, args.classifier, args.clf_arch, num_classes,
                                  args.det_input_size, args.cls_input_size, False, args.detector_threads, args.device,
                                  args.batch_size, precision=args.precision, max_batch=nb, max_det=max_det,
                                  device=args.hip_device, max_rois=args.max_rois, numerics=args.numerics)
    out_dir = Path(args.output) / combo
    out_dir.mkdir(parents=True, exist_ok=True)

    inp = Path(args.input)
    if inp.is_file():
        files = [inp]
        label_dir = Path(args.labels) if args.labels else None
    else:
        label_dir = Path(args.labels) if args.labels else inp / "labels"
        files = sorted(list(inp.glob("*.jpg")) + list(inp.glob("*.png")) + list(inp.glob("*.jpeg")))
        if args.num_samples:
            files = sample_images(files, args.num_samples, args.seed)
    say(f"\nFound {len(files)} images for processing" + (f" ({world} ranks, contiguous shards)" if world > 1 else ""))
    n_total = len(files)
    lo, hi = D.shard_range(n_total, rank, world)
    files = files[lo:hi]

    all_preds, all_gts, bench_time, names = [], [], 0.0, []
    try:
        n_warm = args.warmup if args.warmup is not None else (10 if args.numerics == "e2e_optimize" else 0)
        if n_warm > 0:   # warmup_pipeline (e2e_optimize.py:552-570): random frame, conf 0.5
            dummy = np.random.randint(0, 255, (640, 640, 3), dtype=np.uint8)
            for _ in range(n_warm):
                pipeline.run(dummy, conf_threshold=0.5)
            say(f"Warmup complete ({n_warm} passes)")
        for i in range(0, len(files), nb):
            chunk, imgs = [], []
            for f in files[i:i + nb]:
                im = read_image_bgr(f)
                if im is None:
                    print(f"\nSkipping {f.name}")
                    continue
                chunk.append(f)
                imgs.append(im)
            if not imgs:
                continue
            bench = pipeline.run_batch(imgs, args.benchmark_conf, args.iou_threshold, args.min_area)
            bench_time += sum(m.t_total for _, m in bench) / 1000.0
            ev = bench if args.yolo_conf == args.benchmark_conf else pipeline.run_batch(imgs, args.yolo_conf, args.iou_threshold, args.min_area)
            for f, im, (res, _) in zip(chunk, imgs, ev):
                lp = (label_dir / f"{f.stem}.txt") if label_dir else f.parent / "labels" / f"{f.stem}.txt"
                all_gts.append(parse_yolo_label(lp, im.shape[1], im.shape[0]))
                if args.save_viz:   # overlays of the evaluation pass, e2e.py:1003-1009
                    viz_dir = out_dir / "visualizations"
                    viz_dir.mkdir(parents=True, exist_ok=True)
                    visualize_prediction(im, res, all_gts[-1], class_names, viz_dir / f"vis_{f.stem}.png")
                all_preds.append([{"bbox": r["bbox"], "conf": r.get("det_conf", 0.0), "cls_class": r.get("cls_class", -1)} for r in res])
                names.append(f.name)
    finally:
        pipeline.engine.close()
    rank_times = [bench_time]
    if world > 1:   # the one exchange of the sharded evaluation: every rank's predictions + ground truths -> rank 0
        import torch.distributed as dist
        got = D.gather_eval_shards(all_preds, all_gts, bench_time, device=gather_device)
        dist.barrier()
        if rank != 0:
            return {"rank": rank, "files": names}
        all_preds, all_gts, rank_times = got
    m = evaluate_predictions(all_preds, all_gts, num_classes, args.iou_threshold)
    # one rank: the reference's figure, mean per-image benchmark time; N ranks: the sharded job takes as long as its slowest rank
    avg = max(rank_times) / len(all_preds) if all_preds else 0.0
    return {"combo": combo, "detector": detector_name, "class_names": class_names, "metrics": m, "avg_time": avg,
            "fps": 1.0 / avg if avg > 0 else 0.0, "all_preds": all_preds, "all_gts": all_gts, "files": names, "max_det": max_det,
            "rank": 0, "world": world, "rank_bench_times": rank_times}


def main(argv=None) -> int:
    import pandas as pd

    args = build_parser().parse_args(argv)
    r = run_evaluation(args)
    if r.get("rank", 0) != 0:   # --gpus N: rank 0 alone scores, prints and writes
        return 0
    combo, m, class_names, fps, avg = r["combo"], r["metrics"], r["class_names"], r["fps"], r["avg_time"]
    print("\n" + "=" * 80 + f"\nEVALUATION RESULTS - {combo}\n" + "=" * 80)
    print(f"\nPerformance Metrics (at conf={args.benchmark_conf}):\n  Avg Inference Time: {avg * 1000:.2f} ms\n  Real FPS:           {fps:.2f} FPS")
    print(f"\nAccuracy Metrics (at conf={args.yolo_conf}):")
    print(f"{'Class':<20} {'Precision':>10} {'Recall':>10} {'F1':>10} {'TP':>6} {'FP':>6} {'FN':>6}\n" + "-" * 80)
    valid = m["classes_present"]
    for c, name in enumerate(class_names):
        if valid[c]:
            print(f"{name:<20} {m['precision'][c]:>10.3f} {m['recall'][c]:>10.3f} {m['f1'][c]:>10.3f} "
                  f"{int(m['tp'][c]):>6} {int(m['fp'][c]):>6} {int(m['fn'][c]):>6}")
    mean = lambda k: float(np.mean(m[k][valid])) if valid.any() else 0.0  # noqa: E731
    print("-" * 80 + f"\n{'MEAN':<20} {mean('precision'):>10.3f} {mean('recall'):>10.3f} {mean('f1'):>10.3f}")
    print(f"{'mAP@0.5':<20} {m['mAP50']:>10.3f}\n{'mAP@0.5:0.95':<20} {m['mAP50_95']:>10.3f}")
    row = pd.DataFrame([{"model_combination": combo, "detector": r["detector"], "classifier": args.clf_arch,
                         "num_test_images": len(r["all_preds"]), "mean_precision": mean("precision"), "mean_recall": mean("recall"),
                         "mean_f1": mean("f1"), "fps": fps, "mAP50": m["mAP50"], "mAP50-95": m["mAP50_95"]}])
    summary = Path(args.output) / "comparison_summary.csv"
    if summary.exists():
        row = pd.concat([pd.read_csv(summary), row], ignore_index=True)
    row.to_csv(summary, index=False)
    print(f"Updated comparison summary at {summary}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
