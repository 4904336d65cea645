"""GPU test of the evaluation harness (litepi.e2e, the drop-in for the reference's src/tt100k/pipeline/e2e.py main()):
runs main() on a temporary directory of synthetic PNGs + YOLO labels and checks the two-pass semantics (e2e.py:953-1011), the
comparison_summary.csv schema and append behaviour (e2e.py:1166-1184), the keep-everything evaluation pass (the reference has
no detection cap: e2e.py:280-296) and the PipelineMetrics fields the reference fills (e2e.py:454-457, 509-516)."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-litepi_amd"))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

CSV_COLUMNS = ["model_combination", "detector", "classifier", "num_test_images", "mean_precision", "mean_recall", "mean_f1", "fps",
               "mAP50", "mAP50-95"]   # e2e.py:1166-1177


def _dataset(tmp_path, n_img=4, ncls=5):
    from PIL import Image
    import torch
    from litepi import ncnn_export
    from oracle import shufflenet_ref as S
    p, b = str(tmp_path / "synthetic.ncnn.param"), str(tmp_path / "synthetic.ncnn.bin")
    # class bias 0: thousands of anchors pass conf 0.001 (far more than 300 NMS survivors per image), a few pass 0.25 after
    # the calibration below
    ncnn_export.export_detector(p, b, "v1", seed=1234, cls_bias=0.0)
    rng = np.random.default_rng(11)
    img_dir, lab_dir = tmp_path / "images", tmp_path / "labels"
    img_dir.mkdir(); lab_dir.mkdir()
    imgs = []
    for i in range(n_img):
        hw = (640, 640) if i < n_img - 1 else (360, 480)   # the last one exercises the on-device letterbox
        im = rng.integers(0, 256, (hw[0], hw[1], 3), dtype=np.uint8)
        Image.fromarray(im[:, :, ::-1]).save(img_dir / f"im{i}.png")   # files hold RGB; the harness hands BGR to the pipeline
        imgs.append(im)
        with open(lab_dir / f"im{i}.txt", "w") as f:
            for _ in range(3):
                xc, yc, w, h = rng.uniform(0.2, 0.8), rng.uniform(0.2, 0.8), rng.uniform(0.05, 0.2), rng.uniform(0.05, 0.2)
                f.write(f"{int(rng.integers(0, ncls))} {xc:.6f} {yc:.6f} {w:.6f} {h:.6f}\n")
    classes = tmp_path / "idx2label.json"
    classes.write_text(json.dumps({str(i): f"sign_{i}" for i in range(ncls)}))
    cls_path = str(tmp_path / "cls.pth")
    torch.save(S.seeded_state_dict(ncls), cls_path)
    return p, b, cls_path, img_dir, lab_dir, classes, imgs


def _calibrate(p, b, imgs, target):
    """Shift the class-projection biases so that about `target` anchors per image pass conf 0.25 (as bench.py does)."""
    from litepi import Engine, ncnn_export
    e = Engine(precision="fp16", max_batch=len(imgs))
    e.load_detector(p, b)
    s = np.sort(e.detect_raw(np.stack(imgs))[:, 4].ravel())[::-1]
    e.close()
    thr = float(s[target * len(imgs)])
    ncnn_export.shift_cls_bias(p, b, float(np.log(0.25 / 0.75) - np.log(thr / (1 - thr))))


def test_harness_main_two_passes_csv_and_keep_all(tmp_path, monkeypatch, capsys):
    import pandas as pd
    from litepi import backend, e2e
    from oracle import postprocess_ref as P
    p, b, cls_path, img_dir, lab_dir, classes, imgs = _dataset(tmp_path)
    _calibrate(p, b, imgs[:3], 6)
    out = tmp_path / "out"
    argv = ["--detector_param", p, "--detector_bin", b, "--classifier", cls_path, "--clf_arch", "shufflenetv2", "--input", str(img_dir),
            "--labels", str(lab_dir), "--classes", str(classes), "--output", str(out), "--batch_images", "2"]

    calls, made = [], []
    real_run_batch = backend.HybridPipeline.run_batch
    real_init = backend.HybridPipeline.__init__

    def spy_init(self, *a, **k):
        real_init(self, *a, **k)
        made.append(self)

    def spy_run_batch(self, images, conf_threshold=0.5, iou_threshold=0.45, min_area=100):
        r = real_run_batch(self, images, conf_threshold, iou_threshold, min_area)
        calls.append((len(images), float(conf_threshold), float(iou_threshold), int(min_area), [len(res) for res, _ in r],
                      [m for _, m in r]))
        return r

    monkeypatch.setattr(backend.HybridPipeline, "__init__", spy_init)
    monkeypatch.setattr(backend.HybridPipeline, "run_batch", spy_run_batch)

    assert e2e.main(argv) == 0
    text = capsys.readouterr().out
    assert "Real FPS" in text and "mAP@0.5:0.95" in text and "MODEL COMBINATION: synthetic.ncnn+shufflenetv2" in text

    # ---- two passes per chunk: benchmark conf (CLI default 0.25) then evaluation conf (0.001), same iou / min_area defaults
    assert [c[0] for c in calls] == [2, 2, 2, 2]
    assert [c[1] for c in calls] == [0.25, 0.001, 0.25, 0.001]
    assert all(c[2] == 0.45 and c[3] == 50 for c in calls)
    # ---- the pipeline was built with one slot per anchor: the evaluation pass keeps every survivor
    assert made[0].engine.cfg.max_det == 8400
    n_eval = calls[1][4] + calls[3][4]
    assert max(n_eval) > 300, f"evaluation pass survivors per image {n_eval}: the scenario must exceed the old cap"
    # ---- PipelineMetrics as the reference fills them
    for c in calls:
        for m in c[5]:
            assert m.t_total > 0 and m.fps > 0 and m.t_detection > 0
            assert m.memory_mb > 0 and m.cpu_percent >= 0.0
            assert m.num_detections >= 0 and 0.0 <= m.det_confidence_avg <= 1.0
    # ---- CSV: schema, one row, appended on the second run
    summary = out / "comparison_summary.csv"
    df = pd.read_csv(summary)
    assert list(df.columns) == CSV_COLUMNS and len(df) == 1
    assert df.loc[0, "model_combination"] == "synthetic.ncnn+shufflenetv2" and df.loc[0, "num_test_images"] == 4
    assert df.loc[0, "fps"] > 0 and 0.0 <= df.loc[0, "mAP50"] <= 1.0
    assert (out / "synthetic.ncnn+shufflenetv2").is_dir()
    # equal thresholds: ONE pass per chunk (e2e.py:983-984), and the row is appended
    calls.clear()
    # (+ e2e_optimize.py's extras: --warmup N = N passes at conf 0.5 on a random frame before the loop, --no_jit accepted and ignored)
    assert e2e.main(argv + ["--yolo_conf", "0.25", "--num_samples", "2", "--save_viz", "1", "--warmup", "2", "--no_jit"]) == 0
    assert [(c[0], c[1]) for c in calls] == [(1, 0.5), (1, 0.5), (2, 0.25)]
    df = pd.read_csv(summary)
    assert len(df) == 2 and df.loc[1, "num_test_images"] == 2
    # ---- --save_viz: one overlay per processed image (e2e.py:1003-1009), readable, the image's size, and drawn on
    #      (predictions in green: the reference's (0, 255, 0) rectangles)
    from PIL import Image
    viz = sorted((out / "synthetic.ncnn+shufflenetv2" / "visualizations").glob("vis_*.png"))
    assert len(viz) == 2
    for vp in viz:
        im = np.asarray(Image.open(vp).convert("RGB"))
        src = next(f for f in img_dir.iterdir() if f.stem == vp.stem[4:])
        assert im.shape[:2] == np.asarray(Image.open(src)).shape[:2]   # the overlay is the image itself, drawn on
        assert (im[6:34, 6:min(399, im.shape[1] - 1)] == 0).mean() > 0.5   # the summary bar
    count = lambda rgb: sum(int(((np.asarray(Image.open(vp).convert("RGB")) == rgb).all(-1)).sum()) for vp in viz)   # noqa: E731
    assert count((0, 0, 255)) > 100                                      # ground truths in blue (the reference's BGR (255, 0, 0))
    if sum(calls[-1][4]) > 0:   # (calls[0], calls[1] are the warm-up passes)
        assert count((0, 255, 0)) > 100                                  # predictions in green
    # ---- t_roi_extract is measured (the ROI resize launch), not a constant 0 (e2e.py:475)
    assert any(m.t_roi_extract > 0 for c in calls for m in c[5])

    # ---- nothing dropped: the evaluation pass returns exactly what the oracle's postprocess + ROI filter keep on the
    #      device's own out0 (640x640 images: identity letterbox)
    args = e2e.build_parser().parse_args(argv + ["--batch_images", "1"])
    monkeypatch.setattr(backend.HybridPipeline, "run_batch", real_run_batch)
    r = e2e.run_evaluation(args)
    assert r["max_det"] == 8400 and r["files"] == [f"im{i}.png" for i in range(4)]
    eng = backend.Engine(precision="fp16", max_batch=1)
    eng.load_detector(p, b)
    try:
        for i in range(3):
            out0 = eng.detect_raw(imgs[i][None])[0]
            eb, es, ec = P.postprocess(out0, (640, 640), 1.0, (0.0, 0.0), 0.001, 0.45)
            rects, valid = P.roi_rects(eb, 640, 640, 50)
            preds = r["all_preds"][i]
            assert len(preds) == len(valid), f"image {i}: {len(preds)} predictions vs {len(valid)} oracle survivors"
            assert len(valid) > 300
            for pr, k in zip(preds, valid):
                assert pr["bbox"] == tuple(eb[k].astype(int)) and pr["conf"] == float(es[k]) and pr["cls_class"] >= 0
    finally:
        eng.close()
    assert len(r["all_gts"][0]) == 3 and len(r["all_gts"][0][0]) == 5
