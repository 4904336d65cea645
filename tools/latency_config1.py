#!/usr/bin/env python3
"""BASELINE.json configs[1]: detector only, batch 1, 640x640 -- latency of one NCNNDetector.detect() call (host image in,
boxes out: upload + letterbox + forward + decode + NMS + download), median / p95 of 1000 calls after 50 warm-ups.
Usage (GPU box): python tools/latency_config1.py"""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "yolo-litepi_amd"))
from litepi import NCNNDetector, ncnn_export  # noqa: E402

d = tempfile.mkdtemp()
p, b = os.path.join(d, "m.param"), os.path.join(d, "m.bin")
ncnn_export.export_detector(p, b, "v1", seed=1234, cls_bias=0.0)
img = np.random.default_rng(0).integers(0, 256, (640, 640, 3), dtype=np.uint8)
# class bias calibrated as in bench.py: ~8 candidates pass conf 0.25 (TT100K averages 2.8 boxes per image); an un-calibrated
# synthetic detector saturates max_det and the call then measures the NMS of thousands of candidates (0.21 ms), not the detector
from litepi import Engine  # noqa: E402
_e = Engine(precision="fp16", max_batch=1)
_e.load_detector(p, b)
_s = np.sort(_e.detect_raw(img[None])[0, 4].astype(np.float64))[::-1]
_e.close()
_k = min(max(_s[8], 1e-6), 1 - 1e-6)
ncnn_export.shift_cls_bias(p, b, float(np.log(0.25 / 0.75) - np.log(_k / (1 - _k))))
for prec in ("fp32", "fp16"):
    det = NCNNDetector(p, b, precision=prec, max_batch=1)
    for _ in range(50):
        det.detect(img, 0.25, 0.45)
    t = []
    for _ in range(1000):
        t0 = time.perf_counter()
        boxes, scores, cls = det.detect(img, 0.25, 0.45)
        t.append(time.perf_counter() - t0)
    t = np.array(t) * 1e3
    print(f"config1 {prec}: median {np.median(t):.3f} ms  p95 {np.percentile(t, 95):.3f} ms  ({len(boxes)} boxes)")
    det.engine.close()
