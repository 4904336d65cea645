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
ncnn_export.export_detector(p, b, "v1", seed=1234, cls_bias=-4.0)
img = np.random.default_rng(0).integers(0, 256, (640, 640, 3), dtype=np.uint8)
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
