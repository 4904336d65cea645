#!/usr/bin/env python3
"""Per-kernel cost of the four --clf_arch classifiers at 64 ROIs of 64x64 (GPU box): one profiled lp_classify per architecture,
seeded random weights (the product's own random_*_state initialisers), fp16.  Prints launches, eager time and the three
largest kernel families per architecture; profiles/r04_cls_archs.txt is this script's output."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "yolo-litepi_amd"))
from litepi import Engine, backend  # noqa: E402

R, NCLS = 64, 91
rois = [np.random.default_rng(i).integers(0, 256, (int(40 + 3 * (i % 9)), int(36 + 5 * (i % 7)), 3), dtype=np.uint8) for i in range(R)]
states = {"shufflenetv2": backend.random_shufflenet_state, "resnet18": backend.random_resnet18_state,
          "mobilenetv2": backend.random_mobilenetv2_state, "efficientnet": backend.random_efficientnet_state}
for arch, make in states.items():
    e = Engine(precision="fp16", max_batch=1, max_det=R, num_classes=NCLS, cls_arch=arch, max_rois=R)
    try:
        e.load_classifier(make(NCLS, seed=0))
        e.classify(rois)                       # warm
        runs = []
        for _ in range(3):
            e.profile_next(True)
            e.classify(rois)
            runs.append(e.profile_read())
        ks = runs[-1]
        tot = np.median([sum(k["ms"] for k in r) for r in runs])
        fam = {}
        for k in ks:
            f = fam.setdefault(k["name"], [0.0, 0, 0.0])
            f[0] += k["ms"]; f[1] += 1; f[2] += k["flops"]
        top = sorted(fam.items(), key=lambda kv: -kv[1][0])[:3]
        gfl = sum(k["flops"] for k in ks) / 1e9
        print(f"{arch:13s} {len(ks):3d} launches  {tot * 1000:7.1f} us eager for {R} ROIs ({tot * 1000 / R:5.2f} us/ROI)  {gfl:6.2f} GFLOP "
              f"= {gfl / tot:7.1f} TFLOP/s;  largest: " + ", ".join(f"{n} x{v[1]} {v[0] * 1000:.1f} us" for n, v in top), flush=True)
    finally:
        e.close()
