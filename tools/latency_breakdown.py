#!/usr/bin/env python3
"""Where one NCNNDetector.detect() call (BASELINE configs[1]: batch 1, host image in, host boxes out) spends its time:
per-launch HIP-event times of an eager pass, against the wall time of the graph-replayed call."""
import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "yolo-litepi_amd"))
from litepi import NCNNDetector, ncnn_export  # noqa: E402
d = tempfile.mkdtemp()
p, b = os.path.join(d, "m.param"), os.path.join(d, "m.bin")
ncnn_export.export_detector(p, b, "v1", seed=1234, cls_bias=-4.0)
img = np.random.default_rng(0).integers(0, 256, (640, 640, 3), dtype=np.uint8)
det = NCNNDetector(p, b, precision="fp16", max_batch=1)
for _ in range(20):
    det.detect(img, 0.25, 0.45)
t = []
for _ in range(300):
    t0 = time.perf_counter(); det.detect(img, 0.25, 0.45); t.append(time.perf_counter() - t0)
print(f"wall median {np.median(t) * 1e3:.3f} ms")
det.engine.profile_next(True)
det.detect(img, 0.25, 0.45)
prof = det.engine.profile_read()
print(f"eager kernel sum {sum(k['ms'] for k in prof) * 1e3:.1f} us over {len(prof)} launches")
for k in sorted(prof, key=lambda k: -k["ms"])[:12]:
    print(f"  {k['name']:34s} {k['layer'][:30]:30s} {k['ms'] * 1e3:6.1f} us")
det.engine.close()
