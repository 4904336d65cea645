import os, sys
sys.path.insert(0, "yolo-litepi_amd")
import numpy as np, torch
from litepi import Engine, ncnn_export
p, b = "/tmp/m.param", "/tmp/m.bin"
ncnn_export.export_detector(p, b, sys.argv[1] if len(sys.argv) > 1 else "v1", seed=1, cls_bias=-6.0)
e = Engine(precision="fp16", max_batch=64, max_det=300, num_classes=91)
e.load_detector(p, b)
imgs = np.random.default_rng(0).integers(0, 256, (64, 640, 640, 3), dtype=np.uint8)
for _ in range(3):
    e.detect_raw(imgs)
e.close()
