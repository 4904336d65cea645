import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "yolo-litepi_amd"))
from litepi import Engine, ncnn_export
from litepi.backend import random_shufflenet_state
d = tempfile.mkdtemp()
p, b = os.path.join(d, "m.param"), os.path.join(d, "m.bin")
ncnn_export.export_detector(p, b, "v1", seed=1234, cls_bias=-2.0)
B = 128
imgs = np.random.default_rng(0).integers(0, 256, (B, 640, 640, 3), dtype=np.uint8)
e = Engine(precision="fp16", max_batch=B, max_det=300, num_classes=91)
e.load_detector(p, b)
e.load_classifier(random_shufflenet_state(91, seed=0))
print("loaded", flush=True)
o = e.detect_raw(imgs)
print("detect_raw ok", o.shape, flush=True)
r = e.run_batch(list(imgs), 0.25, 0.45, 50)
print("run_batch ok", int(np.sum(r[1])), flush=True)
e.close()
