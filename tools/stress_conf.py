"""Stress run of the full pipeline at low confidence thresholds (SURVEY section 8d, config 2 stress mode): an un-calibrated
synthetic detector saturates max_det on every image, so NMS sees thousands of candidates and the classifier 19 200 ROIs per
64-image step.  Usage (GPU box): python tools/stress_conf.py"""
import sys, time, os, numpy as np, torch
sys.path.insert(0, os.path.join(os.getcwd(), "yolo-litepi_amd"))
import tempfile
from litepi import Engine, ncnn_export
from litepi.backend import random_shufflenet_state
from litepi.distributed import alloc_result_buffers
d = tempfile.mkdtemp()
p, b = d + "/m.param", d + "/m.bin"
ncnn_export.export_detector(p, b, "v1", seed=1234, cls_bias=-2.0)
B = 64
dev = torch.device("cuda", 0)
imgs = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (B, 640, 640, 3), dtype=np.uint8)).to(dev)
e = Engine(precision="fp16", max_batch=B, max_det=300, num_classes=91)
e.load_detector(p, b); e.load_classifier(random_shufflenet_state(91))
st = torch.cuda.Stream(device=dev); e.set_stream(st.cuda_stream)
dets, counts = alloc_result_buffers(B, 300, dev)
for conf in (0.25, 0.05, 0.001):
    with torch.cuda.stream(st):
        for _ in range(3): e.run_batch_device(imgs.data_ptr(), B, 640, 640, conf, 0.45, 50, dets.data_ptr(), counts.data_ptr())
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): e.run_batch_device(imgs.data_ptr(), B, 640, 640, conf, 0.45, 50, dets.data_ptr(), counts.data_ptr())
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
    print(f"conf {conf}: {dt*1e3:.2f} ms/step, kept {int(counts[:B].sum())} rois, pre-filter {int(counts[B:2 * B].sum())}")

# per-launch profile of one stressed step (HIP events), top entries
with torch.cuda.stream(st):
    e.profile_next(True)
    e.run_batch_device(imgs.data_ptr(), B, 640, 640, 0.25, 0.45, 50, dets.data_ptr(), counts.data_ptr())
    torch.cuda.synchronize()
    prof = sorted(e.profile_read(), key=lambda k: -k["ms"])
    print("profiled step %.2f ms" % sum(k["ms"] for k in prof))
    for k in prof[:14]:
        print("  %-28s %-26s %8.1f us" % (k["name"], k["layer"], k["ms"] * 1e3))
