#!/usr/bin/env python3
"""Where the drop-in call's time goes (GPU box): HybridPipeline.run_batch on 64 host images, wall time per call split into the
C-ABI call (upload + pipeline + download) and the Python result building, for a few upload-thread counts."""
import os, sys, time, tempfile, io, contextlib
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "yolo-litepi_amd"))
import torch
from litepi import HybridPipeline, ncnn_export
from litepi.backend import random_shufflenet_state

d = tempfile.mkdtemp()
p, b = os.path.join(d, "m.param"), os.path.join(d, "m.bin")
ncnn_export.export_detector(p, b, "v1", seed=1234, cls_bias=0.0)
from litepi import Engine
_e = Engine(precision="fp16", max_batch=8)
_e.load_detector(p, b)
_cal = np.stack([np.random.default_rng(i).integers(0, 256, (640, 640, 3), dtype=np.uint8) for i in range(8)])
_s = np.sort(_e.detect_raw(_cal)[:, 4].astype(np.float64).ravel())[::-1]
_e.close()
_k = min(max(_s[64], 1e-6), 1 - 1e-6)
ncnn_export.shift_cls_bias(p, b, float(np.log(0.25 / 0.75) - np.log(_k / (1 - _k))))   # ~8 candidates per image, as bench.py
cls = os.path.join(d, "cls.pth")
torch.save({k: torch.from_numpy(np.asarray(v)) for k, v in random_shufflenet_state(91, seed=0).items()}, cls)
imgs = [np.random.default_rng(i).integers(0, 256, (640, 640, 3), dtype=np.uint8) for i in range(64)]
for threads in (os.environ.get("THREADS", "8,16,4,0").split(",")):
    os.environ["LITEPI_UPLOAD_THREADS"] = threads   # (LITEPI_DROPIN_LANES=<n> from the environment: the upload lanes of HybridPipeline)
    with contextlib.redirect_stdout(io.StringIO()):
        pipe = HybridPipeline(p, b, cls, "shufflenetv2", num_classes=91, precision="fp16", max_batch=64, max_det=300)
    try:
        for _ in range(3):
            pipe.run_batch(imgs, 0.25, 0.45, 50)
        n = 12
        t0 = time.perf_counter()
        for _ in range(n):
            out = pipe.run_batch(imgs, 0.25, 0.45, 50)
        t_all = (time.perf_counter() - t0) / n
        t0 = time.perf_counter()
        for _ in range(n):
            pipe.engine.run_batch(imgs, 0.25, 0.45, 50)
        t_eng = (time.perf_counter() - t0) / n
        dimg = torch.from_numpy(np.stack(imgs)).cuda()
        res = torch.zeros(64 * 300 * 32 + 3 * 64 * 4, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            pipe.engine.run_batch_device(dimg.data_ptr(), 64, 640, 640, 0.25, 0.45, 50, res.data_ptr(), res[64 * 300 * 32:].data_ptr())
            pipe.engine.synchronize()
        t_dev = (time.perf_counter() - t0) / n
        print(f"lanes {len(pipe._lanes)}, upload threads {threads:>2s}: run_batch {t_all * 1e3:6.3f} ms ({64 / t_all:7.0f} img/s) = C-ABI call {t_eng * 1e3:6.3f} ms "
              f"(device-resident pipeline alone {t_dev * 1e3:5.3f} ms) + Python {1e3 * (t_all - t_eng):5.3f} ms; {sum(len(r[0]) for r in out)} results", flush=True)
    finally:
        pipe.close()
    # the static thread count is read once per process: re-exec is not allowed on the GPU box, so only the first value counts
    break
