#!/usr/bin/env python3
"""Phase stamps of the fused C2f bottleneck kernel (diagnostic; GPU box): LITEPI_BNECK_STAMPS=<file> makes every bottleneck
launch dump 16 clock stamps per workgroup; this runs a warm batch-64 detect and prints per-phase cycle statistics per launch."""
import os, sys, tempfile
import numpy as np
path = os.path.join(tempfile.mkdtemp(), "stamps.bin")
os.environ["LITEPI_BNECK_STAMPS"] = path
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "yolo-litepi_amd"))
from litepi import Engine, ncnn_export  # noqa: E402
d = tempfile.mkdtemp()
p, b = os.path.join(d, "m.param"), os.path.join(d, "m.bin")
ncnn_export.export_detector(p, b, sys.argv[1] if len(sys.argv) > 1 else "v1", seed=1234, cls_bias=-4.0)
B = 64
imgs = np.random.default_rng(0).integers(0, 256, (B, 640, 640, 3), dtype=np.uint8)
e = Engine(precision="fp16", max_batch=B)
e.load_detector(p, b)
e.detect_raw(imgs)
open(path, "wb").close()          # keep only the second (warm) call
e.detect_raw(imgs)
e.close()
raw = np.fromfile(path, dtype=np.uint64)
names = ["start", "", "staging issued", "wait + barrier", "conv_a K loop", "mid epilogue", "conv_b K loop", "epilogue (+cv2)"]
off = 0
while off < len(raw):
    assert raw[off] == 0x424e4543
    grid, H, N, C, T2, TH, TW = (int(v) for v in raw[off + 1: off + 8])
    s = raw[off + 8: off + 8 + grid * 16].reshape(grid, 16).astype(np.int64)
    off += 8 + grid * 16
    wall = (s[:, 15] - s[:, 0])            # 100 MHz ticks
    print(f"map {H}x{H} C={C} cv2 tiles={T2} tile {TH}x{TW}: {grid} workgroups; per-WG wall {np.median(wall) / 100:.2f} us median, "
          f"kernel span {(s[:, 15].max() - s[:, 0].min()) / 100:.1f} us")
    prev = 1
    for k in range(2, 8):
        dt = s[:, k] - s[:, prev]
        print(f"   {names[k]:16s} median {np.median(dt):9.0f} cyc   p90 {np.percentile(dt, 90):9.0f}")
        prev = k
