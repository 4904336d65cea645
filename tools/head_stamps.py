#!/usr/bin/env python3
"""Phase stamps of the fused Detect-head kernel (diagnostic; GPU box): LITEPI_HEAD_STAMPS=<file> makes every head launch dump
16 clock stamps per workgroup; this runs a warm batch-64 detect and prints per-phase cycle statistics per level."""
import os, sys, tempfile
import numpy as np
path = os.path.join(tempfile.mkdtemp(), "stamps.bin")
os.environ["LITEPI_HEAD_STAMPS"] = path
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "yolo-litepi_amd"))
from litepi import Engine, ncnn_export  # noqa: E402
d = tempfile.mkdtemp()
p, b = os.path.join(d, "m.param"), os.path.join(d, "m.bin")
ncnn_export.export_detector(p, b, sys.argv[1] if len(sys.argv) > 1 else "v1", seed=1234, cls_bias=-4.0)
B = 64
imgs = np.random.default_rng(0).integers(0, 256, (B, 640, 640, 3), dtype=np.uint8)
e = Engine(precision="fp16", max_batch=B)
e.load_detector(p, b)
# "raw": the parity hook (every anchor decoded, out0 written); default: the product path (conf filter, decode only where needed)
raw_path = len(sys.argv) > 2 and sys.argv[2] == "raw"
os.environ["LITEPI_NO_GRAPH"] = "1"
run = (lambda: e.detect_raw(imgs)) if raw_path else (lambda: e.detect(list(imgs), 0.25, 0.45))
run()
open(path, "wb").close()          # keep only the second (warm) call
run()
e.close()
raw = np.fromfile(path, dtype=np.uint64)
names = ["start", "", "issued", "chunk0 landed", "stage A loop", "A epilogue", "B box", "B cls", "wait C", "decode"]
off = 0
while off < len(raw):
    assert raw[off] == 0x48454144
    grid, H, N = int(raw[off + 1]), int(raw[off + 2]), int(raw[off + 3])
    s = raw[off + 4: off + 4 + grid * 16].reshape(grid, 16).astype(np.int64)
    off += 4 + grid * 16
    wall = (s[:, 15] - s[:, 0])            # 100 MHz ticks
    print(f"level {H}x{H}: {grid} workgroups; per-WG wall {np.median(wall) / 100:.2f} us median, kernel span {(s[:, 15].max() - s[:, 0].min()) / 100:.1f} us")
    prev = 1
    for k in range(2, 10):
        dt = s[:, k] - s[:, prev]
        print(f"   {names[k]:16s} median {np.median(dt):9.0f} cyc   p90 {np.percentile(dt, 90):9.0f}")
        prev = k
