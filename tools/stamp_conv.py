#!/usr/bin/env python3
"""Diagnostic: per-workgroup phase timing of one conv layer (GPU box).  Usage:
   python tools/stamp_conv.py N Cin Cout H W [k stride]   -> prints phase medians in shader-clock cycles"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "yolo-litepi_amd"))
from litepi import Engine
N, Cin, Cout, H, W = [int(v) for v in sys.argv[1:6]]
k = int(sys.argv[6]) if len(sys.argv) > 6 else 3
stride = int(sys.argv[7]) if len(sys.argv) > 7 else 1
path = "/tmp/stamps.bin"
os.environ["LITEPI_STAMPS"] = path
e = Engine(precision="fp16", max_batch=N)
rng = np.random.default_rng(0)
x = rng.standard_normal((N, Cin, H, W), dtype=np.float32)
w = (rng.standard_normal((Cout, Cin, k, k), dtype=np.float32) / np.sqrt(Cin * k * k)).astype(np.float32)
e.test_conv(x, w, np.zeros(Cout, np.float32), stride=stride, act=1)
s = np.fromfile(path, dtype=np.uint64).reshape(-1, 16).astype(np.int64)
s = s[s[:, 0] > 0]
print("workgroups", len(s))
wall0, wall1 = s[:, 0], s[:, 12]
print("kernel span (wall clock ticks @100MHz): %.1f us" % ((wall1.max() - wall0.min()) / 100.0))
print("workgroup start spread: p50 %.1f us p90 %.1f us max %.1f us" % tuple(np.percentile(wall0 - wall0.min(), [50, 90, 100]) / 100.0))
print("workgroup lifetime: p50 %.1f us p90 %.1f us" % tuple(np.percentile(wall1 - wall0, [50, 90]) / 100.0))
names = ["start", "c0 stage issue+store", "c0 barrier", "c0 K-loop", "c1 wait barrier", "c1 stage", "c1 barrier", "c1 K-loop", "-", "epilogue"]
pairs = [(1, 2), (2, 3), (3, 4), (4, 5), (5, 6), (6, 7), (7, 8), (8, 9), (9, 10), (10, 11)]
for (a, b), nm in zip(pairs, names):
    if (s[:, b] > 0).all() and (s[:, a] > 0).all():
        d = s[:, b] - s[:, a]
        print(f"{nm:24s} median {np.median(d):9.0f} cyc  p90 {np.percentile(d, 90):9.0f}")
tot = s[:, 11] - s[:, 1]
print("stamped lifetime cycles median", np.median(tot), "p90", np.percentile(tot, 90))
