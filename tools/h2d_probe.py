#!/usr/bin/env python3
"""Pinned-host -> device copy rate of one bench batch (78.6 MB) on this box: one stream, then split over 2 / 4 streams
(diagnostic for bench.py's h2d_inclusive: what the link gives with no kernels beside it)."""
import time
import torch

dev = torch.device("cuda", 0)
n = 64 * 640 * 640 * 3
host = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(2)]
dst = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(2)]
for parts in (1, 2, 4):
    streams = [torch.cuda.Stream(device=dev) for _ in range(parts)]
    chunk = n // parts

    def go(k):
        for p, s in enumerate(streams):
            with torch.cuda.stream(s):
                dst[k % 2][p * chunk:(p + 1) * chunk].copy_(host[k % 2][p * chunk:(p + 1) * chunk], non_blocking=True)

    for k in range(4):
        go(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 40
    for k in range(K):
        go(k)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{parts} stream(s): {n * K / dt / 1e9:6.1f} GB/s  ({dt / K * 1e3:.3f} ms per 78.6 MB batch)")
