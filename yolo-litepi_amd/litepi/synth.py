"""Synthetic inputs of BASELINE.json's configurations (no dataset exists here): shared by bench.py and the GPU tests so that
both run the same recipe."""
from __future__ import annotations

import numpy as np


def config4_images(n: int, seed: int = 2, size: int = 2048, grain: int = 0) -> np.ndarray:
    """configs[4] recipe: TT100K-shape `size` x `size` BGR uint8 frames -- low-frequency noise (a 64 x 64
    random image blown up block-wise) with six pasted discs of 20-40 px radius in random colours, so that a detector has
    something to find after the 3.2x letterbox down-scale.  `default_rng(seed)`, images drawn in order.
    grain > 0 (bench.py --config 4): per-pixel noise of +-grain grey levels on top, from a second generator so that the base
    images stay the same.  The bare recipe is piecewise constant: thousands of anchors of a random-weight detector then share
    one score bit for bit, and no class-bias calibration can place ~8 detections per frame (the count jumps from 0 to max_det)."""
    rng = np.random.default_rng(seed)
    rep = size // 64
    out = np.empty((n, size, size, 3), np.uint8)
    for i in range(n):
        low = rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)
        img = np.repeat(np.repeat(low, rep, axis=0), rep, axis=1)
        if img.shape[0] != size:   # sizes that 64 does not divide
            img = np.pad(img, ((0, size - img.shape[0]), (0, size - img.shape[1]), (0, 0)), mode="edge")
        for _d in range(6):
            cy, cx, rad = int(rng.integers(100, size - 100)), int(rng.integers(100, size - 100)), int(rng.integers(20, 41))
            colour = rng.integers(0, 256, 3, dtype=np.uint8)
            yy, xx = np.mgrid[cy - rad:cy + rad + 1, cx - rad:cx + rad + 1]
            win = img[cy - rad:cy + rad + 1, cx - rad:cx + rad + 1]
            win[(yy - cy) ** 2 + (xx - cx) ** 2 <= rad * rad] = colour
        out[i] = img
    if grain > 0:
        g2 = np.random.default_rng(seed + 7919)
        for i in range(n):
            noise = g2.integers(-grain, grain + 1, out[i].shape, dtype=np.int16)
            out[i] = np.clip(out[i].astype(np.int16) + noise, 0, 255).astype(np.uint8)
    return out
