"""Oracle: NumPy restatement of Pillow's ``Image.resize(size, BILINEAR)`` for
8-bit RGB, i.e. what ``transforms.Resize((64,64))`` does to each ROI in the
reference classifier (``src/tt100k/pipeline/e2e.py:366-370,385-389``).

TEST INFRASTRUCTURE ONLY.  Pillow's resampler (libImaging ``Resample.c``) is a
separable convolution: triangle filter whose support grows with the down-scale
factor (antialiasing), coefficients normalised in double then quantised to
22-bit fixed point, horizontal pass first into a uint8 intermediate, then the
vertical pass; each pass rounds to uint8.  Pinned against Pillow itself
(installed in the build container) on the reference's ``debug_rois`` crops and
on random sizes: ``tests/test_oracle_pil.py``; expected outputs for the GPU box
are committed as ``tests/golden/pil_resize.npz``.
"""
from __future__ import annotations

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def precompute_coeffs(in_size: int, out_size: int):
    """Bounds + fixed-point coefficients of one axis (``precompute_coeffs`` +
    ``normalize_coeffs_8bpc`` of Pillow's Resample.c), box = (0, in_size)."""
    scale = in_size / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = 1.0 * filterscale  # bilinear filter support = 1.0
    ksize = int(np.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int64)
    kk = np.zeros((out_size, ksize), np.int64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = np.zeros(ksize, np.float64)
        ww = 0.0
        for x in range(xmax):
            v = (x + xmin - center + 0.5) * ss
            v = -v if v < 0 else v
            wv = 1.0 - v if v < 1.0 else 0.0
            w[x] = wv
            ww += wv
        for x in range(xmax):
            if ww != 0.0:
                w[x] /= ww
        for x in range(ksize):
            if w[x] < 0:
                kk[xx, x] = int(-0.5 + w[x] * (1 << PRECISION_BITS))
            else:
                kk[xx, x] = int(0.5 + w[x] * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _clip8(v: np.ndarray) -> np.ndarray:
    return np.clip(v >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resize_bilinear_u8(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """``Image.fromarray(img).resize((out_w, out_h), Image.BILINEAR)`` for uint8 HxWxC."""
    in_h, in_w = img.shape[:2]
    src = img.astype(np.int64)
    half = 1 << (PRECISION_BITS - 1)
    if out_w != in_w:
        bx, kx = precompute_coeffs(in_w, out_w)
        tmp = np.empty((in_h, out_w, img.shape[2]), np.uint8)
        for xx in range(out_w):
            xmin, n = bx[xx]
            acc = (src[:, xmin:xmin + n, :] * kx[xx, :n][None, :, None]).sum(axis=1) + half
            tmp[:, xx, :] = _clip8(acc)
    else:
        tmp = img
    if out_h != in_h:
        by, ky = precompute_coeffs(in_h, out_h)
        t = tmp.astype(np.int64)
        out = np.empty((out_h, tmp.shape[1], img.shape[2]), np.uint8)
        for yy in range(out_h):
            ymin, n = by[yy]
            acc = (t[ymin:ymin + n] * ky[yy, :n][:, None, None]).sum(axis=0) + half
            out[yy] = _clip8(acc)
    else:
        out = tmp
    return out


def classifier_input(roi_bgr: np.ndarray, size: int = 64) -> np.ndarray:
    """e2e.py:385-389 for one ROI: BGR->RGB, PIL bilinear to size x size (uint8),
    ToTensor (/255, CHW), Normalize(mean .18, std .34).  Returns fp32 [3,size,size]."""
    rgb = np.ascontiguousarray(roi_bgr[:, :, ::-1])
    r = resize_bilinear_u8(rgb, size, size)
    t = r.astype(np.float32) / np.float32(255.0)
    t = (t - np.float32(0.18)) / np.float32(0.34)
    return np.ascontiguousarray(t.transpose(2, 0, 1))
