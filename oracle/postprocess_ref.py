"""Oracle: NumPy restatement of the reference's own pre/post-processing arithmetic.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Citations are to
``src/tt100k/pipeline/e2e.py`` of the reference.  Pinned by
``tests/golden/ref_*.npz``, which ``tools/make_goldens.py`` produced by running
the reference's own ``nms_numpy`` / ``NCNNDetector.postprocess`` /
``HybridPipeline.run`` ROI logic in the build container.

Tie rule (SURVEY §7 "NMS determinism"): the reference sorts with
``scores.argsort()[::-1]`` (e2e.py:96), which leaves the order of equal scores
to the sort implementation.  The oracle and the HIP path both define it as
"descending score, equal scores -> higher candidate index first" (what a stable
ascending sort, reversed, yields); goldens taken from the reference are
tie-free.
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np


# --------------------------------------------------------------------------- a1
def letterbox_params(h: int, w: int, new_shape: int = 640):
    """Geometry of ``letterbox`` (e2e.py:66-86): ratio, unpadded size, float
    half-pads and the integer border split."""
    r = min(new_shape / h, new_shape / w)
    new_unpad = (int(round(w * r)), int(round(h * r)))  # (w, h)
    dw = (new_shape - new_unpad[0]) / 2
    dh = (new_shape - new_unpad[1]) / 2
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    return r, new_unpad, (dw, dh), (top, bottom, left, right)


def resize_linear_u8(img: np.ndarray, dst_w: int, dst_h: int) -> np.ndarray:
    """``cv2.resize(img, (dst_w,dst_h), interpolation=cv2.INTER_LINEAR)`` for
    uint8 HxWxC, restated from OpenCV's published fixed-point algorithm
    (half-pixel centres, 11-bit coefficients, two-stage rounding).

    PARITY UNPINNED: cv2 is not installed in the build container and the
    reference holds no resized fixtures; only the identity and pad-only cases
    of ``letterbox`` are pinned.
    """
    sh, sw = img.shape[:2]
    ONE = 1 << 11

    def coeffs(dst, src):
        scale = 1.0 / (dst / src)  # OpenCV: inv_scale = dsize/ssize; scale = 1./inv_scale
        idx = np.empty(dst, np.int64)
        a = np.empty((dst, 2), np.int64)
        for d in range(dst):
            fx = np.float32((d + 0.5) * scale - 0.5)
            sx = int(np.floor(fx))
            fx = np.float32(fx - sx)
            if sx < 0:
                fx, sx = np.float32(0), 0
            if sx >= src - 1:
                fx, sx = np.float32(0), src - 1
            idx[d] = sx
            # saturate_cast<short>(float * 2048) rounds to nearest even
            a1 = int(np.rint(np.float32(fx) * np.float32(ONE)))
            a0 = int(np.rint((np.float32(1.0) - np.float32(fx)) * np.float32(ONE)))
            a[d] = (a0, a1)
        return idx, a

    xi, xa = coeffs(dst_w, sw)
    yi, ya = coeffs(dst_h, sh)
    src = img.astype(np.int64)
    x1 = np.minimum(xi + 1, sw - 1)
    # horizontal pass: int rows scaled by 2^11
    hrow = src[:, xi, :] * xa[:, 0][None, :, None] + src[:, x1, :] * xa[:, 1][None, :, None]
    y1 = np.minimum(yi + 1, sh - 1)
    s0 = hrow[yi]
    s1 = hrow[y1]
    b0 = ya[:, 0][:, None, None]
    b1 = ya[:, 1][:, None, None]
    out = (((b0 * (s0 >> 4)) >> 16) + ((b1 * (s1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def letterbox(img: np.ndarray, new_shape: int = 640, color: int = 114):
    """e2e.py:66-86 on uint8 HxWx3; returns (img640, ratio, (dw, dh))."""
    h, w = img.shape[:2]
    r, new_unpad, (dw, dh), (top, bottom, left, right) = letterbox_params(h, w, new_shape)
    if (w, h) != new_unpad:
        img = resize_linear_u8(img, new_unpad[0], new_unpad[1])
    out = np.full((img.shape[0] + top + bottom, img.shape[1] + left + right, img.shape[2]),
                  color, np.uint8)
    out[top:top + img.shape[0], left:left + img.shape[1]] = img
    return out, r, (dw, dh)


def preprocess(img_bgr: np.ndarray, new_shape: int = 640):
    """e2e.py:222-238: letterbox, BGR->RGB, x * (1/255) as fp32, HWC->CHW, batch dim."""
    lb, r, pad = letterbox(img_bgr, new_shape)
    rgb = lb[:, :, ::-1].astype(np.float32) * np.float32(1 / 255.0)
    return np.ascontiguousarray(rgb.transpose(2, 0, 1))[None], r, pad


# --------------------------------------------------------------------------- a5
def _order_desc(scores: np.ndarray) -> np.ndarray:
    """Descending score, ties -> higher index first (stable ascending, reversed)."""
    return np.argsort(scores, kind="stable")[::-1]


def nms(boxes: np.ndarray, scores: np.ndarray, iou_threshold: float = 0.45) -> List[int]:
    """e2e.py:89-119, same fp32 arithmetic, deterministic tie rule."""
    if len(boxes) == 0:
        return []
    iou_threshold = float(iou_threshold)
    x1, y1, x2, y2 = boxes.T
    areas = (x2 - x1) * (y2 - y1)
    order = _order_desc(scores)
    keep = []
    while order.size > 0:
        i = order[0]
        keep.append(int(i))
        if len(order) == 1:
            break
        rest = order[1:]
        xx1 = np.maximum(x1[i], x1[rest])
        yy1 = np.maximum(y1[i], y1[rest])
        xx2 = np.minimum(x2[i], x2[rest])
        yy2 = np.minimum(y2[i], y2[rest])
        w = np.maximum(0.0, xx2 - xx1)
        h = np.maximum(0.0, yy2 - yy1)
        inter = w * h
        iou = inter / (areas[i] + areas[rest] - inter + 1e-6)
        order = rest[iou <= iou_threshold]
    return keep


# --------------------------------------------------------------------------- a4
def postprocess(out0: np.ndarray, orig_shape: Tuple[int, int], ratio: float,
                pad: Tuple[float, float], conf_threshold: float = 0.5,
                iou_threshold: float = 0.45):
    """e2e.py:240-296 on one image's ``[4+nc, A]`` fp32 tensor."""
    # the reference receives Python floats here (weak scalars: the arithmetic
    # below stays float32); coerce so NumPy scalars from fixtures behave alike
    ratio, pad = float(ratio), (float(pad[0]), float(pad[1]))
    conf_threshold, iou_threshold = float(conf_threshold), float(iou_threshold)
    pred = np.asarray(out0)
    if pred.ndim == 3:
        pred = pred[0]
    boxes = pred[:4].T
    scores = pred[4:].T
    class_scores = scores.max(axis=1)
    class_ids = scores.argmax(axis=1)
    mask = class_scores > conf_threshold
    boxes, scores, class_ids = boxes[mask], class_scores[mask], class_ids[mask]
    if len(boxes) == 0:
        return np.empty((0, 4)), np.empty((0,)), np.empty((0,))
    xc, yc, bw, bh = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    xyxy = np.stack([xc - bw / 2, yc - bh / 2, xc + bw / 2, yc + bh / 2], axis=1)
    xyxy[:, [0, 2]] -= pad[0]
    xyxy[:, [1, 3]] -= pad[1]
    xyxy /= ratio
    xyxy[:, [0, 2]] = np.clip(xyxy[:, [0, 2]], 0, orig_shape[1])
    xyxy[:, [1, 3]] = np.clip(xyxy[:, [1, 3]], 0, orig_shape[0])
    idx: List[int] = []
    for c in np.unique(class_ids):
        m = class_ids == c
        k = nms(xyxy[m], scores[m], iou_threshold)
        idx.extend(np.where(m)[0][k])
    if not idx:
        return np.empty((0, 4)), np.empty((0,)), np.empty((0,))
    idx = np.array(idx)
    return xyxy[idx], scores[idx], class_ids[idx]


# --------------------------------------------------------------------------- a6
def roi_rects(boxes: np.ndarray, h: int, w: int, min_area: int = 100):
    """e2e.py:465-473: int truncation, clip, area filter.  Returns
    (rects int[N,4] x1,y1,x2,y2 for the valid ones, valid_indices)."""
    rects, valid = [], []
    for idx, box in enumerate(boxes):
        x1, y1, x2, y2 = box.astype(int)
        x1, y1 = np.clip(x1, 0, w - 1), np.clip(y1, 0, h - 1)
        x2, y2 = np.clip(x2, x1 + 1, w), np.clip(y2, y1 + 1, h)
        if (x2 - x1) * (y2 - y1) >= min_area and x2 > x1 and y2 > y1:
            rects.append((int(x1), int(y1), int(x2), int(y2)))
            valid.append(idx)
    return np.array(rects, np.int64).reshape(-1, 4), valid


# ---- helpers of the fp16 comparisons (tests/, __graft_entry__.smoke) -------------------------------------------------
def box_match(b, e, slack_px=4.0):
    return np.abs(np.asarray(b, np.float64) - np.asarray(e, np.float64)).max() <= slack_px + 0.02 * np.abs(np.asarray(e, np.float64)).max()


def stable_boxes(out0, hw, conf, iou, min_area, rng, trials=12, band=0.02):
    """Oracle post-NMS boxes (score > conf + band, area filter passed) whose presence does not hinge on a near-tie: the
    box must survive `trials` re-runs of the oracle's postprocess on out0 perturbed by the documented fp16 error scale
    (scores +-0.015, box centres / sizes +-2 px).  What is left is what an fp16 detector has no excuse to miss; a box
    that an NMS order swap or an IoU within a hair of the threshold can remove is excluded -- that, and the +-band around
    the conf threshold, is the documented exclusion zone of the fp16 comparison."""
    eb, es, _ = postprocess(out0, hw, 1.0, (0.0, 0.0), conf, iou)
    if len(eb) == 0:
        return []
    _, valid = roi_rects(eb, hw[0], hw[1], min_area)
    keep = [i for i in valid if es[i] > conf + band]
    alive = {i: True for i in keep}
    for _ in range(trials):
        o = out0.copy()
        o[4:] = np.clip(o[4:] + rng.uniform(-0.015, 0.015, o[4:].shape).astype(np.float32), 0, 1)
        o[:4] += rng.uniform(-2.0, 2.0, o[:4].shape).astype(np.float32)
        pb, _, _ = postprocess(o, hw, 1.0, (0.0, 0.0), conf, iou)
        for i in keep:
            if alive[i] and not any(box_match(q, eb[i], 6.0) for q in pb):
                alive[i] = False
    return [(eb[i], float(es[i])) for i in keep if alive[i]]
