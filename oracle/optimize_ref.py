"""ORACLE (test infrastructure, never imported by the product path): CPU restatement of the numerics in which the
reference's second pipeline, ``src/tt100k/pipeline/e2e_optimize.py`` (``HybridPipelineOptimized``), differs from
``e2e.py``.  Detector, NMS and the classifier network are the same code in both files; what changes is

  * the ROI rule (e2e_optimize.py:480-497): ``boxes.astype(np.int32)``, x clipped to [0, w] and y to [0, h] (NOT the
    x1 <= w-1 / x2 >= x1+1 rule of e2e.py:465-470), area filter ``>= min_area``, empty rectangles dropped;
  * the classifier pre-processing (e2e_optimize.py:383-400): ``cv2.cvtColor(BGR2RGB)``, ``cv2.resize(.., (S, S),
    interpolation=cv2.INTER_LINEAR)`` -- no antialiasing, unlike PIL's BILINEAR of e2e.py:366-370 -- then ``/255`` and
    ``(x - 0.18) / 0.34``.

PARITY UNPINNED: cv2 is not installed in the build container and the reference holds no resized fixtures, so
``resize_linear_u8`` (postprocess_ref.py) restates OpenCV's published fixed-point algorithm and is pinned only by its own
properties (identity size, constant images, 2x2 -> 4x4 hand values in tests/test_oracle_cpu.py).
"""
from typing import List, Tuple

import numpy as np

from . import postprocess_ref as P

MEAN, STD = 0.18, 0.34  # e2e_optimize.py:371-372


def roi_rects(boxes: np.ndarray, h: int, w: int, min_area: int = 100) -> Tuple[np.ndarray, List[int]]:
    """e2e_optimize.py:480-497.  Returns (int rects [V,4] of the kept boxes, their indices into ``boxes``)."""
    if len(boxes) == 0:
        return np.zeros((0, 4), np.int32), []
    bi = np.asarray(boxes).astype(np.int32)
    bi[:, [0, 2]] = np.clip(bi[:, [0, 2]], 0, w)
    bi[:, [1, 3]] = np.clip(bi[:, [1, 3]], 0, h)
    areas = (bi[:, 2] - bi[:, 0]) * (bi[:, 3] - bi[:, 1])
    keep = [int(i) for i in np.nonzero(areas >= min_area)[0] if bi[i, 2] > bi[i, 0] and bi[i, 3] > bi[i, 1]]
    return bi[keep], keep


def preprocess_rois(rois_bgr: List[np.ndarray], size: int = 64) -> np.ndarray:
    """e2e_optimize.py:383-400: uint8 RGB crops after the cv2-linear resize, [R, size, size, 3]."""
    out = np.empty((len(rois_bgr), size, size, 3), np.uint8)
    for i, roi in enumerate(rois_bgr):
        rgb = np.ascontiguousarray(roi[:, :, ::-1])
        out[i] = rgb if rgb.shape[:2] == (size, size) else P.resize_linear_u8(rgb, size, size)
    return out


def normalize(rgb_u8: np.ndarray) -> np.ndarray:
    """[R,S,S,3] uint8 -> fp32 [R,3,S,S]: /255 then (x - mean) / std."""
    x = rgb_u8.astype(np.float32).transpose(0, 3, 1, 2) / np.float32(255.0)
    return (x - np.float32(MEAN)) / np.float32(STD)
