"""Oracle: CPU restatement of ``HybridPipeline.run`` (reference
``src/tt100k/pipeline/e2e.py:443-531``) on top of the other oracle modules.

TEST INFRASTRUCTURE ONLY.  This is also what ``bench.py`` times as ``cpu_baseline``
(kind "port": torch-CPU/oneDNN convs + NumPy post-processing, i.e. the reference's
ONNX-CPU path restated; it is NOT ONNX Runtime).
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np
import torch

from . import ncnn_ref, postprocess_ref, shufflenet_ref


class CpuPipeline:
    def __init__(self, det_layers, cls_model, input_size: int = 640, cls_input: int = 64, batch_size: int = 8):
        self.layers = det_layers
        self.cls = cls_model
        self.S = input_size
        self.cls_input = cls_input
        self.batch_size = batch_size

    @torch.no_grad()
    def detect_raw(self, image_bgr: np.ndarray) -> Tuple[np.ndarray, float, Tuple[float, float]]:
        x, r, pad = postprocess_ref.preprocess(image_bgr, self.S)
        out0 = ncnn_ref.run_graph(self.layers, torch.from_numpy(x))["out0"].numpy()[0]
        return out0, r, pad

    def detect(self, image_bgr: np.ndarray, conf: float, iou: float):
        out0, r, pad = self.detect_raw(image_bgr)
        return postprocess_ref.postprocess(out0, image_bgr.shape[:2], r, pad, conf, iou)

    def run(self, image_bgr: np.ndarray, conf: float = 0.5, iou: float = 0.45, min_area: int = 100) -> Tuple[List[Dict], int]:
        boxes, scores, det_cls = self.detect(image_bgr, conf, iou)
        num_detections = len(boxes)
        h, w = image_bgr.shape[:2]
        rects, valid = postprocess_ref.roi_rects(boxes, h, w, min_area)
        rois = [image_bgr[y1:y2, x1:x2] for x1, y1, x2, y2 in rects]
        if valid:
            boxes, scores, det_cls = boxes[valid], scores[valid], det_cls[valid]
        else:
            boxes, scores, det_cls = np.empty((0, 4)), np.empty((0,)), np.empty((0,))
        all_cls, all_probs = [], []
        for i in range(0, len(rois), self.batch_size):
            ids, probs = shufflenet_ref.predict_batch(self.cls, rois[i:i + self.batch_size], self.cls_input)
            all_cls.extend(ids)
            all_probs.extend(probs)
        results = []
        for i in range(len(boxes)):
            results.append({
                "bbox": tuple(boxes[i].astype(int)),
                "box_f": boxes[i].astype(np.float32),
                "det_class": int(det_cls[i]),
                "det_conf": float(scores[i]),
                "cls_class": int(all_cls[i]) if i < len(all_cls) else -1,
                "cls_conf": float(np.max(all_probs[i])) if i < len(all_probs) else 0.0,
                "probs": all_probs[i] if i < len(all_probs) else None,
            })
        return results, num_detections
