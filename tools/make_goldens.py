#!/usr/bin/env python3
"""Generate tests/golden/ref_*.npz by running the REFERENCE's own functions.

Runs only in the build container (needs /root/reference).  The reference module
``src/tt100k/pipeline/e2e.py`` imports cv2 / ncnn / seaborn / torchvision at the
top (e2e.py:19-29), none of which is installed here; those names are stubbed as
empty modules so the file imports, and only functions that never touch them are
called:

  nms_numpy                      e2e.py:89-119
  NCNNDetector.postprocess       e2e.py:240-296  (unbound, dummy self)
  HybridPipeline.run ROI logic   e2e.py:443-531  (fake detector / classifier)
  evaluate_predictions           e2e.py:656-824

Inputs are seeded and tie-free (distinct scores), so the goldens do not depend
on how NumPy's unstable argsort orders equal keys.  The committed .npz files
hold inputs AND the reference's outputs; nothing of the reference's code is
stored.
"""
import importlib.util
import os
import sys
import types

import numpy as np

REF = "/root/reference/src/tt100k/pipeline/e2e.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def import_reference():
    for name in ("cv2", "ncnn", "seaborn", "torchvision", "torchvision.transforms",
                 "torchvision.models", "matplotlib", "matplotlib.pyplot", "psutil", "tqdm"):
        if name in sys.modules:
            continue
        try:
            importlib.import_module(name)
        except Exception:
            m = types.ModuleType(name)
            sys.modules[name] = m
    sys.modules["ncnn"].Mat = type("Mat", (), {})
    tv = sys.modules["torchvision"]
    if not hasattr(tv, "transforms"):
        tv.transforms = sys.modules["torchvision.transforms"]
        tv.models = sys.modules["torchvision.models"]
    if not hasattr(sys.modules["tqdm"], "tqdm"):
        sys.modules["tqdm"].tqdm = lambda x, **k: x
    if not hasattr(sys.modules["psutil"], "cpu_percent"):
        sys.modules["psutil"].cpu_percent = lambda: 0.0
    spec = importlib.util.spec_from_file_location("ref_e2e", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def distinct_scores(rng, n, lo=0.0, hi=1.0):
    s = rng.uniform(lo, hi, size=n).astype(np.float32)
    while len(np.unique(s)) != n:
        s = rng.uniform(lo, hi, size=n).astype(np.float32)
    return s


def random_boxes(rng, n, size=640.0, clusters=8):
    """Clustered boxes so that many pairs overlap around the IoU threshold."""
    centers = rng.uniform(40, size - 40, size=(clusters, 2))
    c = centers[rng.integers(0, clusters, n)] + rng.normal(0, 6, size=(n, 2))
    wh = rng.uniform(12, 90, size=(n, 2))
    b = np.concatenate([c - wh / 2, c + wh / 2], axis=1)
    return np.clip(b, 0, size).astype(np.float32)


def gen_nms(ref, rng):
    cases = {}
    for i, (n, thr) in enumerate([(1, 0.45), (2, 0.45), (17, 0.45), (64, 0.45), (65, 0.5),
                                  (300, 0.45), (1000, 0.3), (2500, 0.45), (8400, 0.45)]):
        boxes = random_boxes(rng, n, clusters=max(2, n // 12))
        scores = distinct_scores(rng, n)
        keep = np.array(ref.nms_numpy(boxes, scores, thr), np.int64)
        cases[f"c{i}_boxes"], cases[f"c{i}_scores"] = boxes, scores
        cases[f"c{i}_thr"] = np.float64(thr)
        cases[f"c{i}_keep"] = keep
    # adversarial: pairs engineered to sit right at the threshold
    base = np.array([100, 100, 200, 200], np.float32)
    boxes = [base]
    for d in np.linspace(35.0, 40.0, 41):
        boxes.append(base + np.array([d, 0, d, 0], np.float32))
    boxes = np.stack(boxes).astype(np.float32)
    scores = np.linspace(0.99, 0.30, len(boxes)).astype(np.float32)
    cases["adv_boxes"], cases["adv_scores"], cases["adv_thr"] = boxes, scores, np.float64(0.45)
    cases["adv_keep"] = np.array(ref.nms_numpy(boxes, scores, 0.45), np.int64)
    # degenerate (zero-area) boxes
    boxes = np.array([[10, 10, 10, 10], [10, 10, 10, 10], [0, 0, 5, 5], [0, 0, 5, 5.0001]], np.float32)
    scores = np.array([0.9, 0.8, 0.7, 0.6], np.float32)
    cases["deg_boxes"], cases["deg_scores"], cases["deg_thr"] = boxes, scores, np.float64(0.45)
    cases["deg_keep"] = np.array(ref.nms_numpy(boxes, scores, 0.45), np.int64)
    np.savez_compressed(os.path.join(OUT, "ref_nms.npz"), **cases)
    print("nms:", {k: v.shape for k, v in cases.items() if k.endswith("keep")})


def synth_out0(rng, n_obj, n_bg_pass, conf_lo):
    """A plausible [5,8400] detector output: n_obj clusters of high-score anchors
    plus n_bg_pass weak anchors; all scores distinct."""
    A = 8400
    out = np.zeros((5, A), np.float32)
    out[0] = rng.uniform(0, 640, A)
    out[1] = rng.uniform(0, 640, A)
    out[2] = rng.uniform(4, 120, A)
    out[3] = rng.uniform(4, 120, A)
    score = distinct_scores(rng, A, 0.0, conf_lo * 0.9)
    idx = rng.permutation(A)
    p = 0
    for _ in range(n_obj):
        k = int(rng.integers(3, 12))
        ids = idx[p:p + k]
        p += k
        cx, cy = rng.uniform(60, 580, 2)
        w, h = rng.uniform(20, 100, 2)
        out[0, ids] = cx + rng.normal(0, 1.5, k)
        out[1, ids] = cy + rng.normal(0, 1.5, k)
        out[2, ids] = w + rng.normal(0, 2.0, k)
        out[3, ids] = h + rng.normal(0, 2.0, k)
        score[ids] = rng.uniform(0.3, 0.98, k).astype(np.float32)
    ids = idx[p:p + n_bg_pass]
    score[ids] = rng.uniform(conf_lo * 1.05, max(0.29, conf_lo * 1.05 + 0.05), n_bg_pass).astype(np.float32)
    while len(np.unique(score)) != A:
        score += rng.uniform(0, 1e-6, A).astype(np.float32)
    out[4] = score
    return out


def gen_postprocess(ref, rng):
    cases = {}
    dummy = types.SimpleNamespace()
    geoms = [((640, 640), 1.0, (0.0, 0.0)),
             ((2048, 2048), 0.3125, (0.0, 0.0)),
             ((681, 1198), min(640 / 681, 640 / 1198), ((640 - round(1198 * 640 / 1198)) / 2,
                                                        (640 - round(681 * 640 / 1198)) / 2)),
             ((480, 640), 1.0, (0.0, 80.0))]
    i = 0
    for j, (conf, n_obj, n_bg) in enumerate([(0.25, 4, 6), (0.25, 0, 0), (0.001, 6, 900), (0.5, 3, 0)]):
        for orig, ratio, pad in (geoms[j % 4], geoms[(j + 2) % 4]):
            out0 = synth_out0(rng, n_obj, n_bg, conf)
            b, s, c = ref.NCNNDetector.postprocess(dummy, out0, orig, ratio, pad, conf, 0.45)
            cases[f"c{i}_out0"] = out0
            cases[f"c{i}_geom"] = np.array([orig[0], orig[1], ratio, pad[0], pad[1], conf, 0.45], np.float64)
            cases[f"c{i}_boxes"] = np.asarray(b)
            cases[f"c{i}_scores"] = np.asarray(s)
            cases[f"c{i}_cls"] = np.asarray(c)
            i += 1
    np.savez_compressed(os.path.join(OUT, "ref_postprocess.npz"), **cases)
    print("postprocess:", i, "cases;", [cases[f"c{j}_boxes"].shape[0] for j in range(i)])


def gen_pipeline(ref, rng):
    """HybridPipeline.run with fake detector/classifier: pins the ROI clip / area
    filter / result-dict assembly (e2e.py:460-531)."""
    cases = {}

    class FakeDet:
        def __init__(self, b, s):
            self.b, self.s = b, s

        def detect(self, image, conf, iou):
            return self.b, self.s, np.zeros(len(self.b), np.int64)

    class FakeCls:
        def __init__(self):
            self.shapes = []

        def predict_batch(self, images):
            self.shapes.extend([im.shape for im in images])
            n = len(images)
            probs = np.full((n, 5), 0.1, np.float32)
            ids = np.array([im.shape[0] % 5 for im in images])
            probs[np.arange(n), ids] = 0.6
            return ids, probs

    for i, (h, w, min_area) in enumerate([(640, 640, 50), (681, 1198, 50), (2048, 2048, 100), (64, 48, 50)]):
        n = 40
        b = np.stack([rng.uniform(-20, w + 20, n), rng.uniform(-20, h + 20, n),
                      rng.uniform(-20, w + 20, n), rng.uniform(-20, h + 20, n)], 1)
        small = rng.random(n) < 0.4
        b[small, 2] = b[small, 0] + rng.uniform(0, 9, small.sum())
        b[small, 3] = b[small, 1] + rng.uniform(0, 9, small.sum())
        b[:, [0, 2]] = np.clip(np.sort(b[:, [0, 2]], 1), 0, w)
        b[:, [1, 3]] = np.clip(np.sort(b[:, [1, 3]], 1), 0, h)
        b = b.astype(np.float32)
        s = distinct_scores(rng, n, 0.26, 0.99)
        pipe = ref.HybridPipeline.__new__(ref.HybridPipeline)
        pipe.detector, pipe.classifier, pipe.batch_size = FakeDet(b, s), FakeCls(), 8
        img = np.zeros((h, w, 3), np.uint8)
        res, met = pipe.run(img, 0.25, 0.45, min_area)
        cases[f"c{i}_hw_minarea"] = np.array([h, w, min_area], np.int64)
        cases[f"c{i}_boxes"], cases[f"c{i}_scores"] = b, s
        cases[f"c{i}_roi_shapes"] = np.array(pipe.classifier.shapes, np.int64).reshape(-1, 3)
        cases[f"c{i}_res_bbox"] = np.array([r["bbox"] for r in res], np.int64).reshape(-1, 4)
        cases[f"c{i}_res_det_conf"] = np.array([r["det_conf"] for r in res], np.float64)
        cases[f"c{i}_res_cls"] = np.array([r["cls_class"] for r in res], np.int64)
        cases[f"c{i}_res_cls_conf"] = np.array([r["cls_conf"] for r in res], np.float64)
        cases[f"c{i}_num_detections"] = np.int64(met.num_detections)
    np.savez_compressed(os.path.join(OUT, "ref_pipeline.npz"), **cases)
    print("pipeline:", [cases[f"c{j}_res_bbox"].shape[0] for j in range(4)])


def gen_evaluate(ref, rng):
    """evaluate_predictions (e2e.py:656-824) on toy prediction / ground-truth sets."""
    cases = {}
    nclass = 6
    for i, (nimg, noise, drop, extra) in enumerate([(12, 2.0, 0.1, 1), (30, 6.0, 0.3, 3), (5, 0.5, 0.0, 0), (8, 12.0, 0.5, 4)]):
        preds_all, gts_all = [], []
        for _ in range(nimg):
            ng = int(rng.integers(0, 5))
            gts, preds = [], []
            for _ in range(ng):
                x1, y1 = rng.integers(0, 500, 2)
                w, h = rng.integers(20, 120, 2)
                c = int(rng.integers(0, nclass - 1))  # class nclass-1 never appears in GT
                gts.append((c, int(x1), int(y1), int(x1 + w), int(y1 + h)))
                if rng.random() >= drop:
                    d = rng.normal(0, noise, 4)
                    pc = c if rng.random() < 0.85 else int(rng.integers(0, nclass))
                    preds.append({"bbox": (int(x1 + d[0]), int(y1 + d[1]), int(x1 + w + d[2]), int(y1 + h + d[3])),
                                  "conf": float(rng.uniform(0.05, 0.99)), "cls_class": pc})
            for _ in range(int(rng.integers(0, extra + 1))):
                x1, y1 = rng.integers(0, 500, 2)
                preds.append({"bbox": (int(x1), int(y1), int(x1 + 40), int(y1 + 40)), "conf": float(rng.uniform(0.01, 0.6)),
                              "cls_class": int(rng.integers(-1, nclass))})
            preds_all.append(preds)
            gts_all.append(gts)
        m = ref.evaluate_predictions(preds_all, gts_all, nclass, 0.45)
        flat_p = [(j, *p["bbox"], p["conf"], p["cls_class"]) for j, ps in enumerate(preds_all) for p in ps]
        flat_g = [(j, *g) for j, gs in enumerate(gts_all) for g in gs]
        cases[f"c{i}_preds"] = np.array(flat_p, np.float64).reshape(-1, 7)
        cases[f"c{i}_gts"] = np.array(flat_g, np.int64).reshape(-1, 6)
        cases[f"c{i}_nimg_nclass"] = np.array([nimg, nclass], np.int64)
        for k in ("precision", "recall", "f1", "tp", "fp", "fn", "ap50_per_class"):
            cases[f"c{i}_{k}"] = np.asarray(m[k], np.float64)
        cases[f"c{i}_map"] = np.array([m["mAP50"], m["mAP50_95"]], np.float64)
        cases[f"c{i}_present"] = np.asarray(m["classes_present"], bool)
    np.savez_compressed(os.path.join(OUT, "ref_evaluate.npz"), **cases)
    print("evaluate:", [tuple(np.round(cases[f"c{j}_map"], 3)) for j in range(4)])


def main():
    os.makedirs(OUT, exist_ok=True)
    ref = import_reference()
    rng = np.random.default_rng(20240917)
    gen_nms(ref, rng)
    gen_postprocess(ref, rng)
    gen_pipeline(ref, rng)
    gen_evaluate(ref, np.random.default_rng(99))


if __name__ == "__main__":
    main()
