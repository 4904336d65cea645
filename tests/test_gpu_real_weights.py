"""End-to-end parity on the reference's REAL detector weights and REAL sign crops (``-m gpu``).

Everything else in the suite runs LSUV-scaled random weights on uniform noise.  Here the detector is the reference's
exported YOLO-LitePi v1 (``oracle/_ref/yolo_plus_v1.{param,bin}``) and the inputs are scenes built from the 15 traffic-
sign crops the reference keeps next to its pipeline (``src/vntsr/pipeline/debug_rois``, staged as image DATA under
``oracle/_ref/debug_rois`` by ``__graft_entry__.build()`` when the reference checkout exists; never committed): every
crop pasted at native scale onto a smooth background.  The real model finds real signs in them (scores up to 0.8).

  * fp32: the post-NMS result lists equal the CPU oracle's box for box (count, order, int boxes, scores <= 1e-3,
    classifier arg-max) -- north_star's "identical post-NMS box sets";
  * fp16: NO stability filter.  Every oracle box with score outside +-BAND of conf must be found within BOX_PX, every
    device box with score outside the band must sit on an oracle box; BAND (0.005) / BOX_PX (1) are sized from the fp16
    error measured on these weights (out0 score error ~0.003 on real signs; printed), not from the synthetic-model
    bounds (0.02 / 2 px + 2 %).

Operating points (round 4): besides conf 0.10 the tests run the reference's own two passes (e2e.py:971-992): the benchmark
pass at conf 0.25 and the evaluation pass at conf 0.001, the latter on CLUTTERED scenes (a noise image and scenes with 60-180
rescaled crops pasted at random: 130-350 candidates and up to 240 kept boxes per image on the real score distribution), and a
capacity-64 handle (the benchmarked plan: other tile shapes and the whole-C2f launches) on 64 distinct scenes.

The float forward itself stays "parity unpinned" (the reference holds no expected outputs; NCNN / ONNX Runtime are not
installable here): the checker is oracle/ncnn_ref.py.  The classifier runs seeded synthetic weights (the reference ships
none)."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_REF = os.path.join(_ROOT, "oracle", "_ref")
_REAL = (os.path.join(_REF, "yolo_plus_v1.param"), os.path.join(_REF, "yolo_plus_v1.bin"))
_CROPS = sorted(glob.glob(os.path.join(_REF, "debug_rois", "*")))
_HAVE = all(os.path.exists(p) for p in _REAL) and len(_CROPS) >= 10

CONF, IOU, MIN_AREA = 0.10, 0.45, 50   # the first operating point; the reference's own are 0.25 (--benchmark_conf) and 0.001 (--yolo_conf)
BAND = 0.005      # score tolerance / band around conf inside which fp16 may decide differently (measured error printed below)
BOX_PX = 1.0      # fp16 box tolerance (pixels) for matched boxes


def _scenes(n=16):
    """n BGR scenes 640x640: every crop once per scene at native scale on a 4 x 4 grid with jitter; backgrounds cycle
    through flat grey, a bicubic colour wash and a vertical gradient."""
    from PIL import Image
    crops = [np.asarray(Image.open(f).convert("RGB"))[..., ::-1].copy() for f in _CROPS]
    out = []
    for s in range(n):
        rng = np.random.default_rng(1000 + s)
        mode = s % 3
        if mode == 0:
            bg = np.full((640, 640, 3), 100 + 5 * (s % 7), np.float32)
        elif mode == 1:
            low = rng.integers(90, 160, (4, 4, 3)).astype(np.uint8)
            bg = np.asarray(Image.fromarray(low).resize((640, 640), Image.BICUBIC)).astype(np.float32)
        else:
            g = np.linspace(80, 170, 640, dtype=np.float32)[:, None, None]
            bg = np.broadcast_to(g, (640, 640, 3)).copy()
        img = np.clip(bg, 0, 255).astype(np.uint8)
        order = rng.permutation(len(crops))
        for slot, k in enumerate(order[:16]):
            c = crops[k]
            gx, gy = slot % 4, slot // 4
            x = 30 + gx * 150 + int(rng.integers(0, 40))
            y = 30 + gy * 150 + int(rng.integers(0, 40))
            h, w = c.shape[:2]
            img[y:y + h, x:x + w] = c
        out.append(img)
    return np.stack(out)


def _cluttered(n=8):
    """n BGR scenes for the conf 0.001 pass: scene 0 is uniform noise (the real model leaves ~350 anchors above 0.001 on
    it, ~240 survive NMS), the others a clean scene with 60 * (1 + s % 3) crops, rescaled by 0.4-1.6, pasted at random."""
    from PIL import Image
    crops = [np.asarray(Image.open(f).convert("RGB"))[..., ::-1].copy() for f in _CROPS]
    base = _scenes(n)
    rng = np.random.default_rng(7)
    out = []
    for s in range(n):
        if s == 0:
            out.append(rng.integers(0, 256, (640, 640, 3), dtype=np.uint8))
            continue
        img = base[s].copy()
        for _ in range(60 * (1 + s % 3)):
            c = crops[int(rng.integers(len(crops)))]
            sc = rng.uniform(0.4, 1.6)
            h, w = max(8, int(c.shape[0] * sc)), max(8, int(c.shape[1] * sc))
            cc = np.asarray(Image.fromarray(c).resize((w, h), Image.BILINEAR))
            x, y = int(rng.integers(0, 640 - w)), int(rng.integers(0, 640 - h))
            img[y:y + h, x:x + w] = cc
        out.append(img)
    return np.stack(out)


def _iou(a, b):
    iw = max(0.0, min(a[2], b[2]) - max(a[0], b[0]))
    ih = max(0.0, min(a[3], b[3]) - max(a[1], b[1]))
    inter = iw * ih
    return inter / ((a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter + 1e-6)


def _oracle_out0(layers, imgs):
    from oracle import ncnn_ref
    x = torch.from_numpy(imgs[..., ::-1].astype(np.float32) * np.float32(1 / 255.0)).permute(0, 3, 1, 2).contiguous()
    return np.concatenate([ncnn_ref.run_graph(layers, x[i:i + 8])["out0"].numpy() for i in range(0, len(imgs), 8)])


def _decisions_exact_on_device_out0(outs, got0, imgs, conf):
    """The oracle's postprocess + ROI rule on the DEVICE's own out0 reproduce the device's result lists bit for bit: the
    whole decision chain (filter, un-letterbox, per-class greedy NMS over every candidate, int truncation, area filter) on
    the real model's score distribution."""
    from oracle import postprocess_ref as P
    n_boxes = n_cand = 0
    for i in range(len(imgs)):
        hw = imgs[i].shape[:2]
        eb, es, ec = P.postprocess(got0[i], hw, 1.0, (0.0, 0.0), conf, IOU)
        res, met = outs[i]
        assert met.num_detections == len(eb), f"scene {i}: num_detections {met.num_detections} vs {len(eb)} from the device's out0"
        _, valid = P.roi_rects(eb, hw[0], hw[1], MIN_AREA)
        assert len(res) == len(valid), f"scene {i}: {len(res)} results vs {len(valid)}"
        for r, k in zip(res, valid):
            assert r["bbox"] == tuple(eb[k].astype(int)) and r["det_conf"] == float(es[k]) and r["det_class"] == int(ec[k])
        n_boxes += len(res)
        n_cand += int((got0[i, 4:].max(axis=0) > conf).sum())
    return n_boxes, n_cand


def _models(tmp_path):
    from oracle import ncnn_ref, shufflenet_ref as S
    sd = S.seeded_state_dict(91)
    cls_path = str(tmp_path / "cls.pth")
    torch.save(sd, cls_path)
    return ncnn_ref.load_model(*_REAL), S.build(91, sd), cls_path


@pytest.mark.skipif(not _HAVE, reason="reference v1 model / debug_rois crops not staged under oracle/_ref")
@pytest.mark.parametrize("CONF", [0.10, 0.25], ids=["conf0.10", "conf0.25"])
def test_real_weights_fp32_post_nms_sets_equal_oracle(tmp_path, CONF):
    from litepi import HybridPipeline
    from oracle import pipeline_ref
    layers, cls_model, cls_path = _models(tmp_path)
    imgs = _scenes(16)
    cpu = pipeline_ref.CpuPipeline(layers, cls_model)
    pipe = HybridPipeline(_REAL[0], _REAL[1], cls_path, "shufflenetv2", num_classes=91, precision="fp32", max_batch=16, max_det=300)
    try:
        outs = pipe.run_batch(list(imgs), CONF, IOU, MIN_AREA)
        got0 = pipe.engine.detect_raw(imgs)
    finally:
        pipe.engine.close()
    _decisions_exact_on_device_out0(outs, got0, imgs, CONF)
    total, smax, per_image = 0, 0.0, []
    for i in range(len(imgs)):
        exp, exp_numdet = cpu.run(imgs[i], CONF, IOU, MIN_AREA)
        res, met = outs[i]
        assert met.num_detections == exp_numdet, f"scene {i}: {met.num_detections} boxes before the area filter vs oracle {exp_numdet}"
        assert len(res) == len(exp), f"scene {i}: {len(res)} results vs oracle {len(exp)}"
        for r, x in zip(res, exp):
            assert abs(r["det_conf"] - x["det_conf"]) <= 1e-3
            assert np.abs(np.array(r["bbox"]) - np.array(x["bbox"])).max() <= 1   # int truncation of boxes equal within 1e-3 px
            assert r["cls_class"] == x["cls_class"] and abs(r["cls_conf"] - x["cls_conf"]) <= 2e-3
            smax = max(smax, x["det_conf"])
        total += len(res)
        per_image.append(len(res))
    print(f"real v1 weights, fp32, conf {CONF}: {total} post-NMS boxes on 16 scenes equal the oracle's (per scene {per_image}), best score {smax:.3f}")
    assert total >= 24 and smax >= 0.5, "the real model must find real signs in these scenes"


@pytest.mark.skipif(not _HAVE, reason="reference v1 model / debug_rois crops not staged under oracle/_ref")
def test_real_weights_fp32_evaluation_pass_conf_0_001(tmp_path):
    """The reference's evaluation pass (--yolo_conf 0.001, e2e.py:975-992) on cluttered scenes: hundreds of candidates per
    image through the sort + greedy sweep on the real score distribution.
      1. decisions exact on the device's own out0 (bit for bit, every image);
      2. against the ORACLE's out0 the post-NMS lists are equal box for box, except boxes that the documented fp32
         tolerance (1e-3 on scores, north_star) can flip: a one-sided box must have its score within 2e-3 of conf, or an NMS
         near-tie (a higher-scored candidate of the other side whose IoU with it is within 2e-3 of the threshold, or whose
         score is within 2e-3 of its own: the greedy order can swap).  Flips are counted, printed and capped at 1 % of the boxes."""
    from litepi import HybridPipeline
    from oracle import postprocess_ref as P
    layers, cls_model, cls_path = _models(tmp_path)
    imgs = _cluttered(8)
    conf = 0.001
    ref0 = _oracle_out0(layers, imgs)
    pipe = HybridPipeline(_REAL[0], _REAL[1], cls_path, "shufflenetv2", num_classes=91, precision="fp32", max_batch=8, max_det=300)
    try:
        outs = pipe.run_batch(list(imgs), conf, IOU, MIN_AREA)
        got0 = pipe.engine.detect_raw(imgs)
    finally:
        pipe.engine.close()
    assert np.abs(got0[:, 4] - ref0[:, 4]).max() <= 1e-3
    n_boxes, n_cand = _decisions_exact_on_device_out0(outs, got0, imgs, conf)
    assert n_cand >= 800 and n_boxes >= 250, f"the evaluation pass must be busy: {n_cand} candidates, {n_boxes} boxes"
    n_equal = n_flip = 0
    for i in range(len(imgs)):
        hw = imgs[i].shape[:2]
        eb, es, _ = P.postprocess(ref0[i], hw, 1.0, (0.0, 0.0), conf, IOU)
        _, valid = P.roi_rects(eb, hw[0], hw[1], MIN_AREA)
        exp = [(eb[k].astype(int), float(es[k])) for k in valid]
        dev = [(np.array(r["bbox"]), r["det_conf"]) for r in outs[i][0]]
        cb, cs, _ = P.postprocess(ref0[i], hw, 1.0, (0.0, 0.0), max(conf - 2e-3, 0.0), 1.0)   # every oracle candidate
        same = lambda a, b: np.abs(a[0] - b[0]).max() <= 1 and abs(a[1] - b[1]) <= 1e-3
        for mine, other in ((exp, dev), (dev, exp)):
            for bx in mine:
                if any(same(bx, q) for q in other):
                    n_equal += 1
                    continue
                near_conf = bx[1] <= conf + 2e-3
                near_tie = any(sc >= bx[1] - 2e-3 and (abs(_iou(q, bx[0]) - IOU) <= 2e-3 or (abs(sc - bx[1]) <= 2e-3 and _iou(q, bx[0]) > IOU - 2e-3))
                               for q, sc in zip(cb, cs))
                assert near_conf or near_tie, f"scene {i}: box {bx} is in one list only and neither conf nor an NMS near-tie explains it"
                n_flip += 1
    print(f"real v1 weights, fp32, conf 0.001 on cluttered scenes: {n_cand} candidates -> {n_boxes} boxes; decisions exact on the device's out0; "
          f"{n_equal // 2} boxes equal the oracle's, {n_flip} one-sided (all explained by the 1e-3 tolerance)")
    assert n_flip <= max(2, n_boxes // 100)


@pytest.mark.skipif(not _HAVE, reason="reference v1 model / debug_rois crops not staged under oracle/_ref")
@pytest.mark.parametrize("CONF,cap", [(0.10, 16), (0.25, 16), (0.25, 64)], ids=["conf0.10-cap16", "conf0.25-cap16", "conf0.25-cap64"])
def test_real_weights_fp16_no_stability_filter(tmp_path, CONF, cap):
    """cap 64 = the benchmarked plan (handle built for 64 images: the whole-C2f launches and the capacity-64 tile shapes) on
    64 distinct scenes; cap 16 = round 3's handle."""
    from litepi import HybridPipeline
    from oracle import postprocess_ref as P
    layers, cls_model, cls_path = _models(tmp_path)
    imgs = _scenes(cap)
    ref0 = _oracle_out0(layers, imgs)
    pipe = HybridPipeline(_REAL[0], _REAL[1], cls_path, "shufflenetv2", num_classes=91, precision="fp16", max_batch=cap, max_det=300)
    try:
        got0 = pipe.engine.detect_raw(imgs)
        outs = pipe.run_batch(list(imgs), CONF, IOU, MIN_AREA)
    finally:
        pipe.engine.close()
    # measured fp16 error of the raw head output on these weights (all 8400 anchors; boxes where the score matters)
    err_s = float(np.abs(got0[:, 4] - ref0[:, 4]).max())
    hot = ref0[:, 4] > 0.02
    err_b = float(np.abs(got0[:, :4] - ref0[:, :4]).transpose(0, 2, 1)[hot].max()) if hot.any() else 0.0
    n_oracle = n_found = n_band = n_dev = n_extra = n_swap = 0
    worst_px = worst_sc = 0.0
    for i in range(len(imgs)):
        hw = imgs[i].shape[:2]
        eb, es, _ = P.postprocess(ref0[i], hw, 1.0, (0.0, 0.0), CONF, IOU)
        _, valid = P.roi_rects(eb, hw[0], hw[1], MIN_AREA)
        exp = [(eb[k], float(es[k])) for k in valid]
        res = outs[i][0]
        dev = [(np.array(r["bbox"], np.float64), r["det_conf"]) for r in res]
        # boxes just under the threshold that fp16 may legitimately lift over it
        nb, ns, _ = P.postprocess(ref0[i], hw, 1.0, (0.0, 0.0), CONF - BAND, IOU)
        ab, as_, _ = P.postprocess(ref0[i], hw, 1.0, (0.0, 0.0), CONF - BAND, 1.0)   # every oracle candidate (iou 1.0: nothing suppressed)
        for box, sc in exp:
            n_oracle += 1
            if sc <= CONF + BAND:
                n_band += 1
                continue
            d = [(np.abs(db - box.astype(int)).max(), abs(ds - sc)) for db, ds in dev]
            hit = [q for q in d if q[0] <= BOX_PX + 1 and q[1] <= BAND]   # (+1: int truncation of a box within BOX_PX)
            if not hit:
                # representative swap: the same object has a second oracle candidate whose score is within 2 * BAND of this
                # one's, and the device kept that one (its fp16 score came out on top).  Counted, printed, capped at 2 %.
                swap = [(q, s2) for q, s2 in zip(ab, as_) if abs(s2 - sc) <= 2 * BAND and _iou(q, box) > IOU
                        and any(np.abs(db - q.astype(int)).max() <= BOX_PX + 1 and abs(ds - s2) <= BAND for db, ds in dev)]
                assert swap, f"scene {i}: oracle box {box} score {sc:.4f} not found by the fp16 path (candidates {sorted(d)[:2]})"
                n_swap += 1
                continue
            n_found += 1
            worst_px = max(worst_px, min(q[0] for q in hit))
            worst_sc = max(worst_sc, min(q[1] for q in hit))
        for db, ds in dev:
            n_dev += 1
            on_oracle = any(np.abs(db - b.astype(int)).max() <= BOX_PX + 1 for b in (nb if n_swap == 0 else ab))
            if not on_oracle:
                n_extra += 1
                assert ds <= CONF + BAND, f"scene {i}: device box {db} score {ds:.4f} has no oracle counterpart"
    _decisions_exact_on_device_out0(outs, got0, imgs, CONF)
    print(f"real v1 weights, fp16, conf {CONF}, capacity {cap}, no stability filter: out0 score err {err_s:.5f}, box err (score > 0.02) {err_b:.3f} px; "
          f"{n_found} of {n_oracle - n_band} oracle boxes outside the +-{BAND} band found (worst {worst_px:.0f} px / {worst_sc:.5f}), "
          f"{n_band} in the band, {n_swap} representative swaps between near-tied candidates, {n_dev} device boxes, {n_extra} extra (all inside the band)")
    assert err_s <= 2 * BAND and n_oracle - n_band >= 24 and n_swap <= max(1, (n_oracle - n_band) // 50)


@pytest.mark.skipif(not _HAVE, reason="reference v1 model / debug_rois crops not staged under oracle/_ref")
def test_real_weights_fp16_evaluation_pass_conf_0_001(tmp_path):
    """fp16 at the reference's evaluation threshold (--yolo_conf 0.001) on the cluttered scenes.  Below ~0.02 the fp16 score
    error (0.004 measured) is as large as the scores, so box-for-box equality of the junk is not a meaningful ask; what
    the pass feeds is evaluate_predictions.  Checked:
      1. decisions exact on the device's own out0 (every image, hundreds of candidates each);
      2. every oracle box with score > 0.05 is found within 1 px / BAND, unless an NMS near-tie of the fp16 error's size
         explains it (counted, <= 5 %);
      3. mAP of the device's predictions equals the oracle predictions' mAP within 0.01 (mAP@0.5) / 0.02 (mAP@0.5:0.95),
         both scored by the port of evaluate_predictions (e2e.py:656-824) against the oracle's confident boxes
         (score > 0.25) as ground truth."""
    from litepi import HybridPipeline
    from litepi.e2e import evaluate_predictions
    from oracle import postprocess_ref as P
    layers, cls_model, cls_path = _models(tmp_path)
    imgs = _cluttered(8)
    conf = 0.001
    ref0 = _oracle_out0(layers, imgs)
    pipe = HybridPipeline(_REAL[0], _REAL[1], cls_path, "shufflenetv2", num_classes=91, precision="fp16", max_batch=8, max_det=300)
    try:
        got0 = pipe.engine.detect_raw(imgs)
        outs = pipe.run_batch(list(imgs), conf, IOU, MIN_AREA)
    finally:
        pipe.engine.close()
    n_boxes, n_cand = _decisions_exact_on_device_out0(outs, got0, imgs, conf)
    n_conf = n_found = n_tie = 0
    gts, p_dev, p_ora = [], [], []
    for i in range(len(imgs)):
        hw = imgs[i].shape[:2]
        eb, es, _ = P.postprocess(ref0[i], hw, 1.0, (0.0, 0.0), conf, IOU)
        _, valid = P.roi_rects(eb, hw[0], hw[1], MIN_AREA)
        exp = [(eb[k].astype(int), float(es[k])) for k in valid]
        dev = [(np.array(r["bbox"]), r["det_conf"]) for r in outs[i][0]]
        cb, cs, _ = P.postprocess(ref0[i], hw, 1.0, (0.0, 0.0), 0.0, 1.0)
        for bx, sc in exp:
            if sc <= 0.05:
                continue
            n_conf += 1
            if any(np.abs(db - bx).max() <= BOX_PX + 1 and abs(ds - sc) <= BAND for db, ds in dev):
                n_found += 1
                continue
            tie = any(s2 >= sc - 2 * BAND and abs(_iou(q, bx) - IOU) <= 0.03 for q, s2 in zip(cb, cs))
            assert tie, f"scene {i}: oracle box {bx} score {sc:.4f} missing from the fp16 result and no NMS near-tie explains it"
            n_tie += 1
        # the class the mAP port matches on: 0 for every box (single-class detector; the classifier is not under test here)
        gts.append([(0, int(b[0]), int(b[1]), int(b[2]), int(b[3])) for b, sc in exp if sc > 0.25])
        p_ora.append([{"bbox": tuple(int(v) for v in b), "conf": sc, "cls_class": 0} for b, sc in exp])
        p_dev.append([{"bbox": tuple(int(v) for v in b), "conf": sc, "cls_class": 0} for b, sc in dev])
    m_dev = evaluate_predictions(p_dev, gts, num_classes=1)
    m_ora = evaluate_predictions(p_ora, gts, num_classes=1)
    print(f"real v1 weights, fp16, conf 0.001 on cluttered scenes: {n_cand} candidates -> {n_boxes} boxes; {n_found} of {n_conf} oracle boxes with "
          f"score > 0.05 found, {n_tie} NMS near-ties; mAP50 {m_dev['mAP50']:.4f} vs oracle {m_ora['mAP50']:.4f}, "
          f"mAP50-95 {m_dev['mAP50_95']:.4f} vs {m_ora['mAP50_95']:.4f}")
    assert n_conf >= 40 and n_tie <= max(2, n_conf // 20)
    assert abs(m_dev["mAP50"] - m_ora["mAP50"]) <= 0.01 and abs(m_dev["mAP50_95"] - m_ora["mAP50_95"]) <= 0.02

