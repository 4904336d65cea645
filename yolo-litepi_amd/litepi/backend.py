"""Drop-in replacements for the three classes the reference's ``main()`` builds from its CLI
flags (``src/tt100k/pipeline/e2e.py``): ``NCNNDetector`` (:195), ``PyTorchClassifier`` (:350)
and ``HybridPipeline`` (:399), plus the ``PipelineMetrics`` dataclass (:34).  Same constructor
arguments, same method signatures, same return types and empty-result quirks; the arithmetic
runs in liblitepi_hip.so on an MI355X instead of NCNN / torch-CPU / NumPy.

``Engine`` is the thin object wrapper over the C-ABI handle that the three classes share.
"""
from __future__ import annotations

import ctypes as C
import os
import time
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _ffi
from ._ffi import DET_DTYPE, LpConfig, LpKernelTime, LpTiming, check

_PREC = {"fp32": _ffi.LP_FP32, "fp16": _ffi.LP_FP16, "float32": _ffi.LP_FP32, "float16": _ffi.LP_FP16, "half": _ffi.LP_FP16}


def _as_bgr(img: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(img, dtype=np.uint8)
    if a.ndim != 3 or a.shape[2] != 3:
        raise ValueError(f"expected a BGR uint8 HxWx3 image, got shape {img.shape}")
    return a


CLS_ARCHS = ("resnet18", "efficientnet", "mobilenetv2", "shufflenetv2")   # the --clf_arch choices of e2e.py:1021


class Engine:
    """One GPU pipeline handle (not thread-safe; one per GPU)."""

    def __init__(self, precision: str = "fp16", max_batch: int = 1, max_det: int = 300, num_classes: int = 58,
                 det_input: int = 640, cls_input: int = 64, device: int = 0, max_rois: int = 0, conv_impl: int = 0,
                 numerics: str = "e2e", cls_arch: str = "shufflenetv2"):
        """numerics: "e2e" = HybridPipeline of e2e.py (default), "e2e_optimize" = HybridPipelineOptimized of e2e_optimize.py
        (other ROI clip rule, cv2-linear ROI resize).  cls_arch: "shufflenetv2" | "resnet18" | "mobilenetv2" | "efficientnet"
        (the four choices of build_classifier, e2e.py:320-333)."""
        self.lib = _ffi.load_library()
        cfg = LpConfig()
        self.lib.lp_default_config(C.byref(cfg))
        cfg.device, cfg.precision, cfg.max_batch, cfg.max_det = device, _PREC[precision], max_batch, max_det
        cfg.num_classes, cfg.det_input, cfg.cls_input, cfg.max_rois, cfg.conv_impl = num_classes, det_input, cls_input, max_rois, conv_impl
        cfg.numerics = {"e2e": 0, "e2e_optimize": 1}[numerics]
        cfg.cls_arch = {"shufflenetv2": 0, "resnet18": 1, "mobilenetv2": 2, "efficientnet": 3}[cls_arch]
        self.numerics, self.cls_arch = numerics, cls_arch
        self.cfg = cfg
        self.precision = precision
        self._h = C.c_void_p()
        check(self.lib, self.lib.lp_create(C.byref(cfg), C.byref(self._h)))
        self.has_detector = False
        self.has_classifier = False

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            self.lib.lp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- models ---------------------------------------------------------------------------
    def load_detector(self, param_path: str, bin_path: str) -> None:
        check(self.lib, self.lib.lp_load_detector_ncnn(self._h, os.fsencode(param_path), os.fsencode(bin_path)))
        self.has_detector = True
        a, nc, rm, macs = C.c_int(), C.c_int(), C.c_int(), C.c_double()
        check(self.lib, self.lib.lp_detector_info(self._h, C.byref(a), C.byref(nc), C.byref(rm), C.byref(macs)))
        self.num_anchors, self.det_classes, self.reg_max, self.det_macs = a.value, nc.value, rm.value, macs.value

    def load_classifier(self, state_dict: Dict[str, object]) -> None:
        """state_dict: torchvision shufflenet_v2_x1_0 keys -> torch tensors or ndarrays."""
        names, arrays = [], []
        for k, v in state_dict.items():
            if k.endswith("num_batches_tracked"):
                continue
            a = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
            names.append(k.encode())
            arrays.append(np.ascontiguousarray(a, dtype=np.float32))
        n = len(names)
        c_names = (C.c_char_p * n)(*names)
        c_data = (C.c_void_p * n)(*[a.ctypes.data for a in arrays])
        shapes = [np.asarray(a.shape, dtype=np.int64) for a in arrays]
        c_shapes = (C.c_void_p * n)(*[s.ctypes.data for s in shapes])
        c_ndims = (C.c_int * n)(*[a.ndim for a in arrays])
        check(self.lib, self.lib.lp_load_classifier_tensors(self._h, n, c_names, c_data, c_shapes, c_ndims))
        self.has_classifier = True

    # ---- inference ---------------------------------------------------------------------------
    def detect_raw(self, bgr: np.ndarray) -> np.ndarray:
        """uint8 BGR [B,S,S,3] -> fp32 out0 [B,4+nc,A] (parity hook for ex.extract('out0'))."""
        a = np.ascontiguousarray(bgr, dtype=np.uint8)
        S = self.cfg.det_input
        if a.ndim == 3:
            a = a[None]
        if a.shape[1:] != (S, S, 3):
            raise ValueError(f"detect_raw expects [B,{S},{S},3], got {a.shape}")
        out = np.empty((a.shape[0], 4 + self.det_classes, self.num_anchors), np.float32)
        check(self.lib, self.lib.lp_detect_raw(self._h, a.ctypes.data, a.shape[0], out.ctypes.data))
        return out

    def _img_args(self, images: Sequence[np.ndarray]):
        imgs = [_as_bgr(i) for i in images]
        n = len(imgs)
        ptrs = (C.c_void_p * n)(*[i.ctypes.data for i in imgs])
        hs = (C.c_int * n)(*[i.shape[0] for i in imgs])
        ws = (C.c_int * n)(*[i.shape[1] for i in imgs])
        return imgs, ptrs, hs, ws

    def detect(self, images: Sequence[np.ndarray], conf: float, iou: float):
        imgs, ptrs, hs, ws = self._img_args(images)
        B = len(imgs)
        dets = np.zeros((B, self.cfg.max_det), dtype=DET_DTYPE)
        counts = (C.c_int * B)()
        check(self.lib, self.lib.lp_detect(self._h, ptrs, hs, ws, B, conf, iou, dets.ctypes.data, counts))
        return dets, np.array(counts[:], dtype=np.int64)

    def run_batch(self, images: Sequence[np.ndarray], conf: float, iou: float, min_area: int):
        imgs, ptrs, hs, ws = self._img_args(images)
        B = len(imgs)
        dets = np.zeros((B, self.cfg.max_det), dtype=DET_DTYPE)
        counts, num_det = (C.c_int * B)(), (C.c_int * B)()
        conf_avg = (C.c_float * B)()
        timing = LpTiming()
        check(self.lib, self.lib.lp_run_batch(self._h, ptrs, hs, ws, B, conf, iou, int(min_area), dets.ctypes.data, counts,
                                              num_det, conf_avg, C.byref(timing)))
        self.last_det_conf_avg = np.array(conf_avg[:], dtype=np.float32)
        return dets, np.array(counts[:], dtype=np.int64), np.array(num_det[:], dtype=np.int64), timing

    def run_batch_device(self, dev_imgs: int, B: int, H: int, W: int, conf: float, iou: float, min_area: int,
                         dev_dets: int, dev_counts: int) -> None:
        check(self.lib, self.lib.lp_run_batch_device(self._h, C.c_void_p(dev_imgs), B, H, W, conf, iou, int(min_area),
                                                     C.c_void_p(dev_dets), C.c_void_p(dev_counts)))

    def roi_overflow(self) -> Tuple[int, int]:
        """(classified, kept) of the last run_batch_device call (synchronises): kept > classified means max_rois was too small."""
        a, b = C.c_int(), C.c_int()
        check(self.lib, self.lib.lp_roi_overflow(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def classify(self, rois: Sequence[np.ndarray]) -> Tuple[np.ndarray, np.ndarray]:
        imgs, ptrs, hs, ws = self._img_args(rois)
        R = len(imgs)
        ids = (C.c_int * R)()
        probs = np.empty((R, self.cfg.num_classes), np.float32)
        check(self.lib, self.lib.lp_classify(self._h, ptrs, hs, ws, R, ids, probs.ctypes.data_as(C.POINTER(C.c_float))))
        return np.array(ids[:], dtype=np.int64), probs

    # ---- streams / profiling -------------------------------------------------------------------
    def set_stream(self, stream_handle: int) -> None:
        check(self.lib, self.lib.lp_set_stream(self._h, C.c_void_p(stream_handle)))

    def synchronize(self) -> None:
        check(self.lib, self.lib.lp_synchronize(self._h))

    def profile_next(self, enable: bool = True) -> None:
        check(self.lib, self.lib.lp_profile_next(self._h, int(enable)))

    def profile_read(self) -> List[dict]:
        n = C.c_int()
        cap = 512
        buf = (LpKernelTime * cap)()
        check(self.lib, self.lib.lp_profile_read(self._h, buf, cap, C.byref(n)))
        return [dict(name=buf[i].name.decode(), layer=buf[i].layer.decode(), ms=buf[i].ms, flops=buf[i].flops,
                     bytes=buf[i].bytes) for i in range(min(cap, n.value))]

    # ---- test hooks ----------------------------------------------------------------------------
    def debug_blob(self, name: str) -> np.ndarray:
        c, h, w = C.c_int(), C.c_int(), C.c_int()
        check(self.lib, self.lib.lp_debug_blob(self._h, name.encode(), None, 0, C.byref(c), C.byref(h), C.byref(w)))
        out = np.empty((1, c.value, h.value, w.value), np.float32)
        check(self.lib, self.lib.lp_debug_blob(self._h, name.encode(), out.ctypes.data_as(C.POINTER(C.c_float)), out.size,
                                               C.byref(c), C.byref(h), C.byref(w)))
        return out

    def test_conv(self, x, w, bias, stride=1, act=0, res=None, impl=0) -> np.ndarray:
        fp = C.POINTER(C.c_float)
        x = np.ascontiguousarray(x, np.float32)
        w = np.ascontiguousarray(w, np.float32)
        N, Cin, H, W = x.shape
        Cout, _, k, _ = w.shape
        Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
        y = np.empty((N, Cout, Ho, Wo), np.float32)
        b = None if bias is None else np.ascontiguousarray(bias, np.float32)
        r = None if res is None else np.ascontiguousarray(res, np.float32)
        check(self.lib, self.lib.lp_test_conv(self._h, impl, x.ctypes.data_as(fp), N, Cin, H, W, w.ctypes.data_as(fp),
                                              None if b is None else b.ctypes.data_as(fp), Cout, k, stride, act,
                                              None if r is None else r.ctypes.data_as(fp), y.ctypes.data_as(fp)))
        return y

    def test_postprocess(self, out0, orig_shape, ratio, pad, conf, iou, min_area: int = -1, max_det: int = 0, with_rects: bool = False):
        """filter + NMS (+ ROI clip / area filter when min_area >= 0) on a host out0 [4+nc, A].  Returns the records, or
        (records, int rects [n,4], pre-filter count) with ``with_rects``."""
        fp = C.POINTER(C.c_float)
        o = np.ascontiguousarray(out0, np.float32)
        nc, A = o.shape[0] - 4, o.shape[1]
        dets = np.zeros(A, dtype=DET_DTYPE)
        rects = np.zeros((A, 4), dtype=np.int32)
        cnt, num = C.c_int(), C.c_int()
        check(self.lib, self.lib.lp_test_postprocess(self._h, o.ctypes.data_as(fp), nc, A, int(orig_shape[0]), int(orig_shape[1]),
                                                     float(ratio), float(pad[0]), float(pad[1]), float(conf), float(iou),
                                                     int(min_area), int(max_det), dets.ctypes.data,
                                                     rects.ctypes.data_as(C.POINTER(C.c_int)), C.byref(cnt), C.byref(num)))
        if with_rects:
            return dets[:cnt.value], rects[:cnt.value], num.value
        return dets[:cnt.value]

    def test_nms_boxes(self, boxes, scores, classes, orig_shape, iou, min_area: int = -1, max_det: int = 0):
        """NMS + ROI clip / area filter on host xyxy boxes -> (records, int rects [n,4], pre-filter count)."""
        fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int)
        b = np.ascontiguousarray(boxes, np.float32).reshape(-1, 4)
        sc = np.ascontiguousarray(scores, np.float32)
        n = len(b)
        cl = None if classes is None else np.ascontiguousarray(classes, np.int32)
        dets = np.zeros(max(n, 1), dtype=DET_DTYPE)
        rects = np.zeros((max(n, 1), 4), dtype=np.int32)
        cnt, num = C.c_int(), C.c_int()
        check(self.lib, self.lib.lp_test_nms_boxes(self._h, b.ctypes.data_as(fp), sc.ctypes.data_as(fp),
                                                   None if cl is None else cl.ctypes.data_as(ip), n, int(orig_shape[0]),
                                                   int(orig_shape[1]), float(iou), int(min_area), int(max_det), dets.ctypes.data,
                                                   rects.ctypes.data_as(ip), C.byref(cnt), C.byref(num)))
        return dets[:cnt.value], rects[:cnt.value], num.value

    def test_roi_resize(self, rois: Sequence[np.ndarray]) -> np.ndarray:
        imgs, ptrs, hs, ws = self._img_args(rois)
        S = self.cfg.cls_input
        out = np.empty((len(imgs), S, S, 3), np.uint8)
        check(self.lib, self.lib.lp_test_roi_resize(self._h, ptrs, hs, ws, len(imgs), out.ctypes.data))
        return out

    def test_letterbox(self, img: np.ndarray):
        a = _as_bgr(img)
        S = self.cfg.det_input
        out = np.empty((S, S, 3), np.uint8)
        r, pw, ph = C.c_float(), C.c_float(), C.c_float()
        check(self.lib, self.lib.lp_test_letterbox(self._h, a.ctypes.data, a.shape[0], a.shape[1], out.ctypes.data,
                                                   C.byref(r), C.byref(pw), C.byref(ph)))
        return out, r.value, (pw.value, ph.value)


# ==================== reference-compatible surface ====================
@dataclass
class PipelineMetrics:
    """Same fields as the reference dataclass (e2e.py:34-62)."""
    t_detection: float = 0.0
    t_roi_extract: float = 0.0
    t_classification: float = 0.0
    t_postprocess: float = 0.0
    t_total: float = 0.0
    fps: float = 0.0
    num_detections: int = 0
    det_confidence_avg: float = 0.0
    cls_confidence_avg: float = 0.0
    cpu_percent: float = 0.0
    memory_mb: float = 0.0
    temperature: float = 0.0
    precision: float = 0.0
    recall: float = 0.0
    f1: float = 0.0
    level: str = "HIP(MI355X)"


_SYSM_CACHE = [0.0, (0.0, 0.0, 0.0)]
_SYSM_PERIOD_S = 0.25


def _system_metrics():
    """cpu_percent / memory_mb / temperature as the reference samples them after each run (e2e.py:509-516).  One sample costs
    ~1.2 ms (psutil + a sysfs read): more than a 64-image batch takes on the GPU.  The reference's calls are 50+ ms apart; here
    the sample is refreshed at most every 0.25 s and calls in between report the last one."""
    now = time.monotonic()
    if now - _SYSM_CACHE[0] < _SYSM_PERIOD_S:
        return _SYSM_CACHE[1]
    cpu = mem = temp = 0.0
    try:
        import psutil
        cpu = float(psutil.cpu_percent())
        mem = float(psutil.Process().memory_info().rss) / 1024 / 1024
    except Exception:  # noqa: BLE001 - psutil missing: fields stay 0 like the reference's temperature fallback
        pass
    try:
        with open("/sys/class/thermal/thermal_zone0/temp") as f:
            temp = float(f.read()) / 1000.0
    except Exception:  # noqa: BLE001
        pass
    _SYSM_CACHE[0], _SYSM_CACHE[1] = now, (cpu, mem, temp)
    return cpu, mem, temp


def _empty_detect():
    # the reference returns float64 empties here (e2e.py:264,292-294)
    return np.empty((0, 4)), np.empty((0,)), np.empty((0,))


class NCNNDetector:
    """e2e.py:195-316.  ``use_gpu``/``num_threads`` are accepted and ignored (the detector always
    runs on the HIP device)."""

    def __init__(self, param_path: str, bin_path: str, input_size: int = 640, use_gpu: bool = False, num_threads: int = 4,
                 input_name: str = "in0", output_name: str = "out0", *, precision: str = "fp16", max_batch: int = 1,
                 max_det: int = 300, device: int = 0, _engine: Optional[Engine] = None):
        self.input_size = input_size
        self.input_name = input_name
        self.output_name = output_name
        print("[HIP Detector] Loading model...")
        print(f"  Param: {param_path}")
        print(f"  Bin: {bin_path}")
        self.engine = _engine or Engine(precision=precision, max_batch=max_batch, max_det=max_det, det_input=input_size,
                                        device=device)
        try:
            self.engine.load_detector(param_path, bin_path)
        except _ffi.LitepiError as e:  # e2e.py:213-216 raises RuntimeError on load failure
            raise RuntimeError(str(e)) from e
        print(f"  Device: HIP:{self.engine.cfg.device} ({self.engine.precision})")
        print(f"  Input size: {input_size}x{input_size}")

    def detect_batch(self, images: Sequence[np.ndarray], conf_threshold: float = 0.5, iou_threshold: float = 0.45):
        try:
            dets, counts = self.engine.detect(images, conf_threshold, iou_threshold)
        except _ffi.LitepiError as e:
            # the reference returns empty arrays when the ENGINE call fails (extract != 0, e2e.py:309-310): that is a HIP
            # runtime failure here.  Misuse (batch over capacity, unsupported shapes, model not loaded) is raised.
            if e.code != _ffi.LP_ERR_HIP:
                raise
            print(f"[HIP Detector] engine failure, returning no detections: {e}")
            return [_empty_detect() for _ in images]
        out = []
        for i, n in enumerate(counts):
            if n == 0:
                out.append(_empty_detect())
                continue
            d = dets[i, :n]
            boxes = np.stack([d["x1"], d["y1"], d["x2"], d["y2"]], axis=1).astype(np.float32)
            out.append((boxes, d["det_conf"].astype(np.float32), d["det_class"].astype(np.int64)))
        return out

    def detect(self, image: np.ndarray, conf_threshold: float = 0.5, iou_threshold: float = 0.45):
        return self.detect_batch([image], conf_threshold, iou_threshold)[0]


def load_classifier_state(model_path: Optional[str], num_classes: int, arch: str = "shufflenetv2"):
    """torch is used only to READ the checkpoint (weights_only: nothing from the file is executed).
    Returns (state_dict, loaded_flag); a missing/unreadable file gives seeded random weights and a
    warning, like the reference's silent random-init fallback (e2e.py:337-343)."""
    import torch

    if model_path and os.path.exists(model_path):
        try:
            sd = torch.load(model_path, map_location="cpu", weights_only=True)
            if isinstance(sd, dict) and "state_dict" in sd:
                sd = sd["state_dict"]
            print(f"Loaded classifier weights from {model_path}")
            return sd, True
        except Exception as e:  # noqa: BLE001 - mirror the reference's catch-all
            print(f"WARNING: could not load classifier weights ({e}); the classifier stays RANDOM-INIT")
    else:
        print(f"WARNING: classifier weights {model_path!r} not found; the classifier stays RANDOM-INIT")
    return {"resnet18": random_resnet18_state, "mobilenetv2": random_mobilenetv2_state, "efficientnet": random_efficientnet_state,
            "shufflenetv2": random_shufflenet_state}[arch](num_classes), False


def random_shufflenet_state(num_classes: int, seed: int = 0) -> Dict[str, np.ndarray]:
    """Seeded random ShuffleNetV2 x1.0 state_dict with torchvision's key names and shapes."""
    rng = np.random.default_rng(seed)
    sd: Dict[str, np.ndarray] = {}

    def conv(name, co, ci, k):
        sd[name + ".weight"] = (rng.standard_normal((co, ci, k, k)) * (1.7 / (ci * k * k)) ** 0.5).astype(np.float32)

    def bn(name, c):
        sd[name + ".weight"] = rng.uniform(0.75, 1.25, c).astype(np.float32)
        sd[name + ".bias"] = (rng.standard_normal(c) * 0.1).astype(np.float32)
        sd[name + ".running_mean"] = (rng.standard_normal(c) * 0.1).astype(np.float32)
        sd[name + ".running_var"] = rng.uniform(0.75, 1.25, c).astype(np.float32)

    conv("conv1.0", 24, 3, 3)
    bn("conv1.1", 24)
    inp = 24
    for stage, rep, oup in (("stage2", 4, 116), ("stage3", 8, 232), ("stage4", 4, 464)):
        bf = oup // 2
        for r in range(rep):
            p = f"{stage}.{r}."
            if r == 0:
                conv(p + "branch1.0", inp, 1, 3); bn(p + "branch1.1", inp)
                conv(p + "branch1.2", bf, inp, 1); bn(p + "branch1.3", bf)
                conv(p + "branch2.0", bf, inp, 1)
            else:
                conv(p + "branch2.0", bf, bf, 1)
            bn(p + "branch2.1", bf)
            conv(p + "branch2.3", bf, 1, 3); bn(p + "branch2.4", bf)
            conv(p + "branch2.5", bf, bf, 1); bn(p + "branch2.6", bf)
        inp = oup
    conv("conv5.0", 1024, 464, 1)
    bn("conv5.1", 1024)
    sd["fc.weight"] = (rng.standard_normal((num_classes, 1024)) * (1.0 / 1024) ** 0.5).astype(np.float32)
    sd["fc.bias"] = (rng.standard_normal(num_classes) * 0.1).astype(np.float32)
    return sd


def _rand_helpers(rng, sd, gain):
    def conv(name, co, ci, k):
        sd[name + ".weight"] = (rng.standard_normal((co, ci, k, k)) * (gain / (ci * k * k)) ** 0.5).astype(np.float32)

    def bn(name, c):
        sd[name + ".weight"] = rng.uniform(0.75, 1.25, c).astype(np.float32)
        sd[name + ".bias"] = (rng.standard_normal(c) * 0.1).astype(np.float32)
        sd[name + ".running_mean"] = (rng.standard_normal(c) * 0.1).astype(np.float32)
        sd[name + ".running_var"] = rng.uniform(0.75, 1.25, c).astype(np.float32)
    return conv, bn


def random_mobilenetv2_state(num_classes: int, seed: int = 0) -> Dict[str, np.ndarray]:
    """Seeded random mobilenet_v2 state_dict with torchvision's key names and shapes (features.i.conv.*, classifier.1.*)."""
    rng = np.random.default_rng(seed)
    sd: Dict[str, np.ndarray] = {}
    conv, bn = _rand_helpers(rng, sd, 2.0)
    conv("features.0.0", 32, 3, 3); bn("features.0.1", 32)
    inp, idx = 32, 1
    for t, c, n, s in ([1, 16, 1, 1], [6, 24, 2, 2], [6, 32, 3, 2], [6, 64, 4, 2], [6, 96, 3, 1], [6, 160, 3, 2], [6, 320, 1, 1]):
        for _ in range(n):
            p, hid, li = f"features.{idx}.conv.", inp * t, 0
            if t != 1:
                conv(p + "0.0", hid, inp, 1); bn(p + "0.1", hid)
                li = 1
            conv(p + f"{li}.0", hid, 1, 3); bn(p + f"{li}.1", hid)
            conv(p + f"{li + 1}", c, hid, 1); bn(p + f"{li + 2}", c)
            inp = c
            idx += 1
    conv("features.18.0", 1280, inp, 1); bn("features.18.1", 1280)
    sd["classifier.1.weight"] = (rng.standard_normal((num_classes, 1280)) * (1.0 / 1280) ** 0.5).astype(np.float32)
    sd["classifier.1.bias"] = (rng.standard_normal(num_classes) * 0.1).astype(np.float32)
    return sd


def random_efficientnet_state(num_classes: int, seed: int = 0) -> Dict[str, np.ndarray]:
    """Seeded random efficientnet_b0 state_dict with torchvision's key names and shapes (features.s.r.block.*, classifier.1.*)."""
    rng = np.random.default_rng(seed)
    sd: Dict[str, np.ndarray] = {}
    conv, bn = _rand_helpers(rng, sd, 2.0)
    conv("features.0.0", 32, 3, 3); bn("features.0.1", 32)
    cfg = [(1, 3, 1, 32, 16, 1), (6, 3, 2, 16, 24, 2), (6, 5, 2, 24, 40, 2), (6, 3, 2, 40, 80, 3), (6, 5, 1, 80, 112, 3),
           (6, 5, 2, 112, 192, 4), (6, 3, 1, 192, 320, 1)]
    for si, (e, k, s, i, o, n) in enumerate(cfg):
        for r in range(n):
            inp = i if r == 0 else o
            p, exp, li = f"features.{si + 1}.{r}.block.", inp * e, 0
            if e != 1:
                conv(p + "0.0", exp, inp, 1); bn(p + "0.1", exp)
                li = 1
            conv(p + f"{li}.0", exp, 1, k); bn(p + f"{li}.1", exp)
            sq = max(1, inp // 4)
            conv(p + f"{li + 1}.fc1", sq, exp, 1); sd[p + f"{li + 1}.fc1.bias"] = (rng.standard_normal(sq) * 0.1).astype(np.float32)
            conv(p + f"{li + 1}.fc2", exp, sq, 1); sd[p + f"{li + 1}.fc2.bias"] = (rng.standard_normal(exp) * 0.1).astype(np.float32)
            conv(p + f"{li + 2}.0", o, exp, 1); bn(p + f"{li + 2}.1", o)
    conv("features.8.0", 1280, 320, 1); bn("features.8.1", 1280)
    sd["classifier.1.weight"] = (rng.standard_normal((num_classes, 1280)) * (1.0 / 1280) ** 0.5).astype(np.float32)
    sd["classifier.1.bias"] = (rng.standard_normal(num_classes) * 0.1).astype(np.float32)
    return sd


def random_resnet18_state(num_classes: int, seed: int = 0) -> Dict[str, np.ndarray]:
    """Seeded random resnet18 state_dict with torchvision's key names and shapes."""
    rng = np.random.default_rng(seed)
    sd: Dict[str, np.ndarray] = {}

    def conv(name, co, ci, k):
        sd[name + ".weight"] = (rng.standard_normal((co, ci, k, k)) * (1.4 / (ci * k * k)) ** 0.5).astype(np.float32)

    def bn(name, c):
        sd[name + ".weight"] = rng.uniform(0.75, 1.25, c).astype(np.float32)
        sd[name + ".bias"] = (rng.standard_normal(c) * 0.1).astype(np.float32)
        sd[name + ".running_mean"] = (rng.standard_normal(c) * 0.1).astype(np.float32)
        sd[name + ".running_var"] = rng.uniform(0.75, 1.25, c).astype(np.float32)

    conv("conv1", 64, 3, 7)
    bn("bn1", 64)
    cin = 64
    for L, width in enumerate((64, 128, 256, 512)):
        for r in range(2):
            p = f"layer{L + 1}.{r}"
            stride = 2 if (L > 0 and r == 0) else 1
            conv(p + ".conv1", width, cin, 3); bn(p + ".bn1", width)
            conv(p + ".conv2", width, width, 3); bn(p + ".bn2", width)
            if stride != 1 or cin != width:
                conv(p + ".downsample.0", width, cin, 1); bn(p + ".downsample.1", width)
            cin = width
    sd["fc.weight"] = (rng.standard_normal((num_classes, 512)) * (1.0 / 512) ** 0.5).astype(np.float32)
    sd["fc.bias"] = (rng.standard_normal(num_classes) * 0.1).astype(np.float32)
    return sd


class PyTorchClassifier:
    """e2e.py:350-396 (name kept for drop-in use; the model runs in HIP, not torch)."""

    def __init__(self, model_path: str, arch: str, num_classes: int = 58, input_size: int = 64, device: str = "cpu", *,
                 precision: str = "fp16", max_rois: int = 1024, conv_impl: int = 0, _engine: Optional[Engine] = None):
        if arch not in CLS_ARCHS:
            raise ValueError(f"Unknown architecture: {arch}")   # as build_classifier, e2e.py:334
        self.input_size = input_size
        self.num_classes = num_classes
        self.arch = arch
        print(f"[HIP Classifier] Loading {arch} model...")
        print(f"  Model: {model_path}")
        self.engine = _engine or Engine(precision=precision, max_batch=1, max_det=max_rois, num_classes=num_classes,
                                        cls_input=input_size, max_rois=max_rois, cls_arch=arch, conv_impl=conv_impl)
        if self.engine.cls_arch != arch:
            raise ValueError(f"the engine was created for {self.engine.cls_arch}, not {arch}")
        sd, self.weights_loaded = load_classifier_state(model_path, num_classes, arch)
        self.state_dict = sd   # (HybridPipeline's upload lanes load the same weights)
        self.engine.load_classifier(sd)
        print(f"  Architecture: {arch}")
        print(f"  Input size: {input_size}x{input_size}")
        print(f"  Num classes: {num_classes}")

    def predict_batch(self, images: List[np.ndarray]) -> Tuple[np.ndarray, np.ndarray]:
        if len(images) == 0:
            return np.array([]), np.array([])  # e2e.py:380-381
        ids, probs = self.engine.classify(images)
        return ids, probs


class HybridPipeline:
    """e2e.py:399-531 with one extra method, ``run_batch``.  Detector and classifier share one
    device handle, so detections never leave the GPU between the two stages."""

    def __init__(self, detector_param: str, detector_bin: str, classifier_path: str, classifier_arch: str,
                 num_classes: int = 58, det_input_size: int = 640, cls_input_size: int = 64, use_gpu_detector: bool = False,
                 detector_threads: int = 4, classifier_device: str = "cpu", batch_size: int = 8, *, precision: str = "fp16",
                 max_batch: int = 1, max_det: int = 300, device: int = 0, max_rois: int = 0, numerics: str = "e2e"):
        print("\n" + "=" * 70)
        print("HYBRID PIPELINE: HIP Detector + HIP Classifier (MI355X)")
        print("=" * 70)
        self.engine = Engine(precision=precision, max_batch=max_batch, max_det=max_det, num_classes=num_classes,
                             det_input=det_input_size, cls_input=cls_input_size, device=device, max_rois=max_rois,
                             numerics=numerics, cls_arch=classifier_arch if classifier_arch in CLS_ARCHS else "shufflenetv2")
        self.detector = NCNNDetector(detector_param, detector_bin, det_input_size, use_gpu_detector, detector_threads,
                                     _engine=self.engine)
        self.classifier = PyTorchClassifier(classifier_path, classifier_arch, num_classes, cls_input_size, classifier_device,
                                            _engine=self.engine)
        self.batch_size = batch_size  # kept for signature parity: all ROIs of a call are classified in one pass
        # ---- upload lanes (an experiment kept behind LITEPI_DROPIN_LANES=<n>, off by default: measured SLOWER, 3.2-3.6 ms per
        # 64-frame call against 2.7-2.9 for one handle -- four 16-frame passes cost 1.4 ms of kernels instead of 0.9 and four
        # sets of copy workers fight over the cores; the overlap of upload and kernels lives inside lp_run_batch instead, which
        # walks a large batch in chunks, api.cpp run_batch_chunked).  Lanes are further handles of max_batch / lanes images each
        # (own stream, staging buffer, copy workers, the same models); a call's images are dealt to them in contiguous slices and
        # every slice runs lp_run_batch on its lane from a worker thread (ctypes drops the GIL).  Results do not depend on the
        # split (tests/test_gpu_device_path.py compares both).
        self._lanes: List[Engine] = []
        self._lane_pool = None
        n_lanes = int(os.environ.get("LITEPI_DROPIN_LANES", "1"))
        if n_lanes > 1 and max_batch >= 2 * n_lanes:
            from concurrent.futures import ThreadPoolExecutor
            lane_cap = -(-max_batch // n_lanes)
            lane_rois = max_rois if max_rois <= 0 else -(-max_rois // n_lanes)
            for _ in range(n_lanes):
                e = Engine(precision=precision, max_batch=lane_cap, max_det=max_det, num_classes=num_classes, det_input=det_input_size,
                           cls_input=cls_input_size, device=device, max_rois=lane_rois, numerics=numerics, cls_arch=self.engine.cls_arch)
                e.load_detector(detector_param, detector_bin)
                e.load_classifier(self.classifier.state_dict)
                self._lanes.append(e)
            self._lane_pool = ThreadPoolExecutor(max_workers=n_lanes, thread_name_prefix="litepi-lane")
            self._lane_cap = lane_cap
        print("\nPipeline ready!")
        print("=" * 70 + "\n")

    def close(self) -> None:
        if self._lane_pool is not None:
            self._lane_pool.shutdown(wait=True)
            self._lane_pool = None
        for e in self._lanes:
            e.close()
        self._lanes = []
        self.engine.close()

    def _run_lanes(self, images, conf, iou, min_area):
        """run_batch's engine call over the upload lanes: contiguous slices, results concatenated in image order."""
        B, n = len(images), len(self._lanes)
        per = -(-B // n)
        slices = [(k, k * per, min(B, (k + 1) * per)) for k in range(n) if k * per < B]

        def one(arg):
            k, lo, hi = arg
            e = self._lanes[k]
            d, c, nd, t = e.run_batch(images[lo:hi], conf, iou, min_area)
            return d, c, nd, t, e.last_det_conf_avg

        parts = list(self._lane_pool.map(one, slices))
        dets = np.concatenate([p[0] for p in parts], 0)
        counts = np.concatenate([p[1] for p in parts])
        num_det = np.concatenate([p[2] for p in parts])
        timing = LpTiming()
        for f in ("t_detection", "t_roi_extract", "t_classification", "t_total"):   # the lanes run side by side: the longest one
            setattr(timing, f, max(getattr(p[3], f) for p in parts))
        self.engine.last_det_conf_avg = np.concatenate([p[4] for p in parts])
        return dets, counts, num_det, timing

    def run_batch(self, images: Sequence[np.ndarray], conf_threshold: float = 0.5, iou_threshold: float = 0.45,
                  min_area: int = 100) -> List[Tuple[List[Dict], PipelineMetrics]]:
        t0 = time.perf_counter()
        try:
            if self._lanes and len(images) >= 32 and len(images) <= self._lane_cap * len(self._lanes):
                dets, counts, num_det, timing = self._run_lanes(list(images), conf_threshold, iou_threshold, min_area)
            else:
                dets, counts, num_det, timing = self.engine.run_batch(images, conf_threshold, iou_threshold, min_area)
        except _ffi.LitepiError as e:
            if e.code != _ffi.LP_ERR_HIP:  # misuse / capacity errors are raised, only an engine failure yields "nothing found"
                raise
            print(f"[HIP Pipeline] engine failure, returning no detections: {e}")
            return [([], PipelineMetrics()) for _ in images]
        wall_ms = (time.perf_counter() - t0) * 1000.0   # t_total ends here; system metrics are sampled after it (e2e.py:505-516)
        conf_avg = self.engine.last_det_conf_avg
        sysm = _system_metrics()
        B = len(images)
        # every used record of the batch decoded in ONE pass (a Python loop over structured-array fields cost 5 us per
        # detection: 2 ms for the ~400 results of a 64-image batch): the float box is truncated to int like the reference's
        # tuple(box.astype(int)) (e2e.py:522), float32 fields become Python floats of the same value
        cnt_a = np.asarray(counts[:B], dtype=np.int64)
        cnt = cnt_a.tolist()
        used = dets[:B][np.arange(dets.shape[1])[None, :] < cnt_a[:, None]]
        boxes = list(map(tuple, np.stack([used["x1"], used["y1"], used["x2"], used["y2"]], 1).astype(int).tolist()))
        det_cls, cls_cls = used["det_class"].tolist(), used["cls_class"].tolist()
        det_cf, cls_cf = used["det_conf"].astype(np.float64).tolist(), used["cls_conf"].astype(np.float64).tolist()
        # per-image mean classifier confidence in one pass (a NumPy slice + mean per image cost 5 us each): float64 sums of the
        # float32 probabilities, as the reference's np.mean over Python floats (e2e.py:500-501)
        img_of = np.repeat(np.arange(B), cnt_a)
        ok = used["cls_class"] >= 0
        n_ok = np.bincount(img_of, weights=ok, minlength=B)
        s_ok = np.bincount(img_of, weights=np.where(ok, used["cls_conf"].astype(np.float64), 0.0), minlength=B)
        cls_avg = (s_ok / np.maximum(n_ok, 1.0)).tolist()
        n_ok = n_ok.tolist()
        num_det_l = np.asarray(num_det[:B]).tolist()
        conf_avg_l = np.asarray(conf_avg[:B], dtype=np.float64).tolist()
        t_det, t_roi, t_cls, t_tot = timing.t_detection / B, timing.t_roi_extract / B, timing.t_classification / B, wall_ms / B
        fps = 1000.0 / t_tot if t_tot > 0 else 0
        cpu_p, mem_mb, temp = sysm
        out = []
        pos = 0
        for i in range(B):
            # device stage times are per batch call; report the per-image share like a sequential loop would
            nd = num_det_l[i]   # counted BEFORE the min-area filter (e2e.py:454)
            n = cnt[i]
            # det_confidence_avg: averaged over ALL detector boxes, before the min-area filter (e2e.py:456-457)
            m = PipelineMetrics(t_detection=t_det, t_roi_extract=t_roi, t_classification=t_cls, t_total=t_tot, fps=fps, num_detections=nd,
                                det_confidence_avg=conf_avg_l[i] if nd else 0.0, cls_confidence_avg=cls_avg[i] if n_ok[i] else 0.0,
                                cpu_percent=cpu_p, memory_mb=mem_mb, temperature=temp)
            results = []
            if n:
                td, tc = t_det / n, t_cls / n
                e = pos + n
                results = [{"bbox": bb, "det_class": dc, "det_conf": df, "cls_class": cc, "cls_conf": cf, "time_det": td, "time_cls": tc}
                           for bb, dc, df, cc, cf in zip(boxes[pos:e], det_cls[pos:e], det_cf[pos:e], cls_cls[pos:e], cls_cf[pos:e])]
                pos = e
            out.append((results, m))
        return out

    def run(self, image: np.ndarray, conf_threshold: float = 0.5, iou_threshold: float = 0.45,
            min_area: int = 100) -> Tuple[List[Dict], PipelineMetrics]:
        return self.run_batch([image], conf_threshold, iou_threshold, min_area)[0]
