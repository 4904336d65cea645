"""ctypes binding of liblitepi_hip.so (C-ABI: include/litepi.h).

The product path has no CPU fallback: if the library is missing or no gfx950 device is
usable, loading / ``lp_create`` fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblitepi_hip.so")

LP_FP32, LP_FP16 = 0, 1
LP_OK, LP_ERR_ARG, LP_ERR_IO, LP_ERR_GRAPH, LP_ERR_HIP, LP_ERR_STATE, LP_ERR_NODEVICE = 0, -1, -2, -3, -4, -5, -6


class LpConfig(C.Structure):
    _fields_ = [("device", C.c_int), ("precision", C.c_int), ("max_batch", C.c_int), ("max_det", C.c_int),
                ("num_classes", C.c_int), ("det_input", C.c_int), ("cls_input", C.c_int), ("max_rois", C.c_int),
                ("conv_impl", C.c_int), ("numerics", C.c_int), ("cls_arch", C.c_int), ("reserved", C.c_int * 5)]


class LpDet(C.Structure):
    _fields_ = [("x1", C.c_float), ("y1", C.c_float), ("x2", C.c_float), ("y2", C.c_float), ("det_conf", C.c_float),
                ("det_class", C.c_int32), ("cls_class", C.c_int32), ("cls_conf", C.c_float)]


class LpTiming(C.Structure):
    _fields_ = [("t_detection", C.c_float), ("t_roi_extract", C.c_float), ("t_classification", C.c_float),
                ("t_total", C.c_float)]


class LpKernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("layer", C.c_char * 32), ("ms", C.c_float), ("flops", C.c_double),
                ("bytes", C.c_double)]


# numpy view of lp_det records
DET_DTYPE = [("x1", "<f4"), ("y1", "<f4"), ("x2", "<f4"), ("y2", "<f4"), ("det_conf", "<f4"), ("det_class", "<i4"),
             ("cls_class", "<i4"), ("cls_conf", "<f4")]

SYMBOLS = [
    "lp_last_error", "lp_version", "lp_default_config", "lp_create", "lp_destroy", "lp_load_detector_ncnn",
    "lp_load_classifier_tensors", "lp_detect_raw", "lp_detect", "lp_run_batch", "lp_run_batch_device", "lp_classify",
    "lp_set_stream", "lp_synchronize", "lp_profile_next", "lp_profile_read", "lp_detector_info", "lp_debug_blob",
    "lp_test_conv", "lp_test_postprocess", "lp_test_nms_boxes", "lp_test_roi_resize", "lp_test_letterbox", "lp_roi_overflow",
    "lp_comm_unique_id", "lp_comm_init", "lp_gather", "lp_comm_destroy",
]
ABI_VERSION = 310   # include/litepi.h LP_ABI_VERSION: a library built from another header is refused (load_library)

_lib: Optional[C.CDLL] = None


def load_library(path: Optional[str] = None) -> C.CDLL:
    """dlopen the library and declare prototypes.  Raises ImportError when it has not been built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise ImportError(f"{p} not found: build it with `python yolo-litepi_amd/build.py` "
                          "(hipcc --offload-arch=gfx950); litepi has no CPU fallback")
    lib = C.CDLL(p)
    vp, ip, fp = C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_float)
    u8pp = C.POINTER(C.c_void_p)
    lib.lp_last_error.restype = C.c_char_p
    lib.lp_version.restype = C.c_int
    lib.lp_default_config.argtypes = [C.POINTER(LpConfig)]
    lib.lp_default_config.restype = None
    lib.lp_create.argtypes = [C.POINTER(LpConfig), C.POINTER(vp)]
    lib.lp_destroy.argtypes = [vp]
    lib.lp_destroy.restype = None
    lib.lp_load_detector_ncnn.argtypes = [vp, C.c_char_p, C.c_char_p]
    lib.lp_load_classifier_tensors.argtypes = [vp, C.c_int, C.POINTER(C.c_char_p), C.POINTER(vp), C.POINTER(vp), ip]
    lib.lp_detect_raw.argtypes = [vp, vp, C.c_int, vp]
    lib.lp_detect.argtypes = [vp, u8pp, ip, ip, C.c_int, C.c_float, C.c_float, vp, ip]
    lib.lp_run_batch.argtypes = [vp, u8pp, ip, ip, C.c_int, C.c_float, C.c_float, C.c_int, vp, ip, ip, fp, C.POINTER(LpTiming)]
    lib.lp_run_batch_device.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, vp, vp]
    lib.lp_classify.argtypes = [vp, u8pp, ip, ip, C.c_int, ip, fp]
    lib.lp_set_stream.argtypes = [vp, vp]
    lib.lp_synchronize.argtypes = [vp]
    lib.lp_profile_next.argtypes = [vp, C.c_int]
    lib.lp_profile_read.argtypes = [vp, C.POINTER(LpKernelTime), C.c_int, ip]
    lib.lp_detector_info.argtypes = [vp, ip, ip, ip, C.POINTER(C.c_double)]
    lib.lp_debug_blob.argtypes = [vp, C.c_char_p, fp, C.c_int64, ip, ip, ip]
    lib.lp_test_conv.argtypes = [vp, C.c_int, fp, C.c_int, C.c_int, C.c_int, C.c_int, fp, fp, C.c_int, C.c_int, C.c_int,
                                 C.c_int, fp, fp]
    lib.lp_test_postprocess.argtypes = [vp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float,
                                        C.c_float, C.c_float, C.c_int, C.c_int, vp, ip, ip, ip]
    lib.lp_test_nms_boxes.argtypes = [vp, fp, fp, ip, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, vp, ip, ip, ip]
    lib.lp_test_roi_resize.argtypes = [vp, u8pp, ip, ip, C.c_int, vp]
    lib.lp_test_letterbox.argtypes = [vp, vp, C.c_int, C.c_int, vp, fp, fp, fp]
    lib.lp_roi_overflow.argtypes = [vp, ip, ip]
    lib.lp_comm_unique_id.argtypes = [vp]
    lib.lp_comm_init.argtypes = [vp, vp, C.c_int, C.c_int]
    lib.lp_gather.argtypes = [vp, vp, C.c_size_t, vp, C.c_int]
    lib.lp_comm_destroy.argtypes = [vp]
    for s in SYMBOLS:
        if s not in ("lp_last_error", "lp_default_config", "lp_destroy"):
            getattr(lib, s).restype = C.c_int
    got = lib.lp_version()
    if got != ABI_VERSION:
        raise ImportError(f"{p} has ABI version {got}, these bindings need {ABI_VERSION}: rebuild with `python yolo-litepi_amd/build.py`")
    if path is None:
        _lib = lib
    return lib


class LitepiError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"litepi error {code}: {msg}")
        self.code = code


def check(lib: C.CDLL, rc: int) -> None:
    if rc != 0:
        raise LitepiError(rc, lib.lp_last_error().decode(errors="replace"))
