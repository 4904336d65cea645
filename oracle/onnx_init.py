"""Oracle helper: read the fp32 initializers out of an ONNX file without `onnx`.

TEST INFRASTRUCTURE ONLY.  Used to cross-check that ``oracle/ncnn_ref.load_bin``
extracts the same tensors from ``model.ncnn.bin`` as the ONNX export of the
same checkpoint holds (``src/vntsr/convert/model/yolo_plus/yolo_plus.onnx``;
SURVEY §8(c)).  Minimal protobuf wire-format walker: ModelProto.graph (field 7)
-> GraphProto.initializer (field 5) -> TensorProto{dims=1, data_type=2,
float_data=4, name=8, raw_data=9}.
"""
from __future__ import annotations

from typing import Dict, Iterator, Tuple

import numpy as np


def _varint(buf: bytes, pos: int) -> Tuple[int, int]:
    val = 0
    shift = 0
    while True:
        b = buf[pos]
        pos += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, pos
        shift += 7


def _fields(buf: bytes) -> Iterator[Tuple[int, int, object]]:
    pos = 0
    n = len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = buf[pos:pos + 8]
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            v = buf[pos:pos + 4]
            pos += 4
        else:
            raise ValueError(f"unsupported wire type {wt}")
        yield fno, wt, v


def read_initializers(path: str) -> Dict[str, np.ndarray]:
    with open(path, "rb") as f:
        model = f.read()
    out: Dict[str, np.ndarray] = {}
    for fno, wt, graph in _fields(model):
        if fno != 7 or wt != 2:
            continue
        for gno, gwt, tensor in _fields(graph):
            if gno != 5 or gwt != 2:
                continue
            dims, dtype, name, raw, fdata = [], None, None, None, []
            for tno, twt, v in _fields(tensor):
                if tno == 1:
                    if twt == 0:
                        dims.append(v)
                    else:  # packed
                        p = 0
                        while p < len(v):
                            d, p = _varint(v, p)
                            dims.append(d)
                elif tno == 2:
                    dtype = v
                elif tno == 8:
                    name = v.decode()
                elif tno == 9:
                    raw = v
                elif tno == 4:
                    if twt == 2:
                        fdata.append(np.frombuffer(v, "<f4"))
                    else:
                        fdata.append(np.frombuffer(v, "<f4", 1))
            if dtype == 1:  # FLOAT
                arr = np.frombuffer(raw, "<f4") if raw is not None else np.concatenate(fdata)
                out[name] = arr.reshape(dims).copy()
            elif dtype == 7 and raw is not None:  # INT64 (shape constants)
                out[name] = np.frombuffer(raw, "<i8").reshape(dims).copy()
    return out
