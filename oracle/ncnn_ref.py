"""Oracle: NCNN ``.param``/``.bin`` reader + fp32 torch-CPU graph interpreter.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Restates what ``ncnn.Net.load_param/load_model`` + ``Extractor.extract`` do for
the detector graphs the reference ships
(``src/vntsr/convert/model/yolo_plus/yolo_plus_ncnn_model/model.ncnn.param:3-208``,
call sites ``src/tt100k/pipeline/e2e.py:209-216,305-307``).  NCNN itself
("1.0.20250916", paper §4.2) is not vendored in the reference and not installed
here, so the op semantics below follow NCNN's published layer definitions for
the 14 layer types these graphs use.  Forward-output parity is UNPINNED: the
reference holds no expected outputs for this graph; the weights this reader
extracts are cross-checked against the ONNX initializers of the same export
(``oracle/onnx_init.py``, ``tests/test_oracle_detector.py``).

Tensors are torch fp32 in NCNN dim order with an optional leading batch dim:
3-D blobs are ``[B, c, h, w]``, 2-D ``[B, h, w]``, 1-D ``[B, w]``.  MemoryData
blobs carry no batch dim and broadcast.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

MAGIC = "7767517"


@dataclass
class Layer:
    type: str
    name: str
    inputs: List[str]
    outputs: List[str]
    params: Dict[int, object] = field(default_factory=dict)
    weight: Optional[np.ndarray] = None
    bias: Optional[np.ndarray] = None
    data: Optional[np.ndarray] = None  # MemoryData payload


def _parse_value(v: str):
    try:
        return int(v)
    except ValueError:
        return float(v)


def parse_param(path: str) -> List[Layer]:
    """Parse the text graph (format: SURVEY Appendix C)."""
    with open(path, "r") as f:
        lines = [ln.strip() for ln in f if ln.strip()]
    if lines[0] != MAGIC:
        raise ValueError(f"bad ncnn magic {lines[0]!r}")
    n_layers, _n_blobs = (int(x) for x in lines[1].split())
    layers = []
    for ln in lines[2:]:
        tok = ln.split()
        ltype, name, n_in, n_out = tok[0], tok[1], int(tok[2]), int(tok[3])
        ins = tok[4:4 + n_in]
        outs = tok[4 + n_in:4 + n_in + n_out]
        params: Dict[int, object] = {}
        for kv in tok[4 + n_in + n_out:]:
            k, v = kv.split("=", 1)
            k = int(k)
            if k <= -23300:  # array parameter: -23300-id=count,v0,v1,...
                vals = v.split(",")
                params[-k - 23300] = [_parse_value(x) for x in vals[1:1 + int(vals[0])]]
            else:
                params[k] = _parse_value(v)
        layers.append(Layer(ltype, name, ins, outs, params))
    if len(layers) != n_layers:
        raise ValueError(f"layer count {len(layers)} != header {n_layers}")
    return layers


def load_bin(layers: List[Layer], path: str) -> None:
    """Attach weights: read strictly in layer order (Appendix C)."""
    with open(path, "rb") as f:
        blob = f.read()
    off = 0
    for L in layers:
        if L.type in ("Convolution", "ConvolutionDepthWise"):
            out_ch = int(L.params[0])
            kw = int(L.params.get(1, 1))
            kh = int(L.params.get(11, kw))
            wcount = int(L.params[6])
            (flag,) = struct.unpack_from("<I", blob, off)
            off += 4
            if flag != 0:
                raise ValueError(f"{L.name}: unsupported weight storage flag {flag:#x}")
            in_ch = wcount // (out_ch * kw * kh)  # per group for ConvolutionDepthWise (1 for a depthwise conv)
            L.weight = np.frombuffer(blob, "<f4", wcount, off).reshape(out_ch, in_ch, kh, kw).copy()
            off += 4 * wcount
            if int(L.params.get(5, 0)):
                L.bias = np.frombuffer(blob, "<f4", out_ch, off).copy()
                off += 4 * out_ch
        elif L.type == "MemoryData":
            w = int(L.params.get(0, 0))
            h = int(L.params.get(1, 0))
            c = int(L.params.get(2, 0))
            shape = [d for d in (c, h, w) if d > 0]
            n = int(np.prod(shape))
            L.data = np.frombuffer(blob, "<f4", n, off).reshape(shape).copy()
            off += 4 * n
    if off != len(blob):
        raise ValueError(f"bin size mismatch: consumed {off} of {len(blob)} bytes")


def load_model(param_path: str, bin_path: str) -> List[Layer]:
    layers = parse_param(param_path)
    load_bin(layers, bin_path)
    return layers


def _axis(t: torch.Tensor, ncnn_axis: int, batched: bool) -> int:
    """NCNN positive axis counts from the outermost non-batch dim."""
    return ncnn_axis + (1 if batched else 0)


@torch.no_grad()
def run_graph(layers: List[Layer], x: torch.Tensor, keep: Optional[List[str]] = None,
              until: Optional[str] = None) -> Dict[str, torch.Tensor]:
    """Interpret the graph on ``x`` = fp32 ``[B,3,H,W]`` RGB in [0,1].

    Returns ``{blob_name: tensor}`` for the blobs named in ``keep`` (default:
    the last layer's outputs).  Every blob is kept while running; graphs here
    are small.
    """
    blobs: Dict[str, torch.Tensor] = {}
    batched: Dict[str, bool] = {}
    for L in layers:
        p = L.params
        t = L.type
        if t == "Input":
            blobs[L.outputs[0]] = x
            batched[L.outputs[0]] = True
        elif t == "MemoryData":
            blobs[L.outputs[0]] = torch.from_numpy(L.data)
            batched[L.outputs[0]] = False
        elif t in ("Convolution", "ConvolutionDepthWise"):
            a = blobs[L.inputs[0]]
            squeeze = False
            if a.dim() == 3:  # [B,h,w] never happens here; keep conv strictly 4-D
                raise ValueError("Convolution on non-3D blob")
            stride = (int(p.get(13, p.get(3, 1))), int(p.get(3, 1)))
            pad = (int(p.get(14, p.get(4, 0))), int(p.get(4, 0)))
            dil = (int(p.get(12, p.get(2, 1))), int(p.get(2, 1)))
            squeeze = a.dim() == 3  # a blob that went through 3-D Reshape/MatMul (YOLO11 attention) has no batch dim
            if squeeze:
                raise ValueError("Convolution on non-3D blob")
            y = F.conv2d(a, torch.from_numpy(L.weight),
                         torch.from_numpy(L.bias) if L.bias is not None else None,
                         stride=stride, padding=pad, dilation=dil, groups=int(p.get(7, 1)) if t == "ConvolutionDepthWise" else 1)
            blobs[L.outputs[0]] = y
            batched[L.outputs[0]] = True
        elif t == "Swish":
            a = blobs[L.inputs[0]]
            blobs[L.outputs[0]] = a * torch.sigmoid(a)
            batched[L.outputs[0]] = batched[L.inputs[0]]
        elif t == "Sigmoid":
            blobs[L.outputs[0]] = torch.sigmoid(blobs[L.inputs[0]])
            batched[L.outputs[0]] = batched[L.inputs[0]]
        elif t == "Split":
            for o in L.outputs:
                blobs[o] = blobs[L.inputs[0]]
                batched[o] = batched[L.inputs[0]]
        elif t == "Slice":
            a = blobs[L.inputs[0]]
            b = batched[L.inputs[0]]
            ax = _axis(a, int(p.get(1, 0)), b)
            sizes = list(p[0])
            total = a.shape[ax]
            known = sum(s for s in sizes if s != -233)
            n_auto = sum(1 for s in sizes if s == -233)
            sizes = [s if s != -233 else (total - known) // n_auto for s in sizes]
            for o, part in zip(L.outputs, torch.split(a, sizes, dim=ax)):
                blobs[o] = part
                batched[o] = b
        elif t == "Concat":
            ins = [blobs[i] for i in L.inputs]
            b = any(batched[i] for i in L.inputs)
            ax = _axis(ins[0], int(p.get(0, 0)), b)
            blobs[L.outputs[0]] = torch.cat(ins, dim=ax)
            batched[L.outputs[0]] = b
        elif t == "BinaryOp":
            op = int(p.get(0, 0))
            a = blobs[L.inputs[0]]
            if int(p.get(1, 0)):
                bb = torch.tensor(float(p.get(2, 0.0)), dtype=torch.float32)
                b = batched[L.inputs[0]]
            else:
                bb = blobs[L.inputs[1]]
                b = batched[L.inputs[0]] or batched[L.inputs[1]]
            y = {0: torch.add, 1: torch.sub, 2: torch.mul, 3: torch.div}[op](a, bb)
            blobs[L.outputs[0]] = y
            batched[L.outputs[0]] = b
        elif t == "Pooling":
            if int(p.get(0, 0)) != 0:
                raise ValueError("only max pooling is used by these graphs")
            k = (int(p.get(11, p.get(1))), int(p.get(1)))
            s = (int(p.get(12, p.get(2, 1))), int(p.get(2, 1)))
            pd = (int(p.get(13, p.get(3, 0))), int(p.get(3, 0)))
            blobs[L.outputs[0]] = F.max_pool2d(blobs[L.inputs[0]], k, s, pd)
            batched[L.outputs[0]] = True
        elif t == "Interp":
            if int(p.get(0, 0)) != 1:
                raise ValueError("only nearest Interp is used by these graphs")
            sh, sw = float(p.get(1, 1.0)), float(p.get(2, 1.0))
            a = blobs[L.inputs[0]]
            blobs[L.outputs[0]] = F.interpolate(a, scale_factor=(sh, sw), mode="nearest")
            batched[L.outputs[0]] = True
        elif t == "Reshape":
            a = blobs[L.inputs[0]]
            b = batched[L.inputs[0]]
            w, h, c = int(p.get(0, 0)), int(p.get(1, 0)), int(p.get(2, 0))
            shape = [d for d in (c, h, w) if d != 0]
            blobs[L.outputs[0]] = a.reshape(([a.shape[0]] if b else []) + shape)
            batched[L.outputs[0]] = b
        elif t == "Permute":
            a = blobs[L.inputs[0]]
            b = batched[L.inputs[0]]
            order = int(p.get(0, 0))
            nd = a.dim() - (1 if b else 0)
            if nd != 3 or order not in (1, 2):
                raise ValueError("only 3-D Permute order_type 1 (h,w,c) / 2 (w,c,h) is used by these graphs")
            o = 1 if b else 0
            if order == 2:   # new (w,h,c) = old (w,c,h): [c,h,w] -> [h,c,w]
                blobs[L.outputs[0]] = a.transpose(o, o + 1).contiguous()
            else:            # new (w,h,c) = old (h,w,c): [c,h,w] -> [c,w,h]   (YOLO11 attention: q^T)
                blobs[L.outputs[0]] = a.transpose(o + 1, o + 2).contiguous()
            batched[L.outputs[0]] = b
        elif t == "MatMul":
            a, bb = blobs[L.inputs[0]], blobs[L.inputs[1]]
            if int(p.get(0, 0)):
                bb = bb.transpose(-1, -2)
            blobs[L.outputs[0]] = torch.matmul(a, bb)
            batched[L.outputs[0]] = batched[L.inputs[0]] or batched[L.inputs[1]]
        elif t == "Softmax":
            a = blobs[L.inputs[0]]
            b = batched[L.inputs[0]]
            blobs[L.outputs[0]] = torch.softmax(a, dim=_axis(a, int(p.get(0, 0)), b))
            batched[L.outputs[0]] = b
        else:
            raise ValueError(f"unsupported NCNN layer type {t}")
        if until is not None and until in L.outputs:
            break
    names = keep if keep is not None else layers[-1].outputs
    return {n: blobs[n] for n in names if n in blobs}


def conv_layers(layers: List[Layer]) -> List[Layer]:
    return [L for L in layers if L.type == "Convolution"]


def conv_macs(layers: List[Layer], h: int = 640, w: int = 640) -> int:
    """MACs of the spatial convs (DFL conv_65 excluded)."""
    probe = torch.zeros(1, 3, h, w)
    names = [L.outputs[0] for L in conv_layers(layers)]
    outs = run_graph(layers, probe, keep=names)
    total = 0
    for L in conv_layers(layers):
        if L.bias is None:  # DFL arange conv
            continue
        y = outs[L.outputs[0]]
        total += int(np.prod(L.weight.shape)) * y.shape[-1] * y.shape[-2]
    return total
