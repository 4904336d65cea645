"""Oracle: fp32 torch-CPU interpreter of an ONNX detector export (no `onnx`, no `onnxruntime`).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

The reference holds TWO independent descriptions of the v1 detector: the NCNN graph
(``src/vntsr/convert/model/yolo_plus/yolo_plus_ncnn_model/model.ncnn.param``, restated by
``oracle/ncnn_ref.py``) and the ONNX export of the same checkpoint
(``src/vntsr/convert/model/yolo_plus/yolo_plus.onnx``, opset 12: the path its notebooks run through
ONNX Runtime, ``evaluation_tsd_single_img.ipynb:342,379``).  This module walks the ONNX node list with the
operator semantics of the ONNX specification (opset 12) for the 14 operator types the export uses, so that
``tests/test_oracle_cpu.py`` can assert that both readings give the same ``out0`` on the same input: a
cross-check of ``ncnn_ref``'s reading of Slice / Interp / Pooling pad mode / Permute / the DFL ordering
against a second, spec-defined description.  It pins no output of the reference (none exists): forward
parity stays UNPINNED.

Wire format (protobuf, walked by ``oracle/onnx_init._fields``): ModelProto.graph = 7; GraphProto.node = 1,
.initializer = 5, .input = 11, .output = 12; NodeProto.input = 1, .output = 2, .name = 3, .op_type = 4,
.attribute = 5; AttributeProto.name = 1, .f = 2, .i = 3, .s = 4, .floats = 7, .ints = 8.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import Dict, List

import numpy as np
import torch
import torch.nn.functional as F

from .onnx_init import _fields, _varint, read_initializers


@dataclass
class Node:
    op: str
    name: str
    inputs: List[str]
    outputs: List[str]
    attrs: Dict[str, object] = field(default_factory=dict)


def _ints(values) -> List[int]:
    out: List[int] = []
    for wt, v in values:
        if wt == 0:
            out.append(v)
        else:  # packed
            p = 0
            while p < len(v):
                d, p = _varint(v, p)
                out.append(d)
    return [x - (1 << 64) if x >= 1 << 63 else x for x in out]


def read_graph(path: str):
    """-> (nodes in file order (= topological for an exporter's output), initializers, input names, output names)."""
    with open(path, "rb") as f:
        model = f.read()
    graph = next(v for fno, wt, v in _fields(model) if fno == 7 and wt == 2)
    nodes: List[Node] = []
    g_in: List[str] = []
    g_out: List[str] = []
    for gno, gwt, body in _fields(graph):
        if gno == 1:
            n = Node("", "", [], [])
            for no, nwt, x in _fields(body):
                if no == 1:
                    n.inputs.append(x.decode())
                elif no == 2:
                    n.outputs.append(x.decode())
                elif no == 3:
                    n.name = x.decode()
                elif no == 4:
                    n.op = x.decode()
                elif no == 5:
                    a: Dict[int, list] = {}
                    for ano, awt, y in _fields(x):
                        a.setdefault(ano, []).append((awt, y))
                    name = a[1][0][1].decode()
                    if 8 in a:
                        n.attrs[name] = _ints(a[8])
                    elif 7 in a:
                        vals: List[float] = []
                        for awt, y in a[7]:
                            vals.extend(np.frombuffer(y, "<f4").tolist())
                        n.attrs[name] = vals
                    elif 3 in a:
                        n.attrs[name] = _ints(a[3])[0]
                    elif 2 in a:
                        n.attrs[name] = struct.unpack("<f", a[2][0][1])[0]
                    elif 4 in a:
                        n.attrs[name] = a[4][0][1].decode()
            nodes.append(n)
        elif gno in (11, 12):
            name = next(v for fno, wt, v in _fields(body) if fno == 1).decode()
            (g_in if gno == 11 else g_out).append(name)
    init = read_initializers(path)
    return nodes, init, [n for n in g_in if n not in init], g_out


def _conv(n: Node, x, w, b):
    assert n.attrs.get("group", 1) == 1 and all(d == 1 for d in n.attrs.get("dilations", [1, 1]))
    pads = n.attrs.get("pads", [0, 0, 0, 0])  # [top, left, bottom, right]
    assert pads[0] == pads[2] and pads[1] == pads[3], "asymmetric pads are not used by this export"
    return F.conv2d(x, w, b, stride=tuple(n.attrs.get("strides", [1, 1])), padding=(pads[0], pads[1]))


def _resize(n: Node, vals):
    # opset 11-12 Resize(X, roi, scales, sizes); the export uses mode=nearest with integral scales, for which
    # coordinate_transformation_mode=asymmetric + nearest_mode=floor is out[y, x] = in[y // s, x // s]
    assert n.attrs.get("mode", "nearest") == "nearest"
    assert n.attrs.get("coordinate_transformation_mode", "half_pixel") == "asymmetric"
    assert n.attrs.get("nearest_mode", "round_prefer_floor") == "floor"
    x = vals[0]
    scales = vals[2] if len(vals) > 2 and vals[2] is not None and vals[2].numel() else None
    assert scales is not None, "Resize by sizes is not used by this export"
    sc = [float(s) for s in scales.tolist()]
    assert sc[0] == 1.0 and sc[1] == 1.0 and sc[2] == int(sc[2]) and sc[3] == int(sc[3])
    return x.repeat_interleave(int(sc[2]), dim=2).repeat_interleave(int(sc[3]), dim=3)


def run(nodes: List[Node], init: Dict[str, np.ndarray], feeds: Dict[str, torch.Tensor], outputs: List[str]) -> Dict[str, torch.Tensor]:
    """Execute the node list on fp32 torch tensors (NCHW, ONNX axis order).  Shape-carrying tensors stay int64."""
    env: Dict[str, torch.Tensor] = {k: torch.from_numpy(v) for k, v in init.items()}
    env.update(feeds)
    for n in nodes:
        v = [env[i] if i else None for i in n.inputs]
        op = n.op
        if op == "Conv":
            out = _conv(n, v[0], v[1], v[2] if len(v) > 2 else None)
        elif op == "Sigmoid":
            out = torch.sigmoid(v[0])
        elif op == "Mul":
            out = v[0] * v[1]
        elif op == "Add":
            out = v[0] + v[1]
        elif op == "Sub":
            out = v[0] - v[1]
        elif op == "Div":
            out = v[0] / v[1]
        elif op == "Concat":
            out = torch.cat(v, dim=n.attrs["axis"])
        elif op == "Split":
            parts = torch.split(v[0], n.attrs["split"], dim=n.attrs.get("axis", 0))
            for name, p in zip(n.outputs, parts):
                env[name] = p
            continue
        elif op == "MaxPool":
            k, s, p = n.attrs["kernel_shape"], n.attrs.get("strides", [1, 1]), n.attrs.get("pads", [0, 0, 0, 0])
            assert p[0] == p[2] and p[1] == p[3] and n.attrs.get("ceil_mode", 0) == 0
            out = F.max_pool2d(v[0], tuple(k), tuple(s), (p[0], p[1]))  # ONNX pads MaxPool with -inf, as torch does
        elif op == "Resize":
            out = _resize(n, v)
        elif op == "Reshape":
            shape = [int(s) for s in v[1].tolist()]
            shape = [v[0].shape[i] if s == 0 else s for i, s in enumerate(shape)]  # 0 = copy the input dimension
            out = v[0].reshape(shape)
        elif op == "Transpose":
            out = v[0].permute(n.attrs["perm"])
        elif op == "Softmax":
            # opset < 13: coerce to 2-D at `axis`, softmax over the flattened tail; the export applies it on the last axis
            ax = n.attrs.get("axis", 1)
            assert ax == v[0].dim() - 1 or ax == -1
            out = torch.softmax(v[0], dim=-1)
        elif op == "Slice":
            starts, ends = v[1].tolist(), v[2].tolist()
            axes = v[3].tolist() if len(v) > 3 and v[3] is not None else list(range(len(starts)))
            steps = v[4].tolist() if len(v) > 4 and v[4] is not None else [1] * len(starts)
            out = v[0]
            for st, en, ax, sp in zip(starts, ends, axes, steps):
                assert sp == 1
                dim = out.shape[ax]
                st = max(0, min(dim, st + dim if st < 0 else st))
                en = max(0, min(dim, en + dim if en < 0 else en))
                out = out.narrow(ax, st, en - st)
        else:
            raise NotImplementedError(f"ONNX operator {op} ({n.name})")
        env[n.outputs[0]] = out
    return {k: env[k] for k in outputs}


def forward(path: str, x: torch.Tensor) -> torch.Tensor:
    """`x` fp32 [B, 3, H, W] RGB / 255 -> `output0` [B, 4 + nc, A] (what NCNN calls out0)."""
    nodes, init, g_in, g_out = read_graph(path)
    with torch.no_grad():
        return run(nodes, init, {g_in[0]: x}, g_out)[g_out[0]]
