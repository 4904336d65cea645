"""Writer of synthetic YOLOv8-family detectors in NCNN ``.param``/``.bin`` format.

The reference ships real weights only for YOLO-LitePi v1 and they cannot travel to the
GPU box, so tests and ``bench.py`` run on seeded random-weight models of the SAME
architecture (same layer sequence, channel widths and NCNN file format as
``src/*/convert/model/yolo_plus/yolo_plus_ncnn_model/model.ncnn.param`` of the reference;
SURVEY Appendix A/C).  Presets:

    v1  : widths 8/16/32/64/128   (VN-Signs model, 0.97 M params, 2.84 GFLOP @640)
    v2  : widths 16/24/48/96/192  (the paper's YOLO-LitePi, 1.80 M params, 5.09 GFLOP @640)

Weights are seeded normal, then rescaled layer by layer (LSUV style, on a seeded random
input) so that every conv's pre-activation has unit standard deviation: activations stay O(1)
through the ~20 SiLU convs of the deepest path (safe in fp16).  ``cls_bias`` shifts the three
class-branch biases; ``bench.py`` calibrates it so that a handful of anchors per image pass
conf 0.25.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

PRESETS: Dict[str, dict] = {
    "v1": dict(c=(8, 16, 32, 64, 128), n=(1, 2, 2, 1), f4=64, f3=32, d4conv=64, d4=64, d5conv=128, d5=128,
               head_box=64, head_cls=32),
    "v2": dict(c=(16, 24, 48, 96, 192), n=(1, 2, 2, 1), f4=96, f3=48, d4conv=48, d4=96, d5conv=96, d5=192,
               head_box=64, head_cls=48),
}


@dataclass
class _Op:
    type: str
    name: str
    inputs: List[int]
    outputs: List[int]
    params: str = ""
    weight: Optional[np.ndarray] = None
    bias: Optional[np.ndarray] = None
    data: Optional[np.ndarray] = None


class _Builder:
    def __init__(self, rng: np.random.Generator, gain: float):
        self.ops: List[_Op] = []
        self.nblobs = 0
        self.rng = rng
        self.gain = gain
        self.counts: Dict[str, int] = {}

    def _blob(self) -> int:
        self.nblobs += 1
        return self.nblobs - 1

    def _name(self, prefix: str) -> str:
        i = self.counts.get(prefix, 0)
        self.counts[prefix] = i + 1
        return f"{prefix}_{i}"

    def input(self) -> int:
        b = self._blob()
        self.ops.append(_Op("Input", "in0", [], [b]))
        return b

    def conv(self, x: int, cin: int, cout: int, k: int, s: int = 1, act: bool = True, bias_init: float = 0.0,
             w_scale: float = 1.0) -> int:
        fan_in = cin * k * k
        w = self.rng.standard_normal((cout, cin, k, k)).astype(np.float32) * np.float32(
            w_scale * (self.gain / fan_in) ** 0.5)
        b = (self.rng.standard_normal(cout).astype(np.float32) * np.float32(0.05) + np.float32(bias_init))
        y = self._blob()
        p = k // 2
        self.ops.append(_Op("Convolution", self._name("conv"), [x], [y],
                            f"0={cout} 1={k} 11={k} 12=1 13={s} 14={p} 2=1 3={s} 4={p} 5=1 6={cout * cin * k * k}",
                            weight=w, bias=b))
        if act:
            z = self._blob()
            self.ops.append(_Op("Swish", self._name("silu"), [y], [z]))
            return z
        return y

    def slice2(self, x: int, a: int, b: int) -> Tuple[int, int]:
        o1, o2 = self._blob(), self._blob()
        self.ops.append(_Op("Slice", self._name("split"), [x], [o1, o2], f"-23300=2,{a},{b} 1=0"))
        return o1, o2

    def concat(self, xs: Sequence[int]) -> int:
        y = self._blob()
        self.ops.append(_Op("Concat", self._name("cat"), list(xs), [y], "0=0"))
        return y

    def add(self, a: int, b: int) -> int:
        y = self._blob()
        self.ops.append(_Op("BinaryOp", self._name("add"), [a, b], [y], "0=0"))
        return y

    def pool5(self, x: int) -> int:
        y = self._blob()
        self.ops.append(_Op("Pooling", self._name("maxpool2d"), [x], [y], "0=0 1=5 11=5 12=1 13=2 2=1 3=2 5=1"))
        return y

    def up2(self, x: int) -> int:
        y = self._blob()
        self.ops.append(_Op("Interp", self._name("upsample"), [x], [y], "0=1 1=2.0 2=2.0 6=0"))
        return y

    def raw(self, type_: str, prefix: str, ins: Sequence[int], nout: int, params: str = "", **kw) -> List[int]:
        outs = [self._blob() for _ in range(nout)]
        self.ops.append(_Op(type_, self._name(prefix), list(ins), outs, params, **kw))
        return outs

    # ---- composite modules -----------------------------------------------------------
    def c2f(self, x: int, cin: int, cout: int, n: int) -> int:
        h = cout // 2
        y = self.conv(x, cin, cout, 1)
        a, b = self.slice2(y, h, h)
        parts = [a, b]
        cur = b
        for _ in range(n):
            t = self.conv(cur, h, h, 3)
            t = self.conv(t, h, h, 3, w_scale=0.7)
            cur = self.add(cur, t)
            parts.append(cur)
        return self.conv(self.concat(parts), (2 + n) * h, cout, 1)

    def sppf(self, x: int, c: int) -> int:
        h = c // 2
        y = self.conv(x, c, h, 1)
        p1 = self.pool5(y)
        p2 = self.pool5(p1)
        p3 = self.pool5(p2)
        return self.conv(self.concat([y, p1, p2, p3]), 4 * h, c, 1)

    # ---- LSUV-style calibration: rescale every conv so its pre-activation std is ~target -----
    def calibrate(self, size: int = 320, target: float = 1.0) -> None:
        import torch
        import torch.nn.functional as F

        g = torch.Generator().manual_seed(7)
        vals = {}
        with torch.no_grad():
            for op in self.ops:
                if op.type == "Input":
                    vals[op.outputs[0]] = torch.rand(2, 3, size, size, generator=g)
                elif op.type == "Convolution":
                    if op.bias is None:
                        break  # DFL conv: the decode tail needs no calibration
                    x = vals[op.inputs[0]]
                    k = op.weight.shape[-1]
                    s = int(op.params.split("3=")[1].split()[0])
                    y = F.conv2d(x, torch.from_numpy(op.weight), None, stride=s, padding=k // 2)
                    sd = float(y.std())
                    if sd > 0:
                        op.weight *= np.float32(target / sd)
                        y = y * (target / sd)
                    vals[op.outputs[0]] = y + torch.from_numpy(op.bias).view(1, -1, 1, 1)
                elif op.type == "Swish":
                    x = vals[op.inputs[0]]
                    vals[op.outputs[0]] = x * torch.sigmoid(x)
                elif op.type == "Slice":
                    sizes = [int(v) for v in op.params.split("=")[1].split()[0].split(",")[1:]]
                    for o, part in zip(op.outputs, torch.split(vals[op.inputs[0]], sizes, dim=1)):
                        vals[o] = part
                elif op.type == "Concat":
                    vals[op.outputs[0]] = torch.cat([vals[i] for i in op.inputs], dim=1)
                elif op.type == "BinaryOp":
                    vals[op.outputs[0]] = vals[op.inputs[0]] + vals[op.inputs[1]]
                elif op.type == "Pooling":
                    vals[op.outputs[0]] = F.max_pool2d(vals[op.inputs[0]], 5, 1, 2)
                elif op.type == "Interp":
                    vals[op.outputs[0]] = F.interpolate(vals[op.inputs[0]], scale_factor=2.0, mode="nearest")
                elif op.type == "MemoryData":
                    continue
                else:
                    break

    # ---- emission: insert Split layers, number blobs, write files ---------------------------
    def write(self, param_path: str, bin_path: str) -> None:
        uses: Dict[int, int] = {}
        for op in self.ops:
            for i in op.inputs:
                uses[i] = uses.get(i, 0) + 1
        final: List[_Op] = []
        remap: Dict[int, List[int]] = {}
        nsplit = 0
        for op in self.ops:
            ins = []
            for i in op.inputs:
                if i in remap:
                    ins.append(remap[i].pop(0))
                else:
                    ins.append(i)
            final.append(_Op(op.type, op.name, ins, op.outputs, op.params, op.weight, op.bias, op.data))
            for o in op.outputs:
                if uses.get(o, 0) > 1:
                    outs = [self._blob() for _ in range(uses[o])]
                    final.append(_Op("Split", f"splitncnn_{nsplit}", [o], outs))
                    nsplit += 1
                    remap[o] = list(outs)
        names: Dict[int, str] = {}
        for op in final:
            for o in op.outputs:
                names[o] = str(len(names))
        names[final[0].outputs[0]] = "in0"
        names[final[-1].outputs[0]] = "out0"
        with open(param_path, "w") as f:
            f.write("7767517\n")
            f.write(f"{len(final)} {len(names)}\n")
            for op in final:
                toks = [f"{op.type:<24}", f"{op.name:<24}", str(len(op.inputs)), str(len(op.outputs))]
                toks += [names[i] for i in op.inputs] + [names[o] for o in op.outputs]
                if op.params:
                    toks.append(op.params)
                f.write(" ".join(toks) + "\n")
        with open(bin_path, "wb") as f:
            for op in final:
                if op.type == "Convolution":
                    f.write(struct.pack("<I", 0))
                    f.write(np.ascontiguousarray(op.weight, "<f4").tobytes())
                    if op.bias is not None:
                        f.write(np.ascontiguousarray(op.bias, "<f4").tobytes())
                elif op.type == "MemoryData":
                    f.write(np.ascontiguousarray(op.data, "<f4").tobytes())


def make_anchors(size: int, strides: Sequence[int] = (8, 16, 32)) -> Tuple[np.ndarray, np.ndarray]:
    """Anchor points (grid units, +0.5) as [2, A] and per-anchor stride [A], levels in order."""
    xs, ys, ss = [], [], []
    for s in strides:
        n = size // s
        gy, gx = np.meshgrid(np.arange(n, dtype=np.float32) + 0.5, np.arange(n, dtype=np.float32) + 0.5, indexing="ij")
        xs.append(gx.ravel())
        ys.append(gy.ravel())
        ss.append(np.full(n * n, s, np.float32))
    return np.stack([np.concatenate(xs), np.concatenate(ys)]), np.concatenate(ss)


def export_detector(param_path: str, bin_path: str, preset: str = "v1", seed: int = 1234, nc: int = 1,
                    reg_max: int = 16, size: int = 640, cls_bias: float = -4.0, gain: float = 2.0, box_ramp: float = 0.45,
                    spec: Optional[dict] = None) -> dict:
    """Write a seeded random-weight detector; returns the spec used."""
    sp = dict(PRESETS[preset]) if spec is None else dict(spec)
    c1, c2, c3, c4, c5 = sp["c"]
    n = sp["n"]
    g = _Builder(np.random.default_rng(seed), gain)
    x = g.input()
    x = g.conv(x, 3, c1, 3, 2)
    x = g.conv(x, c1, c2, 3, 2)
    x = g.c2f(x, c2, c2, n[0])
    x = g.conv(x, c2, c3, 3, 2)
    p3 = g.c2f(x, c3, c3, n[1])
    x = g.conv(p3, c3, c4, 3, 2)
    p4 = g.c2f(x, c4, c4, n[2])
    x = g.conv(p4, c4, c5, 3, 2)
    x = g.c2f(x, c5, c5, n[3])
    p5 = g.sppf(x, c5)
    f4 = g.c2f(g.concat([g.up2(p5), p4]), c5 + c4, sp["f4"], 1)
    f3 = g.c2f(g.concat([g.up2(f4), p3]), sp["f4"] + c3, sp["f3"], 1)
    d4 = g.c2f(g.concat([g.conv(f3, sp["f3"], sp["d4conv"], 3, 2), f4]), sp["d4conv"] + sp["f4"], sp["d4"], 1)
    d5 = g.c2f(g.concat([g.conv(d4, sp["d4"], sp["d5conv"], 3, 2), p5]), sp["d5conv"] + c5, sp["d5"], 1)

    anchors, strides = make_anchors(size)
    A = anchors.shape[1]
    stride_blob = g.raw("MemoryData", "pnnx", [], 1, f"0={A}", data=strides)[0]
    hb, hc = sp["head_box"], sp["head_cls"]
    level_cats = []
    for feat, cin in ((f3, sp["f3"]), (d4, sp["d4"]), (d5, sp["d5"])):
        b = g.conv(feat, cin, hb, 3)
        b = g.conv(b, hb, hb, 3)
        b = g.conv(b, hb, 4 * reg_max, 1, act=False)
        # DFL bias ramp: favour the low distance bins so boxes are a few strides wide (traffic
        # signs: ~54 px in TT100K frames) instead of the ~8-bin mean of flat logits
        g.ops[-1].bias = (g.ops[-1].bias + np.tile(-box_ramp * np.arange(reg_max, dtype=np.float32), 4)).astype(np.float32)
        c = g.conv(feat, cin, hc, 3)
        c = g.conv(c, hc, hc, 3)
        c = g.conv(c, hc, nc, 1, act=False, bias_init=cls_bias)
        level_cats.append(g.concat([b, c]))
    rs = []
    for cat, s in zip(level_cats, (8, 16, 32)):
        hw = (size // s) ** 2
        rs.append(g.raw("Reshape", "reshape", [cat], 1, f"0={hw} 1={4 * reg_max + nc}")[0])
    allc = g.raw("Concat", "cat", rs, 1, "0=1")[0]
    box, cls = g.raw("Slice", "split", [allc], 2, f"-23300=2,{4 * reg_max},{nc} 1=0")
    box = g.raw("Reshape", "reshape", [box], 1, f"0={A} 1={reg_max} 2=4")[0]
    box = g.raw("Permute", "transpose", [box], 1, "0=2")[0]
    box = g.raw("Softmax", "softmax", [box], 1, "0=0 1=1")[0]
    dfl = _Op("Convolution", "conv_dfl", [box], [g._blob()],
              f"0=1 1=1 11=1 12=1 13=1 14=0 2=1 3=1 4=0 5=0 6={reg_max}",
              weight=np.arange(reg_max, dtype=np.float32).reshape(1, reg_max, 1, 1))
    g.ops.append(dfl)
    dist = g.raw("Reshape", "reshape", [dfl.outputs[0]], 1, f"0={A} 1=4")[0]
    a1 = g.raw("MemoryData", "pnnx_fold_anchor_points", [], 1, f"0={A} 1=2", data=anchors)[0]
    a2 = g.raw("MemoryData", "pnnx_fold_anchor_points", [], 1, f"0={A} 1=2", data=anchors)[0]
    lt, rb = g.raw("Slice", "chunk", [dist], 2, "-23300=2,-233,-233 1=0")
    x1y1 = g.raw("BinaryOp", "sub", [a1, lt], 1, "0=1")[0]
    x2y2 = g.raw("BinaryOp", "add", [a2, rb], 1, "0=0")[0]
    cxy = g.raw("BinaryOp", "add", [x1y1, x2y2], 1, "0=0")[0]
    cxy = g.raw("BinaryOp", "div", [cxy], 1, "0=3 1=1 2=2.0")[0]
    wh = g.raw("BinaryOp", "sub", [x2y2, x1y1], 1, "0=1")[0]
    xywh = g.raw("Concat", "cat", [cxy, wh], 1, "0=0")[0]
    st = g.raw("Reshape", "reshape", [stride_blob], 1, f"0={A} 1=1")[0]
    xywh = g.raw("BinaryOp", "mul", [xywh, st], 1, "0=2")[0]
    sc = g.raw("Sigmoid", "sigmoid", [cls], 1)[0]
    g.raw("Concat", "cat", [xywh, sc], 1, "0=0")
    g.calibrate()
    g.write(param_path, bin_path)
    sp.update(preset=preset, seed=seed, nc=nc, reg_max=reg_max, size=size, cls_bias=cls_bias, num_anchors=A)
    return sp


def shift_cls_bias(param_path: str, bin_path: str, delta: float, nc: int = 1) -> None:
    """Add ``delta`` to the biases of the class-projection convs (out channels == nc, 1x1,
    no activation) of an exported model, in place."""
    from .ncnn_io import read_param_layers  # local import: tiny text helper

    layers = read_param_layers(param_path)
    with open(bin_path, "rb") as f:
        blob = bytearray(f.read())
    off = 0
    swish_inputs = {l["inputs"][0] for l in layers if l["type"] == "Swish"}
    for l in layers:
        if l["type"] == "Convolution":
            out_ch, wcount, has_bias = l["params"][0], l["params"][6], l["params"].get(5, 0)
            off += 4 + 4 * wcount
            if has_bias:
                if out_ch == nc and l["params"][1] == 1 and l["outputs"][0] not in swish_inputs:
                    b = np.frombuffer(blob, "<f4", out_ch, off).copy() + np.float32(delta)
                    blob[off:off + 4 * out_ch] = b.astype("<f4").tobytes()
                off += 4 * out_ch
        elif l["type"] == "MemoryData":
            n = 1
            for k in (0, 1, 2):
                v = l["params"].get(k, 0)
                if v:
                    n *= v
            off += 4 * n
    with open(bin_path, "wb") as f:
        f.write(bytes(blob))


def seeded_bin_for_param(param_path: str, bin_path: str, seed: int = 1234, gain: float = 1.6, size: int = 640,
                         cls_bias: float = -2.0) -> None:
    """Write a seeded random fp32 ``.bin`` that matches an existing NCNN ``.param`` of the YOLOv8 family
    (same file layout as a real export: per Convolution a zero flag word, weights, optional bias; MemoryData
    payloads).  Used to exercise graphs whose weights are not available (the reference ships the YOLOv8n /
    YOLOv5nu graphs of its baseline comparison without weights).  He-style init scaled by ``gain``; the Detect
    tail's DFL conv (no bias, one output channel) gets arange(reg_max); MemoryData blobs get the anchor grid /
    strides of ``size``; convs that feed a Sigmoid through at most a Concat/Slice get ``cls_bias``."""
    from .ncnn_io import read_param_layers
    layers = read_param_layers(param_path)
    rng = np.random.default_rng(seed)
    anchors, strides = make_anchors(size)
    # class count: the Detect tail slices [4*reg_max, nc] off the head concat; reg_max = K of the bias-free 1-channel conv
    reg_max = next((int(L["params"][6]) for L in layers if L["type"] == "Convolution" and int(L["params"][0]) == 1 and
                    int(L["params"].get(5, 0)) == 0), 16)
    nc = next((int(L["params"][-23300][1]) for L in layers if L["type"] == "Slice" and -23300 in L["params"] and
               len(L["params"][-23300]) == 2 and int(L["params"][-23300][0]) == 4 * reg_max), 1)
    with open(bin_path, "wb") as f:
        for L in layers:
            p = L["params"]
            if L["type"] in ("Convolution", "ConvolutionDepthWise"):
                out_ch, kw, wcount = int(p[0]), int(p.get(1, 1)), int(p[6])
                kh = int(p.get(11, kw))
                in_ch = wcount // (out_ch * kw * kh)  # per group for ConvolutionDepthWise
                has_bias = int(p.get(5, 0)) != 0
                if out_ch == 1 and not has_bias and kw == 1 and L["type"] == "Convolution":
                    w = np.arange(in_ch, dtype=np.float32).reshape(1, in_ch, 1, 1)  # DFL expectation
                else:
                    w = rng.standard_normal((out_ch, in_ch, kh, kw)).astype(np.float32) * np.float32(gain / np.sqrt(in_ch * kh * kw))
                f.write(struct.pack("<I", 0))
                f.write(np.ascontiguousarray(w, "<f4").tobytes())
                if has_bias:
                    b = (rng.standard_normal(out_ch) * 0.1).astype(np.float32)
                    if kw == 1 and out_ch == nc:
                        b[:] = cls_bias  # class projections: keeps the scores away from 0.5 on random weights
                    f.write(np.ascontiguousarray(b, "<f4").tobytes())
            elif L["type"] == "MemoryData":
                w, h = int(p.get(0, 0)), int(p.get(1, 0))
                data = strides if h == 0 else anchors
                if data.size != max(w, 1) * max(h, 1):
                    raise ValueError(f"MemoryData {L['name']}: {w}x{h} does not match the {size}x{size} anchor grid")
                f.write(np.ascontiguousarray(data, "<f4").tobytes())
