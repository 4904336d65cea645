"""Tiny NCNN ``.param`` text reader used by host-side tooling (not the inference path:
the library parses model files itself in C++, csrc/ncnn_graph.cpp)."""
from __future__ import annotations

from typing import Dict, List


def _val(v: str):
    try:
        return int(v)
    except ValueError:
        return float(v)


def read_param_layers(path: str) -> List[dict]:
    with open(path) as f:
        lines = [ln.strip() for ln in f if ln.strip()]
    if lines[0] != "7767517":
        raise RuntimeError(f"Failed to load param: {path} (bad magic)")
    layers = []
    for ln in lines[2:]:
        t = ln.split()
        n_in, n_out = int(t[2]), int(t[3])
        params: Dict[int, object] = {}
        for kv in t[4 + n_in + n_out:]:
            k, v = kv.split("=", 1)
            k = int(k)
            if k <= -23300:
                vs = v.split(",")
                params[k] = [_val(x) for x in vs[1:1 + int(vs[0])]]
            else:
                params[k] = _val(v)
        layers.append(dict(type=t[0], name=t[1], inputs=t[4:4 + n_in], outputs=t[4 + n_in:4 + n_in + n_out], params=params))
    return layers
