#!/usr/bin/env python3
"""Bisect aid for the whole-C2f launches (c2f_kernels.hip): run the fp16 detector with and without them
(LITEPI_NO_C2F) on the same images and compare module outputs blob by blob with each other and with the
CPU oracle.  Usage: python tools/c2f_check.py [v1|v2] [batch]"""
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "yolo-litepi_amd"))


def main():
    preset = sys.argv[1] if len(sys.argv) > 1 else "v1"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    from litepi import Engine, ncnn_export
    from litepi.ncnn_io import read_param_layers
    from oracle import ncnn_ref

    tmp = tempfile.mkdtemp(prefix="c2f_check_")
    param, binf = os.path.join(tmp, "m.param"), os.path.join(tmp, "m.bin")
    ncnn_export.export_detector(param, binf, preset, seed=5, cls_bias=0.0)
    rng = np.random.default_rng(11)
    imgs = rng.integers(0, 256, (B, 640, 640, 3), dtype=np.uint8)
    layers = read_param_layers(param)
    # outputs of every Swish that follows a 1x1 conv (module outputs, cv1 outputs) + adds
    names = []
    conv_k = {}
    for l in layers:
        if l["type"] == "Convolution":
            conv_k[l["outputs"][0]] = l["params"].get(1, 1) if "params" in l else 1
    for l in layers:
        if l["type"] in ("Swish", "BinaryOp", "Pooling"):
            names.append(l["outputs"][0])
    ol = ncnn_ref.load_model(param, binf)
    x = torch.from_numpy(imgs[..., ::-1].astype(np.float32) * np.float32(1 / 255.0)).permute(0, 3, 1, 2).contiguous()
    ref = ncnn_ref.run_graph(ol, x, keep=names + ["out0"])

    def run(no_c2f):
        if no_c2f:
            os.environ["LITEPI_NO_C2F"] = "1"
        else:
            os.environ.pop("LITEPI_NO_C2F", None)
        e = Engine(precision="fp16", max_batch=max(B, 4))
        blobs = {}
        try:
            e.load_detector(param, binf)
            out0 = e.detect_raw(imgs)
            e.profile_next(True)
            e.detect_raw(imgs)
            prof = e.profile_read()
            print(f"--- plan ({'old' if no_c2f else 'c2f'}): {len(prof)} launches, {sum(p['ms'] for p in prof) * 1e3:.1f} us")
            for p in prof:
                print(f"   {p['name']:<36} {p['layer']:<34} {p['ms'] * 1e3:8.1f} us")
            for n in names:
                try:
                    blobs[n] = e.debug_blob(n)
                except Exception:
                    pass
        finally:
            e.close()
        return out0, blobs

    new0, newb = run(False)
    old0, oldb = run(True)
    print(f"{'blob':>6} {'shape':>18} {'new-ref':>10} {'old-ref':>10} {'new-old':>10}  ref absmax")
    for n in names:
        if n not in ref:
            continue
        a = newb.get(n)
        r = ref[n].numpy()
        if a is not None:
            r = r[:a.shape[0]]
        b = oldb.get(n)
        ea = float(np.abs(a - r).max()) if a is not None else float("nan")
        eb = float(np.abs(b - r).max()) if b is not None else float("nan")
        eab = float(np.abs(a - b).max()) if (a is not None and b is not None) else float("nan")
        flag = "  <-- " if (a is not None and (not np.isfinite(ea) or ea > 4 * max(eb if np.isfinite(eb) else 0.0, 0.02))) else ""
        print(f"{n:>6} {str(r.shape):>18} {ea:10.4f} {eb:10.4f} {eab:10.4f}  {np.abs(r).max():8.3f}{flag}")
    r0 = ref["out0"].numpy()
    for tag, o in (("new", new0), ("old", old0)):
        print(f"out0 {tag}: score err {np.abs(o[:, 4] - r0[:, 4]).max():.4f}  box err {np.abs(o[:, :4] - r0[:, :4]).max():.3f}  nan {np.isnan(o).sum()}")
    print(f"out0 new-old: score {np.abs(new0[:, 4] - old0[:, 4]).max():.4f} box {np.abs(new0[:, :4] - old0[:, :4]).max():.3f}")


if __name__ == "__main__":
    main()
