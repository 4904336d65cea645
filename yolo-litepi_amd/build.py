#!/usr/bin/env python3
"""Build liblitepi_hip.so for gfx950 (in-tree, so the .so travels with the repo snapshot).

    python yolo-litepi_amd/build.py [--force]

hipcc cross-compiles without a GPU.  One object per source (parallel), then one link.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
OUT = os.path.join(HERE, "litepi", "liblitepi_hip.so")
SOURCES = ["api.cpp", "ncnn_graph.cpp", "detector.cpp", "classifier.cpp", "resnet.cpp", "mbnet.cpp",
           "conv_kernels.hip", "misc_kernels.hip", "post_kernels.hip", "cls_kernels.hip", "cls_fused.hip", "cls_net.hip", "head_kernels.hip", "c2f_kernels.hip"]
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-x", "hip", "-Wall", "-Wno-unused-function",
         "-ffp-contract=off", "-fgpu-flush-denormals-to-zero"]


def newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(SRC, f) for f in os.listdir(SRC) if f.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "include", "litepi.h"))
    hdr_time = newest(headers)
    jobs = []
    for s in SOURCES:
        src = os.path.join(SRC, s)
        obj = os.path.join(OBJ, s + ".o")
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_time):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        return src, r.returncode, r.stdout + r.stderr

    failed = False
    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        for src, rc, log in ex.map(compile_one, jobs):
            if verbose and (rc != 0 or log.strip()):
                print(f"[build] {os.path.basename(src)} rc={rc}\n{log}")
            failed = failed or rc != 0
    if failed:
        raise RuntimeError("hipcc failed")
    objs = [os.path.join(OBJ, s + ".o") for s in SOURCES]
    if jobs or not os.path.exists(OUT) or os.path.getmtime(OUT) < newest(objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            print(r.stdout + r.stderr)
            raise RuntimeError("link failed")
    if verbose:
        print(f"[build] {OUT} ({os.path.getsize(OUT)} bytes)")
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
