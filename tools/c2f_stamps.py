#!/usr/bin/env python3
"""Per-phase cycle medians of the whole-C2f launches from a LITEPI_C2F_STAMPS=<file> dump (c2f_kernels.hip, C2F_STAMP):
stamp 1 start, 2 after the entry conv, 3 after cv1, 4 after the bottlenecks, 5 after cv2, 6 after the SPPF pools, 7 end
(clock64 cycles); 0 / 15 wall clock (100 MHz ticks)."""
import sys
import numpy as np

name, rows, out = None, [], {}
for ln in open(sys.argv[1]):
    if ln.startswith("#"):
        if name and rows:
            out.setdefault(name, []).append(np.array(rows, dtype=np.int64))
        name, rows = ln.split()[1], []
    else:
        rows.append([int(x) for x in ln.split()])
if name and rows:
    out.setdefault(name, []).append(np.array(rows, dtype=np.int64))
labels = ["s2", "cv1", "bnecks", "cv2", "pools", "tail"]
for k, runs in out.items():
    a = runs[-1]
    d = [np.median(a[:, i + 1] - a[:, i]) for i in range(1, 7)]
    wall = np.median(a[:, 15] - a[:, 0]) / 100.0
    span = (a[:, 15].max() - a[:, 0].min()) / 100.0
    order = np.argsort(a[:, 0])
    dur = (a[:, 15] - a[:, 0]) / 100.0
    first, rest = dur[order[:256]], dur[order[256:]]
    if (a[:, 9] > 0).all() and (a[:, 11] == 0).all():   # s2 phase stamps (s2conv kernel, whole-image kernels)
        whole = a[:, 3].max() > 0
        end = a[:, 2] if whole else a[:, 7]
        start = a[:, 1] if whole else a[:, 2]
        print("      s2 phase, wave 0 block 0: before K loop %d  K loop %d  epilogue %d  rest of the phase %d%s" % (
            np.median(a[:, 8] - start), np.median(a[:, 9] - a[:, 8]), np.median(a[:, 10] - a[:, 9]), np.median(end - a[:, 10]),
            "" if whole else "  (weights staged in %d)" % np.median(a[:, 2] - a[:, 1])))
    if a[:, 14].max() > 0:   # whole-image cv2 (pw_sync_phase): 11 K loop start, 12 K loop end, 13 behind the barrier, 14 epilogue end
        print("      cv2 (one round): before K loop %d  K loop %d  barrier %d  epilogue %d" % (
            np.median(a[:, 11] - a[:, 4]), np.median(a[:, 12] - a[:, 11]), np.median(a[:, 13] - a[:, 12]), np.median(a[:, 14] - a[:, 13])))
    a2 = a[(a[:, 8] > 0) & (a[:, 11] > 0) & (a[:, 14] == 0)]
    if len(a2):
        inner = [np.median(a2[:, 8] - a2[:, 2]), np.median(a2[:, 9] - a2[:, 8]), np.median(a2[:, 10] - a2[:, 9]), np.median(a2[:, 3] - a2[:, 10]),
                 np.median(a2[:, 11] - a2[:, 4]), np.median(a2[:, 12] - a2[:, 11]), np.median(a2[:, 13] - a2[:, 12]), np.median(a2[:, 5] - a2[:, 13])]
        print("      cv1: setup %d kloop %d epi %d drain+barrier %d | cv2: setup %d kloop %d epi %d end %d" % tuple(inner))
    extra = f"  first-256 wall {np.median(first):6.1f}" + (f" later {np.median(rest):6.1f}" if len(rest) else "")
    print(f"{k:<44} wgs {len(a):5d}  " + "  ".join(f"{l} {v:8.0f}" for l, v in zip(labels, d)) + f"   wg wall {wall:6.1f} us, launch span {span:6.1f} us" + extra)
