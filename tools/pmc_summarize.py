#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSVs (tools/pmc_profile.sh) per kernel name.

    python tools/pmc_summarize.py gpurun_out/pmc_r1 > profiles/r01_pmc_summary.txt

FETCH_SIZE on gfx950 under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM): the
HBM read column applies that correction (x2); WRITE_SIZE is exact.  Both are in KiB units in the tool.
"""
import collections
import csv
import glob
import os
import re
import sys


def short(name):
    m = re.search(r"lp::?(\w+)|_ZN2lp\d+(\w+?)I", name)
    base = name
    if name.startswith("_ZN2lp12_GLOBAL__N_1"):   # kernels of an anonymous namespace (c2f_kernels.hip): name + integer template arguments
        m = re.match(r"_ZN2lp12_GLOBAL__N_1\d+([a-z0-9_]+?)I", name)
        args = re.findall(r"L[ib](\d+)E", name)
        return (m.group(1) if m else name) + "<" + ",".join(args) + ">"
    if name.startswith("_ZN2lp"):
        m = re.match(r"_ZN2lp\d+([a-z0-9_]+?)(I.*)?E", name)
        base = m.group(1) if m else name
        tm = re.search(r"I(DF16_|f)((?:Li\d+E)*)", name)
        if tm:
            args = re.findall(r"Li(\d+)E", tm.group(2))
            base += "<" + ("f16" if tm.group(1) == "DF16_" else "f32") + ("," + ",".join(args) if args else "") + ">"
    elif "s2conv_kernel<" in name:
        m = re.search(r"S2Cfg<([^>]*)>", name)
        base = "s2conv_kernel<" + (m.group(1).replace(" ", "") if m else "") + ">"
    elif "c2f_kernel<" in name:   # demangled, anonymous namespace: keep the configuration's template arguments
        m = re.search(r"C2fCfg<([^>]*)>", name)
        base = "c2f_kernel<" + (m.group(1).replace(" ", "") if m else "") + ">"
    else:
        base = re.sub(r"\(.*", "", name).replace("void ", "").replace("lp::", "")
    return base


def main(root):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    for f in sorted(glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True)):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (f, r["Dispatch_Id"])
            if key not in seen and "pass1" in f:
                seen.add(key)
                calls[k] += 1
    dur = collections.defaultdict(float)
    for f in sorted(glob.glob(os.path.join(root, "pass1", "**", "*kernel_trace.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            dur[short(r["Kernel_Name"])] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    cols = ["calls", "us(pmc run)", "wave_cyc(M)", "wait_any%", "wait_inst%", "active%", "lds_conf%", "mfma_busy%", "valu/mfma",
            "HBM rd MB", "HBM wr MB", "L2 hit%"]
    print(f"{'kernel':42s}" + "".join(f"{c:>13s}" for c in cols))
    for k in sorted(agg, key=lambda k: -dur.get(k, 0)):
        a = agg[k]
        wc = a.get("SQ_WAVE_CYCLES", 0) or 1
        row = [calls[k], dur.get(k, 0), wc / 1e6, 100 * a.get("SQ_WAIT_ANY", 0) / wc, 100 * a.get("SQ_WAIT_INST_ANY", 0) / wc,
               100 * a.get("SQ_ACTIVE_INST_ANY", 0) / wc,
               100 * a.get("SQ_LDS_BANK_CONFLICT", 0) / (a.get("SQ_LDS_IDX_ACTIVE", 0) or 1),
               100 * a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / ((a.get("GRBM_GUI_ACTIVE", 0) / 8 or 1) * 1024),
               a.get("SQ_INSTS_VALU", 0) / (a.get("SQ_INSTS_MFMA", 0) or 1),
               2 * a.get("FETCH_SIZE", 0) / 1024, a.get("WRITE_SIZE", 0) / 1024,
               100 * a.get("TCC_HIT_sum", 0) / ((a.get("TCC_HIT_sum", 0) + a.get("TCC_MISS_sum", 0)) or 1)]
        print(f"{k[:42]:42s}" + "".join(f"{v:13.1f}" if isinstance(v, float) else f"{v:13d}" for v in row))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc")
