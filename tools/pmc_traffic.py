#!/usr/bin/env python3
"""Per-kernel-family HBM traffic of ONE bench step from the rocprofv3 --pmc passes of tools/pmc_profile.sh.

    python tools/pmc_traffic.py gpurun_out/pmc_r01_final > profiles/r01_pmc_traffic.json

Uses pass3 (FETCH_SIZE) and pass4 (WRITE_SIZE, TCC hit/miss) and only the dispatches of the last step of each pass
(the earlier ones are the untimed calibration passes at a smaller batch).  FETCH_SIZE is doubled as
MI355X_MICROARCH.md prescribes for gfx950 (16-byte-per-lane coalesced reads are tallied at half their size);
both counters are in KiB.  bench.py reads the resulting file to fill roofline.traffic for the dominant family.
"""
import collections
import csv
import glob
import json
import re
import sys


def family(name):
    f16 = "_f16" if "DF16_" in name or "<f16" in name else "_f32"
    ints = [int(v) for v in re.findall(r"Li(\d+)E", name)]
    if "s2conv_kernel" in name or "s2lds_kernel" in name:   # both run under the library's profile name s2conv<Cin,Cout>
        m = re.search(r"S2L?Cfg<(\d+), *(\d+)[,>]", name)
        v = [int(m.group(1)), int(m.group(2))] if m else [int(x) for x in re.findall(r"Li(\d+)E", name)][:2]
        tail = re.search(r"S2LCfg<[^>]*true *>", name) is not None or ("S2LCfg" in name and "Lb1E" in name)   # S2LCfg<.., TAIL>
        return ("s2conv+1x1<%d,%d>_f16" if tail else "s2conv<%d,%d>_f16") % tuple(v)
    if "sppf_kernel" in name:   # SpCfg<C, CIN, COUT, OSPLIT>: the library's profile name is sppf<Cin,C,Cout>
        m = re.search(r"SpCfg<(\d+), *(\d+), *(\d+)", name)
        v = [int(m.group(i)) for i in (1, 2, 3)] if m else (ints + [0, 0, 0])[:3]
        return "sppf<%d,%d,%d>_f16" % (v[1], v[0], v[2])
    if "c2f_kernel" in name:   # C2fCfg<C, NB, KA, KB, UP, COUT, MODE, KS2, TH, NW>: the profiler's name is CfgName of c2f_kernels.hip
        m = re.search(r"C2fCfg<([^>]*)>", name)
        if m:
            t = [x.strip() for x in m.group(1).split(",")]
            v = [int(x) if x.lstrip("-").isdigit() else (1 if x == "true" else 0) for x in t]
        else:
            v = [int(x) for x in re.findall(r"L[ib](\d+)E", name)]
        C, NB, KA, KB, UP, COUT, MODE, KS2 = (v + [0] * 8)[:8]
        if MODE == -1:   # the module without its cv1 (C2fShape::MODE -1)
            src = "y0y1"
        elif MODE >= 1:
            src = "s2+%d" % KB + (",sppf" if MODE == 2 else "")
        else:
            src = ("up%d+%d" % (KA, KB)) if UP else str(KA + KB)
        return "c2f<%d,%d,%s>_f16" % (C, NB, src)
    if "bottleneck_mfma_kernel" in name:   # <T, NT, P1, P2, SEP, T2, SG>: the profiler's name carries <NT,P1,P2,T2,SG>
        v = (ints + [0, 0, 0, 0, 0])[:5]
        # the trailing template argument CL (the "concat from LDS" variant): Lb1E mangled, `true>` demangled
        cl = re.search(r"Lb1EEEv", name) is not None or re.search(r", true>\(", name) is not None
        return "bottleneck3x3x2<%d,%d,%d,%d,%d%s>" % (tuple(v) + (",cl" if cl else "",)) + f16
    if "head_fused_kernel" in name:   # <C3T, PA, PB, NPC, KSA, SLOTF, NRW, OV, A16, NCA>: the profiler appends "a16" for A16 = true
        args = name.split("head_fused_kernel", 1)[1].split(">", 1)[0] if "<" in name else ""
        t = [int(x) for x in re.findall(r"\d+", args)] if args else ints
        toks = [x.strip() for x in args.lstrip("<").split(",")] if args else []
        a16 = (len(toks) > 8 and toks[8] == "true") or (not args and name.count("Lb1E") >= 1 and re.search(r"Lb[01]ELb1E", name) is not None)
        sfx = "a16" if a16 else ""
        if a16 and t[0] == 2 and len(t) > 6:   # two class row tiles (v2): the library's name carries K steps per tap, Cin / 16
            sfx += "k%d" % {(12, 3): 3, (12, 4): 6, (24, 4): 12}.get((t[5], t[6]), 0)
        return "head_fused<%s>%s_f16" % (",".join(str(x) for x in t[:6]), sfx)
    if "conv3x3s2_direct_kernel" in name:   # <T, NT, NP, T2, U>
        t2 = ints[2] if len(ints) > 2 else 0
        return ("conv3x3s2_direct+1x1<%d,%d>" % (ints[0], t2) if t2 else "conv3x3s2_direct<%d>" % ints[0]) + f16
    if "conv3x3_mfma_kernel" in name:       # <T, NT, STRIDE, T2>
        t2 = ints[2] if len(ints) > 2 else 0
        return ("conv3x3_mfma+1x1<%d,%d>" % (ints[0], t2) if t2 else "conv3x3_mfma<%d>" % ints[0]) + f16
    if "conv1x1_mfma_kernel" in name:       # <T, NT, NP, EPI, UPS>
        ups = "Lb1E" in name or ", true>" in name
        return ("conv1x1_mfma<%d,up>" % ints[0] if ups else "conv1x1_mfma<%d>" % ints[0]) + f16
    if "stem_mfma_kernel" in name or "stem_conv" in name:
        return "stem_conv_f16" if "stem_mfma" in name else "stem_conv" + f16
    for key, fam in (("letterbox", "letterbox_u8"), ("stem_block16_kernel", "stem_block16_f16"), ("stem_block_kernel", "stem_block_f16"), ("roi_resize_kernel", "roi_resize_pil"), ("shuffle_stage_kernel", "shuffle_stage_fused_f16"),
                     ("cls_head_kernel", "cls_head_fused_f16"), ("nms_kernel", "nms"), ("roi_index_kernel", "roi_index"),
                     ("cls_front_kernel", "cls_front_f16"), ("cls_back_kernel", "cls_back_f16"),
                     ("sppf_pool", "sppf_pool_f16")):
        if key in name:
            return fam
    m = re.search(r"lp::(\w+)|_ZN2lp\d+([a-z0-9_]+?)I", name)
    base = (m.group(1) or m.group(2)) if m else name
    return base.replace("_kernel", "") + (f16 if "I" in name and "lp::" not in name else "")


def last_step(root, p):
    tr = glob.glob(f"{root}/pass{p}/**/*kernel_trace.csv", recursive=True)[0]
    cc = glob.glob(f"{root}/pass{p}/**/*counter_collection.csv", recursive=True)[0]
    disp = {r["Dispatch_Id"]: (r["Kernel_Name"], int(r["Start_Timestamp"])) for r in csv.DictReader(open(tr))}
    ctr = collections.defaultdict(dict)
    for r in csv.DictReader(open(cc)):
        ctr[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(disp, key=lambda k: disp[k][1])
    stems = [i for i in ids if "stem" in disp[i][0] and "cls_stem" not in disp[i][0]]
    start = disp[stems[-1]][1]
    lbs = [i for i in ids if "letterbox" in disp[i][0] and disp[i][1] < start]   # configs[4]: the step starts with the letterbox
    if lbs and (len(stems) < 2 or disp[lbs[-1]][1] > disp[stems[-2]][1]):
        start = disp[lbs[-1]][1]
    return [(disp[i][0], ctr[i]) for i in ids if disp[i][1] >= start and "lp" in disp[i][0]]


def main(root):
    out = collections.OrderedDict()
    for name, c in last_step(root, 3):
        f = out.setdefault(family(name), {"launches": 0, "hbm_read_bytes": 0.0, "hbm_write_bytes": 0.0})
        f["launches"] += 1
        f["hbm_read_bytes"] += 2.0 * c.get("FETCH_SIZE", 0.0) * 1024.0
    for name, c in last_step(root, 4):
        out[family(name)]["hbm_write_bytes"] += c.get("WRITE_SIZE", 0.0) * 1024.0
    for f in out.values():
        f["hbm_bytes_per_launch"] = (f["hbm_read_bytes"] + f["hbm_write_bytes"]) / max(f["launches"], 1)
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE (x2, gfx950 correction) / WRITE_SIZE, one 64-image step, tools/pmc_profile.sh",
               "families": out}, sys.stdout, indent=1)


if __name__ == "__main__":
    main(sys.argv[1])
