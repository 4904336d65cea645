#!/usr/bin/env python3
"""Adds `rocprof_avg_launch_ms` (average kernel duration per family from a `rocprofv3 --kernel-trace --stats` run of the same
bench command) to a pmc_traffic JSON, so that bench.py can print it beside its own HIP-event figure.

    python tools/rocprof_family_avg.py <kernel_stats.csv> <pmc_traffic.json>   (rewrites the JSON in place)
"""
import collections
import csv
import json
import sys

from pmc_traffic import family


def main(stats_csv, traffic_json):
    tot = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(stats_csv)):
        name = r["Name"]
        if "lp" not in name:
            continue
        f = tot[family(name)]
        f[0] += int(r["Calls"])
        f[1] += float(r["TotalDurationNs"])
    d = json.load(open(traffic_json))
    for k, v in d["families"].items():
        if k in tot and tot[k][0]:
            v["rocprof_avg_launch_ms"] = tot[k][1] / tot[k][0] * 1e-6
            v["rocprof_calls"] = tot[k][0]
    d["rocprof_source"] = "rocprofv3 --kernel-trace --stats of `bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0` (kernel_stats.csv)"
    json.dump(d, open(traffic_json, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
