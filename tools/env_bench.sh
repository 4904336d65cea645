#!/bin/bash
# A/B of environment switches on one box: tools/env_bench.sh <preset> <tag> "ENV=1 ENV2=x" "..." ("-" = no switches).  Prints images/s and the per-launch eager times of the first run.
set -u
PRESET=$1; TAG=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; mkdir -p gpurun_out
i=0
for e in "$@"; do
  [ "$e" = "-" ] && e=""
  env $e timeout -k 10 300 python bench.py --preset $PRESET --steps 30 --warmup 5 --no-cpu-baseline --no-dropin --no-h2d --windows 4 --dump-profile gpurun_out/${TAG}_${i}_launches.json > gpurun_out/${TAG}_${i}.json 2> gpurun_out/${TAG}_${i}.err || { echo "bench [$e] failed"; tail -5 gpurun_out/${TAG}_${i}.err; exit 1; }
  python - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_${i}.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("[%s]: %.0f img/s, %.4f ms/step, median %.4f; %d launches, eager %.3f ms" % ("$e", d["value"], d["ms_per_step"], d["windows"]["ms_per_step_median"], r["profiled_launches"], r["profiled_step_ms"]))
for l in json.load(open("gpurun_out/${TAG}_${i}_launches.json"))["launches"][:${NSHOW:-3}]:
    print("   %-40s %-36s %7.1f us" % (l["name"], l["layer"], l["ms"] * 1e3))
PY
  i=$((i+1))
done
