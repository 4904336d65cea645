#!/bin/bash
# One gpurun call for the v2 plan: numerical bisect against the layer plan and the oracle, then the v2 bench with per-launch
# times, then A/B runs with parts of the whole-C2f plan switched off (LITEPI_C2F_SKIP / LITEPI_NO_STEMBLOCK).  usage: tools/v2_ab.sh [tag] [skip-list ...]
set -u
TAG=${1:-v2ab}
shift || true
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
mkdir -p gpurun_out
LITEPI_C2F_STORE_ALL=1 timeout -k 10 240 python tools/c2f_check.py v2 3 > gpurun_out/${TAG}_check.txt 2>&1 || { echo "c2f_check failed"; tail -20 gpurun_out/${TAG}_check.txt; exit 1; }
tail -4 gpurun_out/${TAG}_check.txt
bench() {  # name, env...
  local name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --preset v2 --steps 30 --warmup 5 --no-cpu-baseline --no-dropin --no-h2d --windows 4 --dump-profile gpurun_out/${TAG}_${name}_launches.json > gpurun_out/${TAG}_${name}.json 2> gpurun_out/${TAG}_${name}.err || { echo "bench $name failed"; tail -5 gpurun_out/${TAG}_${name}.err; return 1; }
  python - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_${name}.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("${name}: %.0f img/s, %.4f ms/step, median %.4f; %d launches, eager %.3f ms" % (d["value"], d["ms_per_step"], d["windows"]["ms_per_step_median"], r["profiled_launches"], r["profiled_step_ms"]))
PY
}
bench default || exit 1
for s in "$@"; do
  n=$(echo "$s" | tr -c 'A-Za-z0-9\n' '_')
  if [ "$s" = "nostem" ]; then bench nostem LITEPI_NO_STEMBLOCK=1 || exit 1; else bench "skip_$n" "LITEPI_C2F_SKIP=$s" || exit 1; fi
done
python - <<PY
import json
for l in json.load(open("gpurun_out/${TAG}_default_launches.json"))["launches"]:
    print("   %-40s %-36s %7.1f us" % (l["name"], l["layer"], l["ms"] * 1e3))
PY
