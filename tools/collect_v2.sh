#!/bin/bash
# v2's share of profiles/rNN_*: bench line + per-launch times, kernel trace, the FETCH / WRITE (+ SQ) passes, marginal costs.  usage: tools/collect_v2.sh r04c
set -u
TAG=${1:-r04c}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
python bench.py --preset v2 --steps 30 --warmup 5 --no-cpu-baseline --no-dropin --dump-profile "$OUT/hipevent_per_launch_v2.json" > "$OUT/bench_v2.log" 2> "$OUT/bench_v2.err" || { echo "v2 bench failed"; exit 1; }
tail -1 "$OUT/bench_v2.log" > "$OUT/bench_v2.json"
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/rocprof_v2" -- python3 "$R/bench.py" --preset v2 --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 --windows 2 --no-h2d --no-dropin --inflight 1 > "$OUT/rocprof_v2.log" 2>&1) || { echo "rocprof v2 failed"; exit 1; }
bash tools/pmc_profile.sh "$TAG/pmc_v2" --preset v2 > "$OUT/pmc_v2.log" 2>&1 || exit 1
PRESET=v2 bash tools/marginal_cost.sh 20 > "$OUT/marginal_cost_v2.txt" 2>&1 || { echo "marginal failed"; exit 1; }
bash tools/inflight_sweep.sh v2 > "$OUT/inflight_sweep_v2.txt" 2>&1
echo "v2 collected"
