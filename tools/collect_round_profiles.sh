#!/bin/bash
# Everything profiles/rNN_* is made from, in one gpurun call (GPU box, ~15 min).  usage: tools/collect_round_profiles.sh r04
set -u
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
python bench.py --steps 40 --warmup 8 --dump-profile "$OUT/hipevent_per_launch.json" > "$OUT/bench_n1.log" 2>&1 || { echo "bench failed"; tail -5 "$OUT/bench_n1.log"; exit 1; }
tail -1 "$OUT/bench_n1.log" > "$OUT/bench_n1.json"
echo "bench done"
python tools/latency_config1.py > "$OUT/latency_config1.txt" 2>&1 || { echo "latency failed"; exit 1; }
grep config1 "$OUT/latency_config1.txt"
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/rocprof" -- python3 "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 --windows 2 --no-h2d --no-dropin --inflight 1 > "$OUT/rocprof.log" 2>&1) || { echo "rocprof failed"; exit 1; }
echo "rocprof done"
bash tools/pmc_profile.sh "$TAG/pmc" || exit 1
python bench.py --preset v2 --steps 30 --warmup 5 --no-cpu-baseline --no-dropin --dump-profile "$OUT/hipevent_per_launch_v2.json" > "$OUT/bench_v2.log" 2> "$OUT/bench_v2.err" || { echo "v2 bench failed"; exit 1; }
tail -1 "$OUT/bench_v2.log" > "$OUT/bench_v2.json"
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/rocprof_v2" -- python3 "$R/bench.py" --preset v2 --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 --windows 2 --no-h2d --no-dropin --inflight 1 > "$OUT/rocprof_v2.log" 2>&1) || { echo "rocprof v2 failed"; exit 1; }
echo "v2 done"
# configs[4]: 2048 x 2048 frames, letterbox on the device: bench line, kernel trace, the two traffic passes
python bench.py --config 4 --steps 20 --warmup 5 --windows 4 > "$OUT/bench_config4.log" 2> "$OUT/bench_config4.err" || { echo "config4 bench failed"; tail -3 "$OUT/bench_config4.err"; exit 1; }
tail -1 "$OUT/bench_config4.log" > "$OUT/bench_config4.json"
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/rocprof_c4" -- python3 "$R/bench.py" --config 4 --steps 10 --warmup 3 --no-cpu-baseline --profile-steps 0 --windows 1 --no-h2d --no-dropin --inflight 1 > "$OUT/rocprof_c4.log" 2>&1) || { echo "rocprof c4 failed"; exit 1; }
for p in 3 4; do
  if [ $p = 3 ]; then set_="FETCH_SIZE"; else set_="WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; fi
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --pmc $set_ --kernel-trace --output-format csv -d "$OUT/pmc_c4/pass$p" -- python3 "$R/bench.py" --config 4 --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 0 --no-h2d --no-dropin --inflight 1 --windows 0 > "$OUT/pmc_c4_pass$p.log" 2>&1) || { echo "pmc c4 pass $p failed"; exit 1; }
done
echo "config4 done"
python tools/cls_arch_profile.py > "$OUT/cls_archs.txt" 2>&1 || { echo "cls archs failed"; tail -3 "$OUT/cls_archs.txt"; }
bash tools/inflight_sweep.sh > "$OUT/inflight_sweep.txt" 2>&1
bash tools/marginal_cost.sh 16 > "$OUT/marginal_cost.txt" 2>&1 || { echo "marginal cost failed"; exit 1; }
python tools/h2d_probe.py > "$OUT/h2d_probe.txt" 2>&1
echo "all done"
