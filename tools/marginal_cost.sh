#!/bin/bash
# Marginal cost of every launch of the step in a PIPELINED step (three in flight): the bench with launch i left out of every
# pass after a handle's first (LITEPI_SKIP_OP / LITEPI_SKIP_STAGE, diagnostics in detector.cpp / api.cpp).  GPU box, ~8 min.
# usage: [PRESET=v2] tools/marginal_cost.sh [n_ops=16] > gpurun_out/marginal.txt
set -u
N=${1:-16}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
ARGS="--preset ${PRESET:-v1} --steps 40 --warmup 8 --no-cpu-baseline --no-h2d --no-dropin --profile-steps 0 --windows 3"
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-12s %8.1f img/s  %.4f ms/step (median of windows %.4f)' % (sys.argv[1], d['value'], d['ms_per_step'], d['windows']['ms_per_step_median']))" "$1"; }
python bench.py $ARGS 2>/dev/null | line base
for i in $(seq 0 $((N-1))); do
  LITEPI_SKIP_OP=$i timeout -k 10 120 python bench.py $ARGS 2>/dev/null | line "op$i" || exit 1
done
for s in nms roi cls; do
  LITEPI_SKIP_STAGE=$s timeout -k 10 120 python bench.py $ARGS 2>/dev/null | line "$s" || exit 1
done
python bench.py $ARGS 2>/dev/null | line base
