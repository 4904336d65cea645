#!/bin/bash
# images/s for 1..4 steps in flight on one box (same bench, no extras).  usage: tools/inflight_sweep.sh [preset]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
for n in 1 2 3 4; do
  python bench.py --preset ${1:-v1} --steps 40 --warmup 8 --no-cpu-baseline --no-dropin --no-h2d --windows 3 --profile-steps 0 --inflight $n 2>/dev/null |
    python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('inflight $n: %.0f img/s  %.4f ms/step' % (d['value'], d['ms_per_step']))"
done
