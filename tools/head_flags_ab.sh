#!/bin/bash
# A/B of the head kernel's experiment switches (LITEPI_HEAD_FLAGS) in one gpurun call: eager head launch times + pipelined rate.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
cat > /tmp/_line.py <<'PY'
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d["roofline"]
print(sys.argv[1], round(d["value"]), "img/s, median of windows", round(d["windows"]["ms_per_step_median"], 4), {k: round(v * 1000, 1) for k, v in r["kernels_ms"].items() if k.startswith("head")}, flush=True)
PY
for v in ${@:-0 1 2 4 5 0}; do
  export LITEPI_HEAD_FLAGS=$v
  python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-dropin --no-h2d --windows 5 2>/dev/null | python /tmp/_line.py "flags=$v"
done
