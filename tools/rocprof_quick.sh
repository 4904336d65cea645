#!/bin/bash
# rocprofv3 --kernel-trace --stats of a short single-handle bench; prints the kernel table.  usage: tools/rocprof_quick.sh tag [ENV=..]
TAG=${1:-q}; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/rocprof_$TAG
mkdir -p "$OUT"
for v in "$@"; do export $v; done
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 --windows 2 --no-h2d --no-dropin --inflight 1 > "$OUT/log.txt" 2>&1) || { echo "rocprof failed"; tail -5 "$OUT/log.txt"; exit 1; }
f=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
echo "== $TAG $*"
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:24]:
    print("%-90s calls %4s avg %8.1f us  min %8.1f  max %8.1f" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
