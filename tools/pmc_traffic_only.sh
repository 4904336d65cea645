#!/bin/bash
# The two traffic passes of tools/pmc_profile.sh alone (FETCH_SIZE; WRITE_SIZE + TCC hit / miss), with the environment as given:
#   LITEPI_BNECK_CL=1 tools/pmc_traffic_only.sh <outdir-under-gpurun_out> [extra bench args]   ->   tools/pmc_traffic.py <outdir>
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/${1:-pmc}
shift || true
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 0 --no-h2d --no-dropin --inflight 1 $*"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pass3" -- $BENCH > "$OUT/pass3.log" 2>&1 || { echo "pass3 failed"; tail -5 "$OUT/pass3.log"; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$OUT/pass4" -- $BENCH > "$OUT/pass4.log" 2>&1 || { echo "pass4 failed"; tail -5 "$OUT/pass4.log"; exit 1; }
cd "$R" && python3 tools/pmc_traffic.py "$OUT" > "$OUT/traffic.json" && python3 - "$OUT/traffic.json" <<'PY'
import json, sys
for k, v in json.load(open(sys.argv[1]))["families"].items():
    if "bottleneck" in k or "stem" in k:
        print("%-44s read %7.1f MB  write %7.1f MB" % (k, v["hbm_read_bytes"] / 1e6, v["hbm_write_bytes"] / 1e6))
PY
