#!/bin/bash
# One gpurun call: numerical bisect of the whole-C2f launches, their phase stamps, and a short bench.  usage: tools/quick_ab.sh [tag]
set -u
TAG=${1:-ab}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
mkdir -p gpurun_out
timeout -k 10 200 python tools/c2f_check.py v1 3 > gpurun_out/${TAG}_check.txt 2>&1 || { echo "c2f_check failed"; tail -20 gpurun_out/${TAG}_check.txt; exit 1; }
tail -8 gpurun_out/${TAG}_check.txt
rm -f gpurun_out/${TAG}_st.txt
LITEPI_NO_GRAPH=1 LITEPI_C2F_STAMPS=gpurun_out/${TAG}_st.txt timeout -k 10 200 python tools/c2f_stamps_run.py || exit 1
python tools/c2f_stamps.py gpurun_out/${TAG}_st.txt
python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-dropin --no-h2d --windows 5 > gpurun_out/${TAG}_bench.json 2>gpurun_out/${TAG}_bench.err || { echo "bench failed"; tail -5 gpurun_out/${TAG}_bench.err; exit 1; }
python - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_bench.json").read().strip().splitlines()[-1])
print("bench: %.0f img/s, %.4f ms/step, median of windows %.4f" % (d["value"], d["ms_per_step"], d["windows"]["ms_per_step_median"]))
print({k: v for k, v in d["roofline"]["kernels_ms"].items() if k.startswith("c2f") or k.startswith("s2conv")})
PY
python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-dropin --no-h2d --windows 3 --inflight 1 --profile-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('inflight 1: %.0f img/s' % d['value'])"
