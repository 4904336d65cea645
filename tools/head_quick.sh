#!/bin/bash
# head parity tests + eager head times + pipelined rate + stamps, one gpurun call.  usage: tools/head_quick.sh [flags...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_real_weights.py tests/test_gpu_device_path.py -x -q -m gpu -k "detector_fp16_out0 or head_projection or bench_configuration or real_weights or pipeline_fp16 or device_path or capacity_128 or wide_towers" > gpurun_out/hq_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/hq_tests.log
cat > /tmp/_line.py <<'PY'
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d["roofline"]
print(sys.argv[1], round(d["value"]), "img/s, median of windows", round(d["windows"]["ms_per_step_median"], 4), {k: round(v * 1000, 1) for k, v in r["kernels_ms"].items() if k.startswith("head")}, flush=True)
PY
for v in ${@:-0 0}; do
  export LITEPI_HEAD_FLAGS=$v
  python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-dropin --no-h2d --windows 5 2>/dev/null | python /tmp/_line.py "flags=$v"
done
unset LITEPI_HEAD_FLAGS
python tools/head_stamps.py v1 2>/dev/null | head -12
