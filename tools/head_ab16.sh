#!/bin/bash
# A/B of stage A on 16-pixel tiles (default) against round 3's 32-pixel slots (LITEPI_HEAD_A32=1): parity tests, eager head times,
# pipelined rate, one step in flight, phase stamps.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_real_weights.py tests/test_gpu_device_path.py -x -q -m gpu -k "detector_fp16_out0 or head_projection or bench_configuration or real_weights or pipeline_fp16 or device_path or capacity_128" > gpurun_out/ab16_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/ab16_tests.log
cat > /tmp/_line.py <<'PY'
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d["roofline"]
print(sys.argv[1], round(d["value"]), "img/s, median of windows", round(d["windows"]["ms_per_step_median"], 4), {k: round(v * 1000, 1) for k, v in r["kernels_ms"].items() if k.startswith("head")}, flush=True)
PY
for v in a32 a16 a32 a16; do
  if [ $v = a32 ]; then export LITEPI_HEAD_A32=1; else unset LITEPI_HEAD_A32; fi
  python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-dropin --no-h2d --windows 5 2>/dev/null | python /tmp/_line.py $v
  python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-dropin --no-h2d --windows 3 --profile-steps 0 --inflight 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   one step in flight: %.0f img/s' % d['value'])"
done
unset LITEPI_HEAD_A32
python tools/head_stamps.py v1 2>/dev/null | head -32
