#!/bin/bash
# A/B of the head shapes with more workgroups per CU (default) against round 2's (LITEPI_HEAD_2WG=1): parity tests, bench, stamps.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_real_weights.py tests/test_gpu_device_path.py -x -q -m gpu -k "detector_fp16_out0 or head_projection or bench_configuration or real_weights or pipeline_fp16 or device_path or capacity_128" 2>&1 | tail -2
cat > /tmp/_line.py <<'PY'
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d["roofline"]
print(sys.argv[1], round(d["value"]), "img/s, median of windows", round(d["windows"]["ms_per_step_median"], 4), {k: round(v * 1000, 1) for k, v in r["kernels_ms"].items() if k.startswith("head")})
PY
for v in old new; do
  if [ $v = old ]; then export LITEPI_HEAD_2WG=1; else unset LITEPI_HEAD_2WG; fi
  python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-dropin --no-h2d --windows 5 2>/dev/null | python /tmp/_line.py $v
  python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-dropin --no-h2d --windows 3 --profile-steps 0 --inflight 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   one step in flight: %.0f img/s' % d['value'])"
done
python tools/head_stamps.py v1 2>/dev/null | head -22
