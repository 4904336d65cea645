#!/bin/bash
# v2 (the paper's YOLO-LitePi widths): parity of the fused head + A/B of the plans, one gpurun call
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wide_towers or (bench_configuration and v2) or (detector_fp16_out0 and v2)" > gpurun_out/v2_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/v2_tests.log
cat > /tmp/_line.py <<'PY'
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d["roofline"]
print("%-40s %7.0f img/s  median %.4f ms  eager %.3f ms  launches %d  %s" % (sys.argv[1] or "(default)", d["value"], d["windows"]["ms_per_step_median"], r["profiled_step_ms"], r["profiled_launches"],
      {k: round(v * 1000, 1) for k, v in r["kernels_ms"].items() if k.startswith("head") or k.startswith("conv3x3_mfma")}), flush=True)
PY
for v in "" "LITEPI_HEADFUSE=narrow" "LITEPI_HEAD_A32=1" "" "LITEPI_HEADFUSE=narrow"; do
  ( for kv in $v; do export $kv; done; python bench.py --preset v2 --steps 30 --warmup 5 --no-cpu-baseline --no-dropin --no-h2d --windows 4 2>/dev/null | python /tmp/_line.py "$v" )
done
