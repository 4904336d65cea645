#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
python bench.py --config 4 --steps 10 --warmup 3 --windows 1 --no-cpu-baseline --no-dropin --no-h2d > gpurun_out/bench_c4.json 2> gpurun_out/bench_c4.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_c4.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["config"]["rois_per_step_rank0"], d["config"]["boxes_pre_area_filter_rank0"], d["config"]["roi_longer_side_px"], d["config"]["detector"])
print({k: round(v * 1000, 1) for k, v in d["roofline"]["kernels_ms"].items()})
PY
