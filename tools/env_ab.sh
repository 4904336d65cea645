#!/bin/bash
# Generic same-box A/B: tools/env_ab.sh "VAR=1" "" "VAR=1" ""  -> one short bench per argument (an empty string = defaults),
# printing the pipelined rate, one-step-in-flight rate and the eager times of the kernels matching $KERNELS (default: all > 15 us)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
cat > /tmp/_line.py <<'PY'
import json, os, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d["roofline"]
pat = os.environ.get("KERNELS", "")
ks = {k: round(v * 1000, 1) for k, v in r["kernels_ms"].items() if (pat and any(p in k for p in pat.split(","))) or (not pat and v > 0.015)}
print("%-28s %7.0f img/s  median %.4f ms  dropin %s  eager step %.3f ms  %s" % (sys.argv[1] or "(default)", d["value"], d["windows"]["ms_per_step_median"],
      ("%.0f" % d["dropin"]["value"]) if d.get("dropin") else "-", r["profiled_step_ms"], ks), flush=True)
PY
for v in "$@"; do
  ( [ -n "$v" ] && export $v; python bench.py --steps 40 --warmup 8 --no-cpu-baseline ${DROPIN:---no-dropin} --no-h2d --windows 5 2>/dev/null | python /tmp/_line.py "$v" )
done
