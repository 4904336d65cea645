#!/bin/bash
# PMC counter passes for the bench step (run on the GPU box through gpurun).
# Counters are collected in separate rocprofv3 runs with --kernel-trace only (MI355X_MICROARCH.md
# "rocprofv3 PMC slots": 8 SQ slots, FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2).
#   usage: tools/pmc_profile.sh <outdir-under-gpurun_out> [extra bench args]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/${1:-pmc}
shift || true
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 0 --no-h2d --no-dropin --inflight 1 $*"
i=0
for set in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
  "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE" \
  "FETCH_SIZE" \
  "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/pass$i" -- $BENCH > "$OUT/pass$i.log" 2>&1
  echo "pass$i rc=$?"
done
ls -R "$OUT" | grep -c csv
