cd $GRAFT_REPO_ROOT
for hf in "" all; do
  LITEPI_HEADFUSE=$hf python bench.py --preset v2 --steps 30 --warmup 5 --no-cpu-baseline --no-dropin --no-h2d --windows 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('HEADFUSE=$hf', round(d['value']), d['ms_per_step'], r['profiled_launches'], round(r['profiled_step_ms'],3)); print({k:round(v*1000,1) for k,v in r['kernels_ms'].items() if 'head' in k or 'decode' in k})"
done
