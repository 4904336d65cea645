set -u
cd $GRAFT_REPO_ROOT
S1='c2f<64,1,s2+256>'
S2='c2f<64,1,s2+256>;c2f<64,1,s2+128,sppf>'
LITEPI_C2F_SKIP="$S2" LITEPI_C2F_STORE_ALL=1 timeout -k 10 240 python tools/c2f_check.py v1 3 > gpurun_out/v1s_check.txt 2>&1; grep -E "c2f<64|s2conv<64|sppf" gpurun_out/v1s_check.txt | head -8; tail -3 gpurun_out/v1s_check.txt
NSHOW=0 bash tools/env_bench.sh v1 v1s - "LITEPI_C2F_SKIP=$S1" "LITEPI_C2F_SKIP=$S2" - "LITEPI_C2F_SKIP=$S1" > gpurun_out/v1s.log 2>&1; grep -E "img/s" gpurun_out/v1s.log
for e in "" "$S1" "$S2"; do
  LITEPI_C2F_SKIP="$e" python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-dropin --no-h2d --windows 3 --profile-steps 0 --inflight 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('inflight 1 [$e]: %.0f img/s  %.4f ms/step' % (d['value'], d['ms_per_step']))"
done
python - <<PY
import json
for i in (0,1,2):
  print([("%s %.1f" % (l["name"], l["ms"]*1e3)) for l in json.load(open("gpurun_out/v1s_%d_launches.json" % i))["launches"] if ("c2f<64" in l["name"] or "s2conv<64" in l["name"] or "sppf" in l["name"] or "conv1x1" in l["name"])])
PY
