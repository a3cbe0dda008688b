# Round 4 same-box A/B of environment knobs, b = 8 headline only (plus optional per-layer tables).
# usage: [TESTS="-k expr files"] [LAYERS=1] [STEPS=20] bash tools/gpurun/r4_ab.sh "ENV_A" "ENV_B" ...
mkdir -p gpurun_out
if [ -n "$TESTS" ]; then
  timeout 900 python -m pytest -x -q -m gpu $TESTS 2>&1 | tail -8
fi
run() { env $1 timeout 400 python bench.py --no-extras --no-cpu-baseline --no-d7 --batch 8 --steps ${STEPS:-20} --warmup 3 2>gpurun_out/bench_err.txt | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%.2f FPS  %.3f ms  split family %.1f TF frac %.3f of_sustained %s verified %s fused %s' % (d['value'], d['ms_per_step'], r['achieved'], r['frac'], r.get('frac_of_sustained'), d.get('verified'), d.get('handle',{}).get('bottleneck_tails_fused')))" || tail -5 gpurun_out/bench_err.txt; }
for rep in 1 2; do for v in "$@"; do
  echo "[$v] rep$rep  b8: $(run "$v")"
done; done
if [ -n "$LAYERS" ]; then
  BATCH=8 bash tools/gpurun/ab_layers_env.sh "$@" 2>&1 | head -45
fi
