# b = 1 same-box A/B of environment knobs on the single-image graph (BASELINE config #2) + per-layer tables
# usage: [TESTS="-k expr files"] bash tools/gpurun/r4_b1_ab.sh "ENV_A" "ENV_B" ...
mkdir -p gpurun_out
if [ -n "$TESTS" ]; then timeout 900 python -m pytest -x -q -m gpu $TESTS 2>&1 | tail -5; fi
run() { env $1 timeout 300 python bench.py --batch 1 --graph single --no-extras --no-cpu-baseline --no-d7 --steps 40 --warmup 5 2>gpurun_out/bench_err.txt | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%.2f FPS  %.3f ms  frac %.3f verified %s' % (d['value'], d['ms_per_step'], r['frac'], d.get('verified')))" || tail -5 gpurun_out/bench_err.txt; }
for rep in 1 2; do for v in "$@"; do echo "[$v] rep$rep  b1: $(run "$v")"; done; done
i=0
for v in "$@"; do env $v timeout 300 python tools/profile_layers.py --batch 1 --steps 5 2>&1 | tail -45 > gpurun_out/layers_b1_env_$i.txt; echo "== $v"; head -12 gpurun_out/layers_b1_env_$i.txt | cut -c1-110; tail -1 gpurun_out/layers_b1_env_$i.txt; i=$((i+1)); done
