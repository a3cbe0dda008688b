# A/B of environment knobs in one box.  usage: bash tools/gpurun/ab_env.sh "ENV_A" "ENV_B"
run() { env $1 timeout 300 python bench.py $EXTRA --batch $2 --steps $3 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f FPS conv %.1f TF' % (d['value'], d['roofline']['achieved']))"; }
for rep in 1 2; do for v in "$@"; do
  echo "[$v] rep$rep  b8: $(run "$v" 8 10) | b1: $(run "$v" 1 30)"
done; done
