run() { timeout 300 python bench.py --streams $1 --batch $2 --steps $3 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f FPS  step %.2f ms' % (d['value'], d['ms_per_step']))"; }
for rep in 1 2; do
  for cfg in "1 8 10" "2 8 8" "2 4 12" "3 4 8" "1 1 40" "2 1 30" "4 1 20" "4 2 12"; do
    set -- $cfg; echo "streams=$1 batch=$2: $(run $1 $2 $3)"
  done
done
