for e in "ODT_CONV_SMALLK=0" "ODT_CONV_SMALLK=1"; do
  echo "--- $e"
  env $e python tools/bench_efficientdet.py --frame 1080x1920 --no-cpu-baseline --steps 12 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('D7 %.2f FPS' % d['value'])"
  env $e python bench.py --batch 1 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('FPN b1 %.2f FPS' % d['value'])"
  env $e python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('FPN b8 %.2f FPS conv %.1f TF' % (d['value'], d['roofline']['achieved']))"
done
