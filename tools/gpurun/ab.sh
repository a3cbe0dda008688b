# A/B in one box: alternate the two library builds
for rep in 1 2; do for v in $VARIANTS; do
  cp ab/$v.so object_detection_tracking_amd/libodt_hip.so
  r8=$(timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f FPS conv %.1f TF' % (d['value'], d['roofline']['achieved']))")
  r1=$(timeout 300 python bench.py --batch 1 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f FPS conv %.1f TF' % (d['value'], d['roofline']['achieved']))")
  echo "$v rep$rep  b8: $r8 | b1: $r1"
done; done
