for v in $VARIANTS; do
  cp ab/$v.so object_detection_tracking_amd/libodt_hip.so
  echo "#### $v"
  (timeout 300 python bench.py --steps 150 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f FPS conv %.1f TF' % (d['value'], d['roofline']['achieved']))") &
  BP=$!
  sleep 9
  for i in 1 2 3 4; do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Socket Power\|mclk" | tr '\n' ' '; echo; sleep 1.5; done
  wait $BP
done
