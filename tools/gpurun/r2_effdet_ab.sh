#!/bin/bash
# EfficientDet-D7 A/B over an environment knob: bench_efficientdet FPS per value.  usage: r2_effdet_ab.sh VAR v1 v2 ...
mkdir -p gpurun_out
VAR=$1; shift
for v in "$@"; do
  r=$(env $VAR=$v timeout 300 python tools/bench_efficientdet.py --no-cpu-baseline --steps 20 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f FPS  %.2f ms  frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))")
  echo "$VAR=$v  $r"
done | tee gpurun_out/effdet_ab.txt
