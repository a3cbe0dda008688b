#!/bin/bash
mkdir -p gpurun_out
for v in "ODT_CONV_SPLIT_MINTILES=384" "ODT_CONV_SPLIT_MINTILES=256" "ODT_CONV_SPLIT_MINTILES=128" "ODT_CONV_SPLIT_MINTILES=64" "ODT_CONV_SPLIT_MINTILES=16"; do
  r8=$(env $v timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f FPS all-conv %.1f TF' % (d['value'], d['roofline']['all_conv_launches']['achieved']))")
  r1=$(env $v timeout 300 python bench.py --batch 1 --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f FPS' % d['value'])")
  echo "$v  b8: $r8 | b1: $r1"
done | tee gpurun_out/split_ab3.txt
