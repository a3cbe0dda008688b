#!/bin/bash
# split tile variants: tests, then bench A/B of ODT_CONV_SPLIT_MINBN / MINTILES in one box
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ops.py -q -m gpu -k "split" -x > gpurun_out/split_tests.log 2>&1
tail -2 gpurun_out/split_tests.log
for v in "ODT_CONV_SPLIT_MINBN=256" "ODT_CONV_SPLIT_MINBN=128" "ODT_CONV_SPLIT_MINBN=64" "ODT_CONV_SPLIT_MINTILES=256" "ODT_CONV_SPLIT_MINBN=256"; do
  r8=$(env $v timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f FPS all-conv %.1f TF' % (d['value'], d['roofline']['all_conv_launches']['achieved']))")
  echo "$v  b8: $r8"
done | tee gpurun_out/split_ab2.txt
BATCH=8 timeout 300 python tools/profile_layers.py --batch 8 --steps 3 2>&1 | tail -40 > gpurun_out/split_layers_b8_v10.txt
head -30 gpurun_out/split_layers_b8_v10.txt
