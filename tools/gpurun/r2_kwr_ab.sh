#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops.py -q -m gpu -k "split" -x 2>&1 | tail -3
BATCH=8 bash tools/gpurun/ab_layers_env.sh "ODT_CONV_SPLIT3_KWR_EARLY=1" "ODT_CONV_SPLIT3_KWR=1" > gpurun_out/kwr_layers_b8.txt 2>&1
head -22 gpurun_out/kwr_layers_b8.txt
for v in "ODT_CONV_SPLIT3_KWR_EARLY=1" "ODT_CONV_SPLIT3_KWR=1" "ODT_CONV_SPLIT3_KWR=0"; do
  r8=$(env $v timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f FPS all-conv %.1f TF split %.1f TF' % (d['value'], d['roofline']['all_conv_launches']['achieved'], d['roofline']['achieved']))")
  echo "$v  b8: $r8"
done | tee gpurun_out/kwr_ab.txt
