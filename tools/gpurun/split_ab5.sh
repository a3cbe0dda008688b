#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ops.py tests/test_e2e.py -q -m gpu -k "split or two_sources or b8_1080p" -x 2>&1 | tail -2
for v in "ODT_CONV_SPLIT_SRC2=0" "ODT_CONV_SPLIT_SRC2=1" "ODT_CONV_SPLIT_SRC2=0" "ODT_CONV_SPLIT_SRC2=1"; do
  r8=$(env $v timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f FPS all-conv %.1f TF' % (d['value'], d['roofline']['all_conv_launches']['achieved']))")
  echo "$v  b8: $r8"
done | tee gpurun_out/split_ab5.txt
