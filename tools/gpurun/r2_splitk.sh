#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops.py tests/test_e2e.py -q -m gpu -x -k "split or forward or arithmetic or determinism" 2>&1 | tail -3
for v in "ODT_CONV_SPLIT3_SPLITK=1" "ODT_CONV_SPLIT3_SPLITK=8"; do
  r1=$(env $v timeout 300 python bench.py --batch 1 --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f FPS all-conv %.1f TF' % (d['value'], d['roofline']['all_conv_launches']['achieved']), d['handle']['split_launches_by_family'])")
  r8=$(env $v timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f FPS all-conv %.1f TF' % (d['value'], d['roofline']['all_conv_launches']['achieved']), d['handle']['split_launches_by_family'])")
  echo "$v  b1: $r1 | b8: $r8"
done | tee gpurun_out/splitk_ab.txt
(timeout 300 python tools/profile_layers.py --batch 1 --steps 3 2>&1 | tail -45) > gpurun_out/layers_b1.txt 2>&1
head -24 gpurun_out/layers_b1.txt
