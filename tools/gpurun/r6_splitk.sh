#!/bin/bash
# Round 6: split-K combined inside the conv kernel (last-arriving range) -- tests on hip, same-box A/B at b = 1 and b = 8.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops.py -q -m gpu -x -k "split or fp16x2 or conv" 2>&1 | tail -3
for v in 0 1 0 1; do
  (ODT_CONV_SPLITK_INKERNEL=$v timeout 300 python bench.py --batch 1 --graph single --steps 40 --warmup 5 --no-d7 --no-extras --no-cpu-baseline 2>/dev/null | tail -1) > gpurun_out/r06d_bench_b1_$v.json
  python -c "
import json; d=json.load(open('gpurun_out/r06d_bench_b1_$v.json')); r=d['roofline']
print('INKERNEL=$v b1 FPS %.2f  frac %.4f  verified %s' % (d['value'], r['frac'], d['verified']))"
done 2>&1 | tee gpurun_out/r06d_splitk_inkernel_ab.txt
for v in 0 1; do
  (ODT_CONV_SPLITK_INKERNEL=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-d7 --no-extras --no-cpu-baseline --no-nn-matching 2>/dev/null | tail -1) > gpurun_out/r06d_bench_b8_$v.json
  python -c "
import json; d=json.load(open('gpurun_out/r06d_bench_b8_$v.json')); r=d['roofline']
print('INKERNEL=$v b8 FPS %.2f  frac %.4f  verified %s' % (d['value'], r['frac'], d['verified']))"
done 2>&1 | tee -a gpurun_out/r06d_splitk_inkernel_ab.txt
BATCH=1 bash tools/gpurun/ab_layers_env.sh "ODT_CONV_SPLITK_INKERNEL=0" "ODT_CONV_SPLITK_INKERNEL=1" 2>&1 | cut -c1-150 | grep -v "rpn/head\|outputs" | head -30 | tee -a gpurun_out/r06d_splitk_inkernel_ab.txt
