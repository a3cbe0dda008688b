#!/bin/bash
# Round 6: two stages of activations in flight (conv_h2_kernel PF = 2) -- test on hip, same-box A/B of the knob mask by bench line
# and by per-layer table.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops.py -q -m gpu -x -k "two_stages_in_flight or fp16x2" 2>&1 | tail -3
for v in 0 1 3 7 0 1; do
  (ODT_CONV_H2_PF2=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-d7 --no-extras --no-cpu-baseline --no-nn-matching 2>/dev/null | tail -1) > gpurun_out/r06c_bench_pf$v.json
  python -c "
import json; d=json.load(open('gpurun_out/r06c_bench_pf$v.json')); r=d['roofline']
print('PF2=$v b8 FPS %.2f  frac %.4f  verified %s' % (d['value'], r['frac'], d['verified']))"
done 2>&1 | tee gpurun_out/r06c_pf2_ab.txt
BATCH=8 bash tools/gpurun/ab_layers_env.sh "ODT_CONV_H2_PF2=0" "ODT_CONV_H2_PF2=1" "ODT_CONV_H2_PF2=7" 2>&1 | cut -c1-150 | tee -a gpurun_out/r06c_pf2_ab.txt
