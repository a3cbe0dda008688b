#!/bin/bash
# Round 6: res5 conv2 (dilated 3x3, M = 16320, N = 512) on the kw-reuse kernel's 256 x 128 tiles without split-K (one round of 256)
# instead of the generic kernel's 128 x 128 tiles: ODT_CONV_H2K_FEWROWS A/B
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in 1 2 1 2; do
  (ODT_CONV_H2K_FEWROWS=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-d7 --no-extras --no-cpu-baseline --no-nn-matching 2>/dev/null | tail -1) > gpurun_out/r06k_bench_$v.json
  python -c "
import json; d=json.load(open('gpurun_out/r06k_bench_$v.json')); r=d['roofline']
print('FEWROWS=$v b8 FPS %.2f  frac %.4f  verified %s' % (d['value'], r['frac'], d['verified']))"
done 2>&1 | tee gpurun_out/r06k_fewrows2_ab.txt
BATCH=8 bash tools/gpurun/ab_layers_env.sh "ODT_CONV_H2K_FEWROWS=1" "ODT_CONV_H2K_FEWROWS=2" 2>&1 | cut -c1-150 | grep "group3\|conv total\|layer \|p5" | tee -a gpurun_out/r06k_fewrows2_ab.txt
