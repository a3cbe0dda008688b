#!/bin/bash
# re-test (round 3 measured "no gain"): short 1x1 reductions on 128 x 128 tiles, two workgroups per CU (one's epilogue under the other's main loop)
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in 0 256 512 1024 0 512; do
  (ODT_CONV_H2S_MAXK=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-d7 --no-extras --no-cpu-baseline --no-nn-matching 2>/dev/null | tail -1) > gpurun_out/r06w_bench_$v.json
  python -c "
import json; d=json.load(open('gpurun_out/r06w_bench_$v.json')); r=d['roofline']
print('H2S_MAXK=$v b8 FPS %.2f  frac %.4f  verified %s' % (d['value'], r['frac'], d['verified']))"
done 2>&1 | tee gpurun_out/r06w_h2s_maxk_ab.txt
BATCH=8 bash tools/gpurun/ab_layers_env.sh "ODT_CONV_H2S_MAXK=0" "ODT_CONV_H2S_MAXK=512" 2>&1 | cut -c1-150 | head -34 | tee -a gpurun_out/r06w_h2s_maxk_ab.txt
