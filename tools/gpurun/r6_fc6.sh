#!/bin/bash
# Round 6: fc6 on the fp16x2 kernels (ROIAlign records the RoI features' range): parity tests on hip, ODT_ROI_AMAX A/B at b = 8 / b = 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_e2e.py tests/test_ops.py -q -m gpu -x -k "1080p or roi or forward or parity or head" 2>&1 | tail -3
for v in 0 1 0 1; do
  (ODT_ROI_AMAX=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-d7 --no-extras --no-cpu-baseline --no-nn-matching 2>/dev/null | tail -1) > gpurun_out/r06l_bench_$v.json
  python -c "
import json; d=json.load(open('gpurun_out/r06l_bench_$v.json')); r=d['roofline']; h=d['handle']
print('ROI_AMAX=$v b8 FPS %.2f  frac %.4f  verified %s  fp16x2 %d bf16x3 %d' % (d['value'], r['frac'], d['verified'], h['fp16x2_split_launches'], h['bf16x3_split_launches']))"
done 2>&1 | tee gpurun_out/r06l_fc6_fp16x2_ab.txt
for v in 0 1 0 1; do
  (ODT_ROI_AMAX=$v timeout 300 python bench.py --batch 1 --graph single --steps 40 --warmup 5 --no-d7 --no-extras --no-cpu-baseline 2>/dev/null | tail -1) > gpurun_out/r06l_bench_b1_$v.json
  python -c "
import json; d=json.load(open('gpurun_out/r06l_bench_b1_$v.json')); h=d['handle']
print('ROI_AMAX=$v b1 FPS %.2f verified %s fp16x2 %d bf16x3 %d' % (d['value'], d['verified'], h['fp16x2_split_launches'], h['bf16x3_split_launches']))"
done 2>&1 | tee -a gpurun_out/r06l_fc6_fp16x2_ab.txt
BATCH=8 bash tools/gpurun/ab_layers_env.sh "ODT_ROI_AMAX=0" "ODT_ROI_AMAX=1" 2>&1 | cut -c1-150 | grep "fastrcnn\|conv total\|layer " | tee -a gpurun_out/r06l_fc6_fp16x2_ab.txt
