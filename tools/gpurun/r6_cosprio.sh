#!/bin/bash
# nn_matching inside the step: the cosine stream at the highest priority (default), plain, lowest
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in 1 -1 0 1 -1 0; do
  (ODT_COSINE_STREAM_PRIORITY=$v ODT_TRACKER_TIMING=1 timeout 300 python bench.py --steps 30 --warmup 5 --no-d7 --no-extras --no-cpu-baseline 2>gpurun_out/r06o_err.log | tail -1) > gpurun_out/r06o_bench_$v.json
  python -c "
import json; d=json.load(open('gpurun_out/r06o_bench_$v.json'))
print('COSINE_STREAM_PRIORITY=$v b8 with nn_matching FPS %.2f  verified %s' % (d['value'], d['verified']))"
  grep cosine gpurun_out/r06o_err.log | tail -1
done 2>&1 | tee gpurun_out/r06o_cosine_priority_ab.txt
(timeout 300 python bench.py --steps 30 --warmup 5 --no-d7 --no-extras --no-cpu-baseline --no-nn-matching 2>/dev/null | tail -1) | python -c "
import sys,json; d=json.load(sys.stdin); print('detector only FPS %.2f' % d['value'])" | tee -a gpurun_out/r06o_cosine_priority_ab.txt
