#!/bin/bash
# Round 6: (1) cost of the SAMPLED range statistics (ODT_RANGE_STATS=0: off) at b = 8 / b = 1; (2) the detect + track leg with
# the round-5 cosine kernel vs the round-6 GEMM one (ab/*.so), same box.
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in 0 1 0 1; do
  (ODT_RANGE_STATS=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-d7 --no-extras --no-cpu-baseline --no-nn-matching 2>/dev/null | tail -1) > gpurun_out/r06f_bench_b8_$v.json
  python -c "
import json; d=json.load(open('gpurun_out/r06f_bench_b8_$v.json')); r=d['roofline']
print('RANGE_STATS=$v b8 FPS %.2f  frac %.4f  verified %s  watch %s' % (d['value'], r['frac'], d['verified'], d['handle'].get('conv_split_family_auto', {}).get('watch')))"
done 2>&1 | tee gpurun_out/r06f_range_stats_ab.txt
for v in 0 1 0 1; do
  (ODT_RANGE_STATS=$v timeout 300 python bench.py --batch 1 --graph single --steps 40 --warmup 5 --no-d7 --no-extras --no-cpu-baseline 2>/dev/null | tail -1) > gpurun_out/r06f_bench_b1_$v.json
  python -c "
import json; d=json.load(open('gpurun_out/r06f_bench_b1_$v.json'))
print('RANGE_STATS=$v b1 FPS %.2f verified %s' % (d['value'], d['verified']))"
done 2>&1 | tee -a gpurun_out/r06f_range_stats_ab.txt
keep=/tmp/keep_lib.so; cp object_detection_tracking_amd/libodt_hip.so $keep
for v in r06_old_cosine r06_cur r06_old_cosine r06_cur; do
  cp ab/$v.so object_detection_tracking_amd/libodt_hip.so
  echo "[$v] $(timeout 300 python tools/detect_track_ab.py 2>/dev/null | tail -1)"
done 2>&1 | tee gpurun_out/r06f_detect_track_cosine_ab.txt
cp $keep object_detection_tracking_amd/libodt_hip.so
timeout 600 python -m pytest tests/test_e2e.py -q -m gpu -x -k "continuous_range_guard" 2>&1 | tail -3
