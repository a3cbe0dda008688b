#!/bin/bash
# Round 6: continuous range guard -- tests on hip (scene-cut tests, guard / ingest / drop-in suites), cost A/B of the producers'
# counters at b = 8 and b = 1 (ODT_RANGE_STATS=0: off), the bench line with the steady-state pipelined rate.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_e2e.py tests/test_ingest.py tests/test_drop_in.py tests/test_abi.py -q -m gpu -x -k "guard or auto or ingest or drop or abi or trained_like or stream or describe" 2>&1 | tail -4 | tee gpurun_out/r06e_pytest_guard.log
for v in 0 1 0 1; do
  (ODT_RANGE_STATS=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-d7 --no-extras --no-cpu-baseline --no-nn-matching 2>/dev/null | tail -1) > gpurun_out/r06e_bench_b8_$v.json
  python -c "
import json; d=json.load(open('gpurun_out/r06e_bench_b8_$v.json')); r=d['roofline']
print('RANGE_STATS=$v b8 FPS %.2f  frac %.4f  verified %s  watch %s' % (d['value'], r['frac'], d['verified'], d['handle'].get('conv_split_family_auto', {}).get('watch')))"
done 2>&1 | tee gpurun_out/r06e_range_stats_ab.txt
for v in 0 1; do
  (ODT_RANGE_STATS=$v timeout 300 python bench.py --batch 1 --graph single --steps 40 --warmup 5 --no-d7 --no-extras --no-cpu-baseline 2>/dev/null | tail -1) > gpurun_out/r06e_bench_b1_$v.json
  python -c "
import json; d=json.load(open('gpurun_out/r06e_bench_b1_$v.json'))
print('RANGE_STATS=$v b1 FPS %.2f verified %s' % (d['value'], d['verified']))"
done 2>&1 | tee -a gpurun_out/r06e_range_stats_ab.txt
(timeout 600 python bench.py --steps 20 --warmup 5 --no-d7 --no-cpu-baseline 2>gpurun_out/r06e_bench_err.log | tail -1) > gpurun_out/r06e_bench_n1.json
python -c "
import json; d=json.load(open('gpurun_out/r06e_bench_n1.json')); e=d['extra']
print('value %.2f' % d['value'], {k: (round(v, 2) if isinstance(v, float) else v) for k, v in e.items() if not isinstance(v, dict)})"
