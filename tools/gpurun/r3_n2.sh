#!/bin/bash
# the N > 1 launch path with real GPU work on the final code: two ranks sharing ONE GPU over gloo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python bench.py --gpus 2 --dist-backend gloo --device 0 --steps 8 --warmup 2 2>gpurun_out/r03_n2_err.log | tail -1 > gpurun_out/r03_bench_n2_gloo_one_gpu.json
python -c "
import json; d=json.load(open('gpurun_out/r03_bench_n2_gloo_one_gpu.json'))
print('n_gpus', d['n_gpus'], 'ranks_seen', d['ranks_seen'], 'total FPS %.1f' % d['value'], 'verified', d['verified'], 'frac', d['roofline']['frac'])
for p in d['per_rank']: print(p['rank'], p['affinity'].get('cpu_range'), p['affinity'].get('bound'), round(p.get('device_resident_fps') or 0, 1), round(p.get('pcie_inclusive_pipelined_fps') or 0, 1), round(p.get('pcie_inclusive_pipelined_pinned_source_fps') or 0, 1))"
tail -3 gpurun_out/r03_n2_err.log
