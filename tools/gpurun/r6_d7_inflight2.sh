#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_efficientnet.py -q -m gpu -x -k "predict_stream or d7" 2>&1 | tail -3
for rep in 1 2; do
timeout 900 python tools/bench_efficientdet.py --no-cpu-baseline --steps 20 --warmup 3 2>gpurun_out/r06ac_err.log | tail -1 > gpurun_out/r06_bench_efficientdet_d7.json
python -c "
import json; d=json.load(open('gpurun_out/r06_bench_efficientdet_d7.json')); x=d['extra']
print('D7 value %.2f verified %s' % (d['value'], d['verified']), 'in flight', {k: (round(v['fps'],1), v['verified']) for k, v in x['frames_in_flight'].items() if isinstance(v, dict)}, 'tmot', round(x['detect_tmot_fps'],1), 'pipelined', round(x['detect_tmot_pipelined_fps'],1), 'three in flight', round(x.get('detect_tmot_three_in_flight_fps',0),1), 'host_to_host_ms', round(x['host_to_host_ms'],2))"
done
tail -2 gpurun_out/r06ac_err.log
