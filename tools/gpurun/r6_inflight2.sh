#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_drop_in.py -q -m gpu -x -k "predict_stream or product_default" 2>&1 | tail -3
(timeout 600 python bench.py --steps 20 --warmup 5 --no-d7 --no-cpu-baseline --no-live-traffic 2>gpurun_out/r06aa_err.log | tail -1) > gpurun_out/r06aa_bench.json
python -c "
import json; d=json.load(open('gpurun_out/r06aa_bench.json')); e=d['extra']; b=e['b1_single_graph']
print('value %.1f' % d['value'], 'b1', round(e['b1_single_graph_fps'],1), b['verified'], 'two in flight', b.get('two_frames_in_flight_fps'), b.get('two_frames_in_flight_verified'))"
tail -2 gpurun_out/r06aa_err.log
