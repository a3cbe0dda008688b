#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_drop_in.py tests/test_ingest.py tests/test_abi.py -q -m gpu -x 2>&1 | tail -3
for rep in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-d7 --no-cpu-baseline --no-live-traffic 2>gpurun_out/r06ae_err.log | tail -1 | python -c "
import sys,json; d=json.load(sys.stdin); e=d['extra']; b=e['b1_single_graph']; print('value %.1f' % d['value'], 'det only', round(e.get('detector_only_fps_without_nn_matching_in_the_step'),1), 'two_streams', round(e.get('two_streams_per_gpu_fps'),1), 'b1', round(e.get('b1_single_graph_fps'),1), 'b1 frames in flight', e.get('b1_frames_in_flight_fps'), b.get('frames_in_flight'), b.get('frames_in_flight_verified'), d['verified'])"
done 2>&1 | tee gpurun_out/r06ae_b1_three_in_flight.txt
tail -2 gpurun_out/r06ae_err.log
