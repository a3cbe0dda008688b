#!/bin/bash
# handles on non-blocking streams + stream-ordered output copies: whole GPU suite, bench legs, D7 legs
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/ -q -m gpu -x 2>&1 | tail -4 | tee gpurun_out/r06ad_pytest_gpu.log
for rep in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-d7 --no-cpu-baseline --no-live-traffic 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.load(sys.stdin); e=d['extra']; print('value %.1f' % d['value'], 'det only', round(e.get('detector_only_fps_without_nn_matching_in_the_step'),1), 'pcie blocking', round(e.get('pcie_inclusive_fps'),1), 'pipelined', round(e.get('pcie_inclusive_pipelined_fps'),1), 'detect_track', round(e.get('detect_track_fps'),1), 'arrays', round(e.get('detect_track_arrays_fps'),1), 'two_streams', round(e.get('two_streams_per_gpu_fps'),1), 'b1', round(e.get('b1_single_graph_fps'),1), 'b1 two in flight', round(e.get('b1_two_frames_in_flight_fps'),1), d['verified'], e['b1_single_graph']['two_frames_in_flight_verified'])"
done 2>&1 | tee gpurun_out/r06ad_bench_legs.txt
timeout 900 python tools/bench_efficientdet.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.load(sys.stdin); x=d['extra']
print('D7 value %.2f (frames in flight %d) verified %s' % (d['value'], x['value_frames_in_flight'], d['verified']), 'one at a time', round(x['one_frame_at_a_time_fps'],1), {k: (round(v['fps'],1), v['verified']) for k, v in x['frames_in_flight'].items() if isinstance(v, dict)}, 'tmot', round(x['detect_tmot_fps'],1), 'pipelined', round(x['detect_tmot_pipelined_fps'],1), 'three in flight', round(x.get('detect_tmot_three_in_flight_fps',0),1))" | tee -a gpurun_out/r06ad_bench_legs.txt
