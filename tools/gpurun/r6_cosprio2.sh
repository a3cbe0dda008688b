#!/bin/bash
# the whole bench (detect + track legs, two streams, pipelined) with the cosine / tracker stream at the highest (1) vs lowest (-1) priority
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in 1 -1 1 -1; do
ODT_COSINE_STREAM_PRIORITY=$v timeout 600 python bench.py --steps 20 --warmup 5 --no-d7 --no-cpu-baseline --no-live-traffic 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.load(sys.stdin); e=d['extra']; print('COSINE_STREAM_PRIORITY=$v value %.1f' % d['value'], 'det only', round(e.get('detector_only_fps_without_nn_matching_in_the_step'),1), 'detect_track', round(e.get('detect_track_fps'),1), round(e.get('detect_track',{}).get('host_tracking_ms_per_frame'),2), 'arrays', round(e.get('detect_track_arrays_fps'),1), round(e.get('detect_track_arrays_host_ms_per_frame'),2), 'pipelined', round(e.get('pcie_inclusive_pipelined_fps'),1), 'two_streams', round(e.get('two_streams_per_gpu_fps'),1), 'b1', round(e.get('b1_single_graph_fps'),1))"
done 2>&1 | tee gpurun_out/r06p_cosine_priority_full_ab.txt
