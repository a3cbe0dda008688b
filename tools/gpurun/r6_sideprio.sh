#!/bin/bash
# with eight hardware queues per priority: side streams (tail, H2D, D2H) at the highest priority (default) vs plain
mkdir -p gpurun_out
export TMPDIR=/tmp
for cfg in "ODT_SIDE_STREAM_PRIORITY=1" "ODT_SIDE_STREAM_PRIORITY=0" "ODT_SIDE_STREAM_PRIORITY=1" "ODT_SIDE_STREAM_PRIORITY=0"; do
env $cfg timeout 600 python bench.py --steps 20 --warmup 5 --no-d7 --no-cpu-baseline --no-live-traffic 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.load(sys.stdin); e=d['extra']; print('[$cfg] value %.1f' % d['value'], 'det only', round(e.get('detector_only_fps_without_nn_matching_in_the_step'),1), 'detect_track', round(e.get('detect_track_fps'),1), 'arrays', round(e.get('detect_track_arrays_fps'),1), 'pipelined', round(e.get('pcie_inclusive_pipelined_fps'),1), 'pinned', round(e.get('pcie_inclusive_pipelined_pinned_source_fps'),1), 'two_streams', round(e.get('two_streams_per_gpu_fps'),1), 'b1', round(e.get('b1_single_graph_fps'),1))"
done 2>&1 | tee gpurun_out/r06r_side_stream_priority_ab.txt
