#!/bin/bash
# cosine / tracker stream: highest priority with 4 (default) vs 8 hardware queues per priority; lowest priority
mkdir -p gpurun_out
export TMPDIR=/tmp
for cfg in "GPU_MAX_HW_QUEUES=4 ODT_COSINE_STREAM_PRIORITY=1" "GPU_MAX_HW_QUEUES=8 ODT_COSINE_STREAM_PRIORITY=1" "GPU_MAX_HW_QUEUES=8 ODT_COSINE_STREAM_PRIORITY=-1" "GPU_MAX_HW_QUEUES=4 ODT_COSINE_STREAM_PRIORITY=1" "GPU_MAX_HW_QUEUES=8 ODT_COSINE_STREAM_PRIORITY=1"; do
env $cfg ODT_TRACKER_TIMING=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-d7 --no-cpu-baseline --no-live-traffic 2>gpurun_out/r06q_err.log | tail -1 | python -c "
import sys,json; d=json.load(sys.stdin); e=d['extra']; print('[$cfg] value %.1f' % d['value'], 'det only', round(e.get('detector_only_fps_without_nn_matching_in_the_step'),1), 'detect_track', round(e.get('detect_track_fps'),1), round(e.get('detect_track',{}).get('host_tracking_ms_per_frame'),2), 'arrays', round(e.get('detect_track_arrays_fps'),1), round(e.get('detect_track_arrays_host_ms_per_frame'),2), 'pipelined', round(e.get('pcie_inclusive_pipelined_fps'),1), 'two_streams', round(e.get('two_streams_per_gpu_fps'),1))"
grep cosine gpurun_out/r06q_err.log | head -2 | tail -1
grep cosine gpurun_out/r06q_err.log | tail -1
done 2>&1 | tee gpurun_out/r06q_hw_queues_priority_ab.txt
