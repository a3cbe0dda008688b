#!/bin/bash
# side streams (tail / H2D / D2H / tracker) on priority queues vs plain: headline step, pipelined ingest, detect+track, in bench.py's own process history
mkdir -p gpurun_out
for p in 0 1; do
  echo "== ODT_SIDE_STREAM_PRIORITY=$p ODT_COSINE_STREAM_PRIORITY=$p"
  ODT_SIDE_STREAM_PRIORITY=$p ODT_COSINE_STREAM_PRIORITY=$p timeout 420 python bench.py --steps 15 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); e=d['extra']
print('b8 %.1f FPS | pipelined %.1f | two handles %.1f | detect_track %.1f (host %.2f ms) | arrays %.1f (host %.2f ms)' % (d['value'], e['pcie_inclusive_pipelined_fps'], e['two_streams_per_gpu_fps'], e['detect_track_fps'], e['detect_track']['host_tracking_ms_per_frame'], e['detect_track_arrays_fps'], e['detect_track_arrays_host_ms_per_frame']))"
done | tee gpurun_out/streams_ab.txt
