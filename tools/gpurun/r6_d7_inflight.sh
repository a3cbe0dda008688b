#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
for q in 8 16 24; do
echo "GPU_MAX_HW_QUEUES=$q $(GPU_MAX_HW_QUEUES=$q timeout 900 python tools/d7_frames_in_flight.py 2>/dev/null | tail -1)"
done | tee gpurun_out/r06_d7_frames_in_flight_hw_queues.txt
