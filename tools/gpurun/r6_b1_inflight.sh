#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
for q in 8 16; do
echo "GPU_MAX_HW_QUEUES=$q $(GPU_MAX_HW_QUEUES=$q timeout 600 python tools/b1_two_in_flight.py 2>/dev/null | tail -1)"
done | tee gpurun_out/r06_b1_frames_in_flight_hw_queues.txt
