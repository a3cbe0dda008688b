#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/b1_two_in_flight.py 2>/dev/null | tail -1 | tee gpurun_out/r06_b1_frames_in_flight.txt
