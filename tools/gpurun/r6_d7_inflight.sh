#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python tools/d7_frames_in_flight.py 2>gpurun_out/r06ab_err.log | tail -1 | tee gpurun_out/r06_d7_frames_in_flight.txt
tail -3 gpurun_out/r06ab_err.log
