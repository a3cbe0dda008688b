#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "[four staging threads] $(timeout 300 python tools/submit_host_time.py 2>/dev/null | tail -1)" | tee gpurun_out/r06t_submit_host_time.txt
echo "[one thread: taskset to one CPU makes hardware_concurrency irrelevant -- use the round-5 tree] $(cd wt_r05 && cp ../tools/submit_host_time.py tools/ && timeout 300 python tools/submit_host_time.py 2>/dev/null | tail -1)" | tee -a gpurun_out/r06t_submit_host_time.txt
