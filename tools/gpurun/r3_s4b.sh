#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/bench_efficientdet.py --no-cpu-baseline --steps 10 2>&1 | tail -15 > gpurun_out/r3_s4b_d7_err.log; cat gpurun_out/r3_s4b_d7_err.log | cut -c1-600
timeout 900 python -m pytest tests/test_e2e.py -q -m gpu -x -k "b16" 2>&1 | grep -E "AssertionError|assert|Error" | head -8 | cut -c1-900 | tee gpurun_out/r3_s4b_b16.log
