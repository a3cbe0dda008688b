#!/bin/bash
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/ -q -m gpu -x 2>&1 | tail -5 | tee gpurun_out/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log
timeout 600 python bench.py --steps 20 --warmup 3 2>gpurun_out/bench_err.log | tail -1 > gpurun_out/bench_b8.json; cat gpurun_out/bench_b8.json | cut -c1-600
