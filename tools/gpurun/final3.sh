#!/bin/bash
mkdir -p gpurun_out
(timeout 400 python bench.py --steps 10 --warmup 2 2>&1 | tail -1) > gpurun_out/bench_n1.log 2>&1
(timeout 200 python tools/profile_layers.py --batch 8 --steps 3 2>&1 | tail -45) > gpurun_out/layers_b8.log 2>&1
cut -c1-200 gpurun_out/bench_n1.log; tail -1 gpurun_out/layers_b8.log
