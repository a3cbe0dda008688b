#!/bin/bash
# round-end validation, part 1: GPU suite, smoke, bench (b=8 with the CPU baseline, b=1), layer table, rocprof kernel stats
mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -q --timeout 400 -p no:cacheprovider 2>&1 | tail -5) > gpurun_out/pytest_gpu.log 2>&1
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3) > gpurun_out/smoke.log 2>&1
(timeout 900 python bench.py --steps 10 --warmup 2 2>&1 | tail -1) > gpurun_out/bench_n1.log 2>&1
(timeout 300 python bench.py --batch 1 --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>&1 | tail -1) > gpurun_out/bench_b1.log 2>&1
(timeout 300 python tools/profile_layers.py --batch 8 --steps 3 2>&1 | tail -45) > gpurun_out/layers_b8.log 2>&1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_r01
(timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r01 -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --profile-steps 1 2>&1 | tail -2) > $R/gpurun_out/rocprof.log 2>&1
cd $R
python tools/kernel_stats.py gpurun_out/prof_r01 > gpurun_out/kernel_stats_b8.txt 2>&1
find gpurun_out/prof_r01 -name "*.db" -size +20M -delete
tail -3 gpurun_out/pytest_gpu.log; cat gpurun_out/smoke.log; cut -c1-300 gpurun_out/bench_n1.log; cut -c1-200 gpurun_out/bench_b1.log; head -12 gpurun_out/kernel_stats_b8.txt | cut -c1-160
