# kernel-trace statistics of the b = 1 single-graph bench: per-kernel time against the step's wall time
R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp; mkdir -p $R/gpurun_out
cd /tmp; rm -rf $R/gpurun_out/prof_b1
(timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_b1 -o bench -- python $R/bench.py --batch 1 --graph single --steps 40 --warmup 5 --no-cpu-baseline --no-extras --no-d7 --profile-steps 1 2>&1 | tail -1 | cut -c1-200) > $R/gpurun_out/r05_b1_rocprof.log 2>&1
cd $R
python tools/kernel_stats.py gpurun_out/prof_b1 > gpurun_out/r05_kernel_stats_bench_b1_single.txt 2>&1
find gpurun_out/prof_b1 -name "*.db" -size +20M -delete
head -24 gpurun_out/r05_kernel_stats_bench_b1_single.txt | cut -c1-160
