mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -q --timeout 400 -p no:cacheprovider 2>&1 | tail -5) > gpurun_out/pytest_gpu.log 2>&1
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3) > gpurun_out/smoke.log 2>&1
(timeout 900 python bench.py --steps 10 --warmup 2 2>&1 | tail -1) > gpurun_out/bench_n1.log 2>&1
(timeout 300 python bench.py --batch 1 --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/bench_b1.log 2>&1
(timeout 300 python tools/profile_layers.py --batch 8 --steps 3 2>&1 | tail -45) > gpurun_out/layers_b8.log 2>&1
(timeout 300 python tools/profile_layers.py --batch 1 --steps 5 2>&1 | tail -45) > gpurun_out/layers_b1.log 2>&1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_r01
(timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r01 -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 1 2>&1 | tail -2) > $R/gpurun_out/rocprof.log 2>&1
cd $R
python tools/kernel_stats.py gpurun_out/prof_r01 > gpurun_out/kernel_stats_b8.txt 2>&1
find gpurun_out/prof_r01 -name "*.db" -size +20M -delete
bash tools/gpurun/pmc.sh > gpurun_out/pmc_run.log 2>&1
python tools/pmc_summary.py gpurun_out gpurun_out/pmc_summary > gpurun_out/pmc_summary.log 2>&1
find gpurun_out -name "*.db" -size +20M -delete; find gpurun_out -name "*.csv" -size +20M -delete
tail -3 gpurun_out/pytest_gpu.log; cat gpurun_out/smoke.log; cat gpurun_out/bench_n1.log; cat gpurun_out/bench_b1.log | cut -c1-200; head -8 gpurun_out/kernel_stats_b8.txt; cat gpurun_out/pmc_summary.txt 2>/dev/null | head -12
# EfficientDet row
(python tools/bench_efficientdet.py --frame 1080x1920 | tail -1) > gpurun_out/bench_effdet_d7.json 2>&1
(python tools/bench_efficientdet.py --model efficientdet-d0 | tail -1) > gpurun_out/bench_effdet_d0.json 2>&1
cd /tmp && export TMPDIR=/tmp; rm -rf $R/gpurun_out/prof_eff
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_eff -o eff -- python $R/tools/bench_efficientdet.py --frame 1080x1920 --steps 10 --no-cpu-baseline > /dev/null 2>&1
cd $R; python tools/kernel_stats.py gpurun_out/prof_eff 18 > gpurun_out/kernel_stats_effdet_d7.txt 2>&1
find gpurun_out/prof_eff -name "*.db" -size +20M -delete
cut -c1-400 gpurun_out/bench_effdet_d7.json; head -12 gpurun_out/kernel_stats_effdet_d7.txt | cut -c1-150
