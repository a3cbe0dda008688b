#!/bin/bash
# round 2 evidence: b=1 bench, per-layer table, rocprofv3 kernel-trace stats, separate --pmc passes of the bench command
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
(timeout 300 python bench.py --batch 1 --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1) > gpurun_out/bench_b1.json
cut -c1-260 gpurun_out/bench_b1.json
(timeout 300 python tools/profile_layers.py --batch 8 --steps 3 2>&1 | tail -45) > gpurun_out/layers_b8.txt 2>&1
(timeout 300 python tools/profile_layers.py --batch 1 --steps 3 2>&1 | tail -45) > gpurun_out/layers_b1.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_r02
(timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r02 -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --profile-steps 1 2>&1 | tail -2) > $R/gpurun_out/rocprof.log 2>&1
cd $R
python tools/kernel_stats.py gpurun_out/prof_r02 > gpurun_out/kernel_stats_b8.txt 2>&1
find gpurun_out/prof_r02 -name "*.db" -size +20M -delete
head -14 gpurun_out/kernel_stats_b8.txt | cut -c1-170
cd /tmp
CMD="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --profile-steps 1"
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-40)
  rm -rf $R/gpurun_out/pmc_$tag
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_$tag -o pmc -- $CMD > $R/gpurun_out/pmc_$tag.log 2>&1
done
cd $R
rm -rf gpurun_out/pmc_SQ_INSTS*
python tools/pmc_summary.py gpurun_out gpurun_out/r02_pmc_summary_split > gpurun_out/pmc_summary.log 2>&1
find gpurun_out -name "*.db" -size +20M -delete; find gpurun_out -name "*.csv" -size +20M -delete
cat gpurun_out/r02_pmc_summary_split.txt | cut -c1-200
