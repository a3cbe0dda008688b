#!/bin/bash
# HBM traffic counters of the EfficientDet-D7 forward: two separate --pmc passes (kernel-trace only), then the summary
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmceff_$c
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmceff_$c -o pmc -- python $R/tools/bench_efficientdet.py --no-cpu-baseline --steps 2 --warmup 1 > $R/gpurun_out/pmceff_$c.log 2>&1
done
cd $R
python tools/pmc_summary_effdet.py gpurun_out gpurun_out/r02_pmc_summary_effdet_d7 2>&1 | tail -45
find gpurun_out -name "*.csv" -size +20M -delete; find gpurun_out -name "*.db" -size +20M -delete
