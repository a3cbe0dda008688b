#!/bin/bash
# round-end validation, part 2: PMC passes (HBM traffic, MFMA busy) of the bench command, EfficientDet-D7 bench
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --profile-steps 1"
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-40)
  rm -rf $R/gpurun_out/pmc_$tag
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_$tag -o pmc -- $CMD > $R/gpurun_out/pmc_$tag.log 2>&1
done
cd $R
rm -rf gpurun_out/pmc_SQ_INSTS*
python tools/pmc_summary.py gpurun_out gpurun_out/pmc_summary_split > gpurun_out/pmc_summary.log 2>&1
find gpurun_out -name "*.db" -size +20M -delete; find gpurun_out -name "*.csv" -size +20M -delete
cat gpurun_out/pmc_summary_split.txt | cut -c1-200
(timeout 200 python tools/bench_efficientdet.py --frame 1080x1920 --no-cpu-baseline | tail -1) > gpurun_out/bench_effdet_d7_nosplit.json 2>&1
(ODT_EFFDET_SPLIT=1 timeout 200 python tools/bench_efficientdet.py --frame 1080x1920 --no-cpu-baseline | tail -1) > gpurun_out/bench_effdet_d7_split.json 2>&1
cut -c1-200 gpurun_out/bench_effdet_d7_nosplit.json; cut -c1-200 gpurun_out/bench_effdet_d7_split.json
