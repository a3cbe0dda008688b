#!/bin/bash
# is the residual 0.8 % between the round-5 tree and round 6 the producers' counting CODE (compiled in, switched off or sampled)?
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$(pwd)
line() { python -c "
import sys,json; d=json.load(sys.stdin); e=d.get('extra',{}); print('value %.1f  ms %.3f' % (d['value'], d['ms_per_step']))"; }
keep=/tmp/keep_lib.so; cp object_detection_tracking_amd/libodt_hip.so $keep
for rep in 1 2; do
cd $R/wt_r05; echo "[r05 tree] $(timeout 600 python bench.py --steps 20 --warmup 5 --no-d7 --no-cpu-baseline --no-extras --no-nn-matching 2>/dev/null | tail -1 | line)"
cd $R
for v in r06_stats r06_nostats; do
  cp ab/$v.so object_detection_tracking_amd/libodt_hip.so
  echo "[$v] $(timeout 600 python bench.py --steps 20 --warmup 5 --no-d7 --no-cpu-baseline --no-extras --no-nn-matching 2>/dev/null | tail -1 | line)"
  echo "[$v ODT_RANGE_STATS=0] $(ODT_RANGE_STATS=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-d7 --no-cpu-baseline --no-extras --no-nn-matching 2>/dev/null | tail -1 | line)"
done
done 2>&1 | tee gpurun_out/r06i_stats_codegen_ab.txt
cp $keep object_detection_tracking_amd/libodt_hip.so
