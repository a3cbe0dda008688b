#!/bin/bash
# re-sweep of policy knobs tuned in rounds 2-4 on today's kernels (b = 8 detector only, same box)
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { (env $1 timeout 300 python bench.py --steps 20 --warmup 5 --no-d7 --no-extras --no-cpu-baseline --no-nn-matching 2>/dev/null | tail -1) | python -c "
import sys,json; d=json.load(sys.stdin); print('[$1] b8 FPS %.2f verified %s' % (d['value'], d['verified']))"; }
for c in "X=0" "ODT_CONV_NT=0" "ODT_CONV_NT=1" "ODT_CONV_NT=2" "ODT_CONV_NT=7" "ODT_FUSE_BOTTLENECK=1" "ODT_FUSE_BOTTLENECK=2" "ODT_CONV_SPLIT3_MINTILES=128" "ODT_CONV_SPLIT3_MINTILES=255" "ODT_CONV_SPLIT3_MINTILES=300" "ODT_CONV_H2_N64_BM512=0" "ODT_CONV_H2_N64_BM512=2" "ODT_FUSE_ROT=0" "ODT_CONV_H2_ROT=0" "ODT_TAIL_OVERLAP=0" "X=0"; do run "$c"; done 2>&1 | tee gpurun_out/r06x_knob_sweep.txt
