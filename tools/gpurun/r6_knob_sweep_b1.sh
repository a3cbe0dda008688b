#!/bin/bash
# re-sweep of the b = 1 policy knobs on today's kernels (BASELINE config #2, Mask_RCNN_FPN, same box)
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { (env $1 timeout 300 python bench.py --batch 1 --graph single --steps 40 --warmup 5 --no-d7 --no-extras --no-cpu-baseline 2>/dev/null | tail -1) | python -c "
import sys,json; d=json.load(sys.stdin); print('[$1] b1 FPS %.2f verified %s' % (d['value'], d['verified']))"; }
for c in "X=0" "ODT_CONV_H2_BM64=0" "ODT_CONV_H2_BM64=1" "ODT_CONV_H2_BM64=2" "ODT_CONV_SPLIT3_FILLDIV=4" "ODT_CONV_SPLIT3_FILLDIV=8" "ODT_SPLIT_REDUCE_BLOCKS=256" "ODT_SPLIT_REDUCE_BLOCKS=1024" "ODT_CONV_H2K_SPLITK=0" "ODT_CONV_SPLIT3_MINTILES=128" "ODT_CONV_SPLIT3_MINTILES=300" "ODT_CONV_H2_BK64=0" "ODT_FUSE_BOTTLENECK=0" "ODT_CONV_SPLIT3_SPLITK=4" "ODT_CONV_SPLIT3_SPLITK=16" "ODT_CONV_H2K_FEWROWS=0" "ODT_ROI_AMAX=0" "X=0"; do run "$c"; done 2>&1 | tee gpurun_out/r06y_knob_sweep_b1.txt
