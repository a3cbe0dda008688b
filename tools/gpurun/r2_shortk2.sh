#!/bin/bash
# short-K layers on the two-stage 128x256 kernel (2 workgroups / CU): persistent workgroups with a dynamic tile counter and per-workgroup start offsets
mkdir -p gpurun_out
for cfg in "0 0 0" "256 0 0" "256 0 512" "256 2000 512" "256 4000 512" "256 6000 512" "256 4000 768"; do
  set -- $cfg
  echo "== ODT_CONV_SPLIT2_SHORTK=$1 ODT_CONV_SPLIT2_STAGGER=$2 ODT_CONV_SPLIT2_PERSIST=$3"
  ODT_CONV_SPLIT2_SHORTK=$1 ODT_CONV_SPLIT2_STAGGER=$2 ODT_CONV_SPLIT2_PERSIST=$3 timeout 300 python tools/profile_layers.py --batch 8 --steps 3 2>&1 | grep -E "conv3|lateral_1x1_c2|conv total"
done 2>&1 | tee gpurun_out/shortk2_ab.txt
