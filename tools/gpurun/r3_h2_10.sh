#!/bin/bash
# round 3: non-temporal hints, second pass: residual + single-use A (3) confirmed against 0, + large-output stores (7), stores alone (4)
mkdir -p gpurun_out
export TMPDIR=/tmp
for nt in 3 0 7 4 3 0 7; do
  ODT_CONV_NT=$nt timeout 300 python tools/profile_layers.py --batch 8 --steps 3 2>&1 | tail -45 > gpurun_out/r3_h2_10_layers_nt$nt.txt
  echo "nt=$nt: $(tail -1 gpurun_out/r3_h2_10_layers_nt$nt.txt)"
done
for nt in 3 7; do
  echo "== nt=$nt"; grep -E "group2/block0/conv3|group2/block1/conv1|group0/block0/conv3|group1/block0/conv3|group2/block0/conv2|posthoc_3x3_p2|lateral_1x1_c2" gpurun_out/r3_h2_10_layers_nt$nt.txt | awk '{printf "   %-42s %7s %6s\n",$1,$6,$7}'
done
