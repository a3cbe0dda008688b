#!/bin/bash
# round 3: non-temporal hint on the kw-reuse kernel's runs of inputs >= 128 MB (ODT_CONV_NT bit 8) against the default mask 3
mkdir -p gpurun_out
export TMPDIR=/tmp
for nt in 11 3 11 3; do
  ODT_CONV_NT=$nt timeout 300 python tools/profile_layers.py --batch 8 --steps 3 2>&1 | tail -45 > gpurun_out/r3_h2_12_layers_nt$nt.txt
  echo "nt=$nt: $(tail -1 gpurun_out/r3_h2_12_layers_nt$nt.txt)"
  grep -E "posthoc_3x3_p2|rpn/conv0@p2|group2/block0/conv2|lateral_1x1_c2" gpurun_out/r3_h2_12_layers_nt$nt.txt | awk '{printf "   %-42s %7s %6s\n",$1,$6,$7}'
done
