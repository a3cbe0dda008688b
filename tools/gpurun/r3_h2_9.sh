#!/bin/bash
# round 3: non-temporal hint on the residual chunks (1) / the single-use activation loads of the 1x1 layers (2) / both (3): per-layer A/B;
# plus the fp16x2 e2e parity subset on the code with the earlier first weight DMA
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops.py tests/test_e2e.py -q -m gpu -x -k "(split and (2/256 or 2/128)) or fp16x2 or multi_r101_b2" 2>&1 | tail -4 | tee gpurun_out/r3_h2_9_pytest.log
for nt in 0 1 2 3 0; do
  ODT_CONV_NT=$nt timeout 300 python tools/profile_layers.py --batch 8 --steps 3 2>&1 | tail -45 > gpurun_out/r3_h2_9_layers_nt$nt.txt
  echo "nt=$nt: $(tail -1 gpurun_out/r3_h2_9_layers_nt$nt.txt)"
  grep -E "group2/block0/conv3|group2/block1/conv1|group0/block0/conv3|group1/block0/conv3" gpurun_out/r3_h2_9_layers_nt$nt.txt | awk '{printf "   %-42s %7s %6s\n",$1,$6,$7}'
done
