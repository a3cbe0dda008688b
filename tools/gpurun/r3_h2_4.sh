#!/bin/bash
# round 3: the 128 x 128 4-wave tile of conv_h2_kernel (two workgroups per CU) -- parity, per-layer A/B over ODT_CONV_H2S_MAXK
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops.py tests/test_e2e.py -q -m gpu -x -k "(split and (2/256 or 2/128)) or fp16x2" 2>&1 | tail -4 | tee gpurun_out/r3_h2_4_pytest.log
for k in 0 512 1024 99999; do
  ODT_CONV_H2S_MAXK=$k timeout 300 python tools/profile_layers.py --batch 8 --steps 3 2>&1 | tail -45 > gpurun_out/r3_h2_4_layers_maxk$k.txt
  echo "maxk=$k: $(tail -1 gpurun_out/r3_h2_4_layers_maxk$k.txt)"
done
