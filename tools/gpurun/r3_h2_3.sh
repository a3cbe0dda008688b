#!/bin/bash
# round 3: K-slice rotation of the fp16x2 1x1 layers (ODT_CONV_H2_ROT) x activations by LDS-DMA (ODT_CONV_H2_ADMA): parity, per-layer A/B
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops.py -q -m gpu -x -k "(split and 2/256) or fp16x2" 2>&1 | tail -4 | tee gpurun_out/r3_h2_3_pytest.log
for cfg in "1 1" "1 0" "0 1" "0 0"; do
  set -- $cfg
  ODT_CONV_H2_ROT=$1 ODT_CONV_H2_ADMA=$2 timeout 300 python tools/profile_layers.py --batch 8 --steps 3 2>&1 | tail -45 > gpurun_out/r3_h2_3_layers_rot$1_adma$2.txt
  echo "rot=$1 adma=$2: $(tail -1 gpurun_out/r3_h2_3_layers_rot$1_adma$2.txt)"
done
