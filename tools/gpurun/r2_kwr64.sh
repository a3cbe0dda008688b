#!/bin/bash
mkdir -p gpurun_out
ODT_CONV_SPLIT3_KWR_N64=1 timeout 600 python -m pytest tests/test_ops.py -q -m gpu -k "split and 256" -x 2>&1 | tail -2
BATCH=8 bash tools/gpurun/ab_layers_env.sh "ODT_CONV_SPLIT3_KWR_N64=0" "ODT_CONV_SPLIT3_KWR_N64=1" > gpurun_out/kwr64_layers_b8.txt 2>&1
grep -E "^\[|layer|group0|conv0" gpurun_out/kwr64_layers_b8.txt
