#!/bin/bash
mkdir -p gpurun_out
timeout 200 python tools/conv_trace.py conv3 conv3nores 2>&1 | grep -E "^==|conv trace" | cut -c1-260 | grep -v "XCD0 first" | head -8
timeout 600 python -m pytest tests/test_ops.py -q -m gpu -k "split" -x 2>&1 | tail -2
BATCH=8 bash tools/gpurun/ab_layers_env.sh "ODT_CONV_SPLIT_PIPE=3" > gpurun_out/epi_layers_b8.txt 2>&1
head -20 gpurun_out/epi_layers_b8.txt
