#!/bin/bash
mkdir -p gpurun_out
BATCH=8 bash tools/gpurun/ab_layers_env.sh "ODT_CONV_DEBUG=0" "ODT_CONV_DEBUG=2" > gpurun_out/ablateA_layers_b8.txt 2>&1
head -14 gpurun_out/ablateA_layers_b8.txt
