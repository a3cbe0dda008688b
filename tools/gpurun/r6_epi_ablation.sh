#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
BATCH=8 bash tools/gpurun/ab_layers_env.sh "X=0" "ODT_CONV_DEBUG=777" 2>&1 | cut -c1-150 | tee gpurun_out/r06v_epilogue_ablation.txt
