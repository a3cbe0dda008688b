#!/bin/bash
mkdir -p gpurun_out
BATCH=8 bash tools/gpurun/ab_layers_env.sh "ODT_CONV_SPLIT3_SHORTK=0" "ODT_CONV_SPLIT3_SHORTK=256" "ODT_CONV_SPLIT3_SHORTK=1024" > gpurun_out/shortk_layers_b8.txt 2>&1
head -34 gpurun_out/shortk_layers_b8.txt
