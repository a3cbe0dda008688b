#!/bin/bash
# round 3: per-phase wall-clock stamps of the fp16x2 kernels on the res4 shapes (b=8), next to the bf16x3 kernels
mkdir -p gpurun_out
for fam in 2 3; do
  echo "--- ODT_CONV_SPLIT_PIPE=$fam" >> gpurun_out/r3_h2_trace.txt
  ODT_CONV_SPLIT_PIPE=$fam timeout 300 python tools/conv_trace.py conv3 conv3nores conv2 conv1 2>&1 | grep -E "^==|conv trace" | cut -c1-260 | grep -v "XCD0 first" >> gpurun_out/r3_h2_trace.txt 2>&1
done
cat gpurun_out/r3_h2_trace.txt
