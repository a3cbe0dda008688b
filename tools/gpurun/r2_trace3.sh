#!/bin/bash
mkdir -p gpurun_out
timeout 200 python tools/conv_trace.py conv3 conv3nores conv2 conv1 2>&1 | grep -E "^==|conv trace" | cut -c1-260 | grep -v "XCD0 first" > gpurun_out/split3_trace.txt 2>&1
cat gpurun_out/split3_trace.txt
