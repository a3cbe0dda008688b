#!/bin/bash
# bf16x3 split-GEMM feasibility probe (tools/experiments/split_gemm.hip); no torch needed.
mkdir -p gpurun_out
B=tools/experiments/build/split_gemm
{
timeout 120 $B 65280 256 2304 6 1 2 1
for v in "6 1" "6 3" "6 5" "6 7" "1 1" "1 5"; do
  timeout 120 $B 65280 256 2304 $v 2 2
done
timeout 120 $B 65280 256 1024 6 1 2 2
timeout 120 $B 1044480 256 2304 6 1 2 2
timeout 120 $B 65280 1024 256 6 1 2 2
} > gpurun_out/split_gemm.log 2>&1
cat gpurun_out/split_gemm.log
