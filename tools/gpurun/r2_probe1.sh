#!/bin/bash
# round 2, call 1: hardware questions for the split-kernel redesign + the A/B round 1 left behind
mkdir -p gpurun_out
{
echo "== bf16 MFMA pipe on the split kernel's instruction mix"; timeout 120 tools/experiments/build/mfma_bf16_peak
echo "== buffer_load ... lds with out-of-range lanes"; timeout 60 tools/experiments/build/glds_oob_probe
} > gpurun_out/r2_probe1_micro.txt 2>&1
cat gpurun_out/r2_probe1_micro.txt
ODT_CONV_SPLIT_PIPE=2 timeout 300 python tools/gpurun/pipe2_check.py 2>&1 | tail -3
for v in "ODT_CONV_SPLIT_PIPE=0" "ODT_CONV_SPLIT_PIPE=2"; do
  r8=$(env $v timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f FPS all-conv %.1f TF split %.1f TF' % (d['value'], d['roofline']['all_conv_launches']['achieved'], d['roofline']['achieved']))")
  echo "$v  b8: $r8"
done | tee gpurun_out/split_pipe_ab.txt
BATCH=8 bash tools/gpurun/ab_layers_env.sh "ODT_CONV_SPLIT_PIPE=0" "ODT_CONV_SPLIT_PIPE=2" > gpurun_out/split_pipe_layers_b8.txt 2>&1
head -30 gpurun_out/split_pipe_layers_b8.txt
for v in 1 0; do
  echo "--- ODT_CONV_SPLIT=$v"; ODT_CONV_SPLIT=$v timeout 120 python tools/conv_trace.py conv3 conv3nores conv2 conv1 2>&1 | grep -E "^==|conv trace"
done > gpurun_out/split_trace.txt 2>&1
head -40 gpurun_out/split_trace.txt
