#!/bin/bash
# first thing to run next round: the experimental two-stage (BK = 16) loop of the split kernel
# (ODT_CONV_SPLIT_PIPE=2) -- parity on the GPU, then same-box A/B against the default loop.
mkdir -p gpurun_out
ODT_CONV_SPLIT_PIPE=2 timeout 600 python -m pytest tests/test_ops.py tests/test_e2e.py -q -m gpu -k "split or two_sources or b8_1080p or arithmetic" -x 2>&1 | tail -2
for v in "ODT_CONV_SPLIT_PIPE=0" "ODT_CONV_SPLIT_PIPE=2" "ODT_CONV_SPLIT_PIPE=0" "ODT_CONV_SPLIT_PIPE=2"; do
  r8=$(env $v timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f FPS all-conv %.1f TF split %.1f TF' % (d['value'], d['roofline']['all_conv_launches']['achieved'], d['roofline']['achieved']))")
  echo "$v  b8: $r8"
done | tee gpurun_out/split_pipe_ab.txt
BATCH=8 bash tools/gpurun/ab_layers_env.sh "ODT_CONV_SPLIT_PIPE=0" "ODT_CONV_SPLIT_PIPE=2" > gpurun_out/split_pipe_layers_b8.txt 2>&1
head -30 gpurun_out/split_pipe_layers_b8.txt
# where does a short-K tile spend its time?  in-kernel stamps of the split kernel (TRACE instantiation)
# next to the exact-f32 kernel on the res4 shapes
for v in 1 0; do
  echo "--- ODT_CONV_SPLIT=$v"; ODT_CONV_SPLIT=$v timeout 120 python tools/conv_trace.py conv3 conv3nores conv2 conv1 2>&1 | grep -E "^==|conv trace"
done > gpurun_out/split_trace.txt 2>&1
head -40 gpurun_out/split_trace.txt
