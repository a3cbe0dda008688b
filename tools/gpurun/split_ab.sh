#!/bin/bash
# bf16x3 split path: parity tests on the GPU, then A/B against the exact-f32 kernels in ONE box.
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ops.py -q -m gpu -k "split or model_shapes" -x > gpurun_out/split_tests.log 2>&1
tail -3 gpurun_out/split_tests.log
ODT_CONV_SPLIT=1 timeout 600 python -m pytest tests/test_ops.py tests/test_e2e.py -q -m gpu -k "model_shapes or b8_1080p or single_r101_1080p or single_small" -x > gpurun_out/split_tests_e2e.log 2>&1
tail -3 gpurun_out/split_tests_e2e.log
for rep in 1 2; do for v in 0 1; do
  r8=$(ODT_CONV_SPLIT=$v timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f FPS conv %.1f TF' % (d['value'], d['roofline']['achieved']))")
  echo "split=$v rep$rep  b8: $r8"
done; done | tee gpurun_out/split_ab.txt
r1=$(ODT_CONV_SPLIT=1 timeout 300 python bench.py --batch 1 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f FPS' % d['value'])")
echo "split=1 b1: $r1" | tee -a gpurun_out/split_ab.txt
BATCH=8 bash tools/gpurun/ab_layers_env.sh "ODT_CONV_SPLIT=0" "ODT_CONV_SPLIT=1" > gpurun_out/split_layers_b8.txt 2>&1
head -45 gpurun_out/split_layers_b8.txt
