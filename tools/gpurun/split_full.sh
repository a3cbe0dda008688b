#!/bin/bash
# full GPU suite with the split path at its default (on), then bench + per-layer table, split off/on
mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu -x > gpurun_out/pytest_gpu.log 2>&1
tail -3 gpurun_out/pytest_gpu.log
for v in 0 1; do
  ODT_CONV_SPLIT=$v timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_split$v.log 2>&1
  tail -1 gpurun_out/bench_split$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('split=$v b8 %.2f FPS' % d['value'], json.dumps(d['roofline'])[:600])"
done
timeout 300 python bench.py --batch 1 --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench_split1_b1.log 2>&1
tail -1 gpurun_out/bench_split1_b1.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('split=1 b1 %.2f FPS' % d['value'])"
BATCH=8 bash tools/gpurun/ab_layers_env.sh "ODT_CONV_SPLIT=0" "ODT_CONV_SPLIT=1" > gpurun_out/split_layers_b8.txt 2>&1
head -24 gpurun_out/split_layers_b8.txt
