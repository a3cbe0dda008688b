#!/bin/bash
# round 3, session 1: the new GPU tests, the extended bench line, a same-box per-layer baseline, D7 with / without graph replay
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ingest.py tests/test_drop_in.py tests/test_tf_golden.py tests/test_distributed.py -q -m gpu -x 2>&1 | tail -8 | tee gpurun_out/r3_s1_pytest_a.log
timeout 900 python -m pytest tests/test_efficientnet.py -q -m gpu -x -k "d7_1536 or d0_end_to_end_512" -s 2>&1 | tail -8 | tee gpurun_out/r3_s1_pytest_d7.log
timeout 900 python bench.py --steps 20 --warmup 3 2>gpurun_out/r3_s1_bench_err.log | tail -1 > gpurun_out/r3_s1_bench.json; cut -c1-1500 gpurun_out/r3_s1_bench.json
timeout 300 python tools/profile_layers.py --batch 8 --steps 3 2>&1 | tail -45 > gpurun_out/r3_s1_layers_b8.txt
for g in 1 0; do
  ODT_GRAPH=$g timeout 300 python tools/bench_efficientdet.py --no-cpu-baseline --steps 30 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ODT_GRAPH=$g', d['value'], d['extra'].get('detect_tmot_fps'), d['extra'].get('detect_tmot_pipelined_fps'))" | tee -a gpurun_out/r3_s1_effdet_graph_ab.txt
done
tail -3 gpurun_out/r3_s1_layers_b8.txt
