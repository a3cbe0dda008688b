#!/bin/bash
# round 3, session 4: EfficientDet levels merged (D7 parity + same-box A/B), b=16, chunked convs, N>1 bench path on one GPU
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_efficientnet.py -q -m gpu -x -k "d7_1536 or levels_merged or d2_end_to_end" -s 2>&1 | tail -6 | tee gpurun_out/r3_s4_pytest_eff.log
for v in 1 0 1 0; do
  ODT_EFFDET_MERGE_LEVELS=$v timeout 300 python tools/bench_efficientdet.py --no-cpu-baseline --steps 30 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('merge_levels=$v fps %.2f ms %.3f tmot %.2f tmot_pipelined %.2f launches %s' % (d['value'], d['ms_per_step'], d['extra'].get('detect_tmot_fps'), d['extra'].get('detect_tmot_pipelined_fps'), d['extra'].get('handle',{}).get('conv_launches')))" | tee -a gpurun_out/r3_s4_effdet_merge_ab.txt
done
timeout 1200 python -m pytest tests/test_e2e.py -q -m gpu -x -k "b24 or batch_ranges or four_coresident" 2>&1 | tail -5 | tee gpurun_out/r3_s4_pytest_b16.log
