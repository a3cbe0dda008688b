#!/bin/bash
# Round 6: EfficientDet graph's tail (top-k / sort / NMS / ROIAlign) under the next frame's backbone: tests on hip, D7 A/B
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_efficientnet.py -q -m gpu -x 2>&1 | tail -3
for v in 0 1 0 1; do
ODT_EFFDET_TAIL_OVERLAP=$v timeout 600 python tools/bench_efficientdet.py --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.load(sys.stdin); print('EFFDET_TAIL_OVERLAP=$v D7 FPS %.2f verified %s' % (d['value'], d.get('verified')), {k: round(v,1) for k,v in d.get('extra',{}).items() if isinstance(v,float)})"
done 2>&1 | tee gpurun_out/r06u_d7_tail_overlap_ab.txt
