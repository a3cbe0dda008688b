#!/bin/bash
# what part of a cosine call costs the detector its 1.3 %?  (TEMP diagnostic: ODT_TRACKER_TIMING=2 no copies, 3 no kernel)
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { (env $1 timeout 300 python bench.py --steps 30 --warmup 5 --no-d7 --no-extras --no-cpu-baseline $2 2>/dev/null | tail -1) | python -c "
import sys,json; d=json.load(sys.stdin); print('[$1 $2] b8 FPS %.2f' % d['value'])"; }
for rep in 1 2; do
run "X=0" "--no-nn-matching"; run "X=0" ""; run "ODT_TRACKER_TIMING=2" ""; run "ODT_TRACKER_TIMING=3" ""
done 2>&1 | tee gpurun_out/r06z_cosine_cost_parts.txt
