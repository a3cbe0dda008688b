#!/bin/bash
# Round 5: same-box A/B of library builds (ab/<name>.so from tools/ab_build_all.sh): b = 8 headline and b = 1 single graph.
#   VARIANTS="amax1 amax16" bash tools/gpurun/r5_lib_ab.sh
mkdir -p gpurun_out
keep=/tmp/keep_lib.so; cp object_detection_tracking_amd/libodt_hip.so $keep
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%.2f FPS  %.3f ms  frac %.3f verified %s' % (d['value'], d['ms_per_step'], r['frac'], d.get('verified')))"; }
for rep in 1 2; do for v in $VARIANTS; do
  cp ab/$v.so object_detection_tracking_amd/libodt_hip.so
  r8=$(timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-d7 2>/dev/null | tail -1 | line)
  r1=$(timeout 300 python bench.py --batch 1 --graph single --steps 40 --warmup 5 --no-cpu-baseline --no-extras --no-d7 2>/dev/null | tail -1 | line)
  echo "[$v] rep$rep  b8: $r8 | b1 single: $r1"
done; done | tee gpurun_out/${OUT:-r05_lib_ab}.txt
cp $keep object_detection_tracking_amd/libodt_hip.so
