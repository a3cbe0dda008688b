#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$(pwd)
line() { python -c "
import sys,json; d=json.load(sys.stdin); e=d['extra']; print('value %.1f' % d['value'], 'det only', round(e.get('detector_only_fps_without_nn_matching_in_the_step'),1), 'detect_track', round(e.get('detect_track_fps'),1), round(e.get('detect_track',{}).get('host_tracking_ms_per_frame'),2), 'arrays', round(e.get('detect_track_arrays_fps'),1), round(e.get('detect_track_arrays_host_ms_per_frame'),2), 'pipelined', round(e.get('pcie_inclusive_pipelined_fps'),1))"; }
cd $R/wt_r05; echo "[r05 tree] $(timeout 600 python bench.py --steps 10 --warmup 2 --no-d7 --no-cpu-baseline 2>/dev/null | tail -1 | line)"
cd $R
for v in "X=1" "ODT_RANGE_HOST=0" "ODT_STAGE_FRAMES=0" "ODT_NO_WATCH=1" "ODT_RANGE_HOST=0 ODT_STAGE_FRAMES=0 ODT_NO_WATCH=1 ODT_RANGE_STATS=0"; do
echo "[r06 $v] $(env $v timeout 600 python bench.py --steps 10 --warmup 2 --no-d7 --no-cpu-baseline --no-live-traffic 2>/dev/null | tail -1 | line)"
done
