#!/bin/bash
# the detect + track leg INSIDE bench.py (T ~ 66 confirmed tracks per class) with the round-5 cosine kernel vs the round-6 one
mkdir -p gpurun_out
export TMPDIR=/tmp
keep=/tmp/keep_lib.so; cp object_detection_tracking_amd/libodt_hip.so $keep
for v in r06_old_cosine r06_cur r06_old_cosine r06_cur; do
cp ab/$v.so object_detection_tracking_amd/libodt_hip.so
ODT_TRACKER_TIMING=1 timeout 600 python bench.py --steps 10 --warmup 2 --no-d7 --no-cpu-baseline --no-live-traffic 2>gpurun_out/r06g_track_timing_bench.err | tail -1 | python -c "
import sys,json; d=json.load(sys.stdin); e=d['extra']; print('[$v] value %.1f' % d['value'], 'det only', round(e.get('detector_only_fps_without_nn_matching_in_the_step'),1), 'detect_track', round(e.get('detect_track_fps'),1), round(e.get('detect_track',{}).get('host_tracking_ms_per_frame'),2), 'arrays', round(e.get('detect_track_arrays_fps'),1), round(e.get('detect_track_arrays_host_ms_per_frame'),2))"
grep "cosine" gpurun_out/r06g_track_timing_bench.err | tail -2
done 2>&1 | tee gpurun_out/r06g_cosine_in_bench_ab.txt
cp $keep object_detection_tracking_amd/libodt_hip.so
