#!/bin/bash
# staging copy by four threads: ingest tests on hip, the bench's host-boundary legs
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ingest.py tests/test_drop_in.py -q -m gpu -x 2>&1 | tail -3
for rep in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-d7 --no-cpu-baseline --no-live-traffic 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.load(sys.stdin); e=d['extra']; print('value %.1f' % d['value'], 'det only', round(e.get('detector_only_fps_without_nn_matching_in_the_step'),1), 'pcie blocking', round(e.get('pcie_inclusive_fps'),1), 'pipelined', round(e.get('pcie_inclusive_pipelined_fps'),1), 'incl fill', round(e.get('pcie_inclusive_pipelined_incl_fill_fps'),1), 'pinned', round(e.get('pcie_inclusive_pipelined_pinned_source_fps'),1), 'detect_track', round(e.get('detect_track_fps'),1), round(e.get('detect_track',{}).get('host_tracking_ms_per_frame'),2), 'arrays', round(e.get('detect_track_arrays_fps'),1), 'two_streams', round(e.get('two_streams_per_gpu_fps'),1))"
done 2>&1 | tee gpurun_out/r06s_staging_threads.txt
