#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
for c in "WITH_MAIN_ENGINE=0" "WITH_MAIN_ENGINE=1"; do
echo "$c $(env $c timeout 600 python tools/b1_stream_set.py 2>/dev/null | tail -1)"
done | tee gpurun_out/r06_b1_stream_set.txt
