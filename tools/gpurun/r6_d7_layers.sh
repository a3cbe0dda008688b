#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/profile_layers.py --effdet efficientdet-d7 --steps 5 --top 70 2>&1 | tail -90 > gpurun_out/r06_d7_layers.txt
cut -c1-170 gpurun_out/r06_d7_layers.txt
