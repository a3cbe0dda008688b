#!/bin/bash
# Round 6, first box: the half-chip two-chain experiment (tools/experiments/half_chip_chains.py), b = 4 plans with the tile
# thresholds halved (HALF_POLICY=1) and as the library plans them
mkdir -p gpurun_out
export TMPDIR=/tmp
HALF_POLICY=1 timeout 600 python tools/experiments/half_chip_chains.py 2>&1 | tail -22 | tee gpurun_out/r06a_half_chip_chains_half_policy.txt
