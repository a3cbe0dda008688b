#!/bin/bash
# round 3: f16 MFMA subnormal / conversion probe, then the whole GPU suite + smoke (after the conv_split file split)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 60 tools/experiments/build/mfma_f16_denorm_probe 2>&1 | tee gpurun_out/r03_mfma_f16_denorm_probe.txt
timeout 2700 python -m pytest tests/ -q -m gpu -x 2>&1 | tail -8 | tee gpurun_out/r03_pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -6 | tee gpurun_out/r03_smoke.log
