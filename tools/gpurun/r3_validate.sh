#!/bin/bash
# round 3: the whole GPU suite + smoke on the final code
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2700 python -m pytest tests/ -q -m gpu -x 2>&1 | tail -8 | tee gpurun_out/r03_pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -6 | tee gpurun_out/r03_smoke.log
