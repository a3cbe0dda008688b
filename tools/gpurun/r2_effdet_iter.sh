#!/bin/bash
# EfficientDet iteration check: GPU parity tests of the EfficientNet / EfficientDet path, D7 bench, per-kernel profile
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_efficientnet.py -x -q -m gpu 2>&1 | tail -3
(timeout 300 python tools/bench_efficientdet.py --no-cpu-baseline --steps 20 2>&1 | tail -1) > gpurun_out/effdet_d7.json
python -c "
import json; d=json.load(open('gpurun_out/effdet_d7.json')); print('D7 FPS %.1f ms %.2f frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))"
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_effdet
(timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_effdet -o eff -- python $R/tools/bench_efficientdet.py --no-cpu-baseline --steps 10 --warmup 2 2>&1 | tail -1) > $R/gpurun_out/effdet_rocprof.log 2>&1
cd $R
python tools/kernel_stats.py gpurun_out/prof_effdet 17 > gpurun_out/kernel_stats_effdet_d7.txt 2>&1
find gpurun_out/prof_effdet -name "*.db" -size +20M -delete
head -36 gpurun_out/kernel_stats_effdet_d7.txt | cut -c1-200
