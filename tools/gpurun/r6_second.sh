#!/bin/bash
# Round 6, second box: nn_matching as a small GEMM (tracker.hip) + the knob reader -- tests on hip, the step with / without the
# per-frame nn_matching calls, the cosine kernel's duration next to the detector (kernel trace).
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops.py tests/test_oracle_golden.py tests/test_tracker_core.py tests/test_drop_in.py tests/test_abi.py -q -m gpu -x -k "cosine or tracker or abi or drop or describe" 2>&1 | tail -5 | tee gpurun_out/r06b_pytest_gpu_subset.log
for v in "" "--no-nn-matching"; do
  (timeout 300 python bench.py --steps 20 --warmup 5 --no-d7 --no-extras --no-cpu-baseline $v 2>gpurun_out/r06b_bench_err.log | tail -1) > gpurun_out/r06b_bench_n1$v.json
  python -c "
import json; d=json.load(open('gpurun_out/r06b_bench_n1$v.json')); r=d['roofline']
print('$v b8 FPS %.2f  frac %.4f  frac_of_sustained %.4f verified %s env %s' % (d['value'], r['frac'], r.get('frac_of_sustained', 0), d['verified'], d['handle'].get('env_overrides')))"
done
cd /tmp; rm -rf $R/gpurun_out/prof_r06b
(timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r06b -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-d7 --profile-steps 1 2>&1 | tail -2) > $R/gpurun_out/r06b_rocprof.log 2>&1
cd $R
python tools/kernel_stats.py gpurun_out/prof_r06b > gpurun_out/r06b_kernel_stats_bench_b8_1080p.txt 2>&1
find gpurun_out/prof_r06b -name "*.db" -size +20M -delete
grep -i "cosine\|^#" gpurun_out/r06b_kernel_stats_bench_b8_1080p.txt | cut -c1-170
