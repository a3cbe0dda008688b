#!/bin/bash
# Round 5, first box: the whole GPU suite on the round's first batch of changes, the bench line (nn_matching inside the
# step, guarded default handle), and the ROIAlign kernel A/B (old two-launch form vs the fused one) by kernel trace.
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/ -q -m gpu -x --durations=15 2>&1 | tail -30 | tee gpurun_out/r05a_pytest_gpu.log
(timeout 600 python bench.py --steps 20 --warmup 5 --no-d7 2>gpurun_out/r05a_bench_err.log | tail -1) > gpurun_out/r05a_bench_n1.json
python -c "
import json; d=json.load(open('gpurun_out/r05a_bench_n1.json')); r=d['roofline']
print('b8 FPS %.2f  frac %.4f  frac_of_sustained %.4f verified %s' % (d['value'], r['frac'], r.get('frac_of_sustained', 0), d['verified']))
print(d['handle'].get('range_guard'), d['handle'].get('conv_split_family_auto'))
e=d['extra']; print({k: (round(v, 2) if isinstance(v, float) else v) for k, v in e.items() if not isinstance(v, dict)})"
(timeout 300 python bench.py --steps 20 --warmup 5 --no-d7 --no-extras --no-cpu-baseline --no-nn-matching 2>/dev/null | tail -1) > gpurun_out/r05a_bench_n1_no_nn.json
python -c "
import json; d=json.load(open('gpurun_out/r05a_bench_n1_no_nn.json')); print('without nn_matching in the step: %.2f FPS' % d['value'])"
# ROIAlign A/B: kernel trace of the b = 1 single-graph bench with either library
for v in roi_old r5_new; do
  cp ab/$v.so object_detection_tracking_amd/libodt_hip.so
  cd /tmp; rm -rf $R/gpurun_out/prof_roi_$v
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_roi_$v -o b1 -- python $R/bench.py --batch 1 --graph single --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-d7 --profile-steps 1 > $R/gpurun_out/roi_$v.log 2>&1
  cd $R
  python tools/kernel_stats.py gpurun_out/prof_roi_$v | grep -i "roi_\|# " | cut -c1-160 | sed "s/^/[$v] /"
  find gpurun_out/prof_roi_$v -name "*.db" -size +20M -delete
done | tee gpurun_out/r05a_roi_align_ab.txt
cp ab/r5_new.so object_detection_tracking_amd/libodt_hip.so
