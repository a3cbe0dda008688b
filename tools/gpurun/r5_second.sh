#!/bin/bash
# Round 5, second box: the rest of the GPU suite (the first box stopped at a test bug), ROIAlign A/B with the 16-byte taps.
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/ -q -m gpu -x --durations=8 --deselect tests/test_e2e.py::test_forward_multi_r101_b8_1080p --deselect tests/test_e2e.py::test_mixed_exposure_batch_b8_1080p_and_batch_independence -k "not test_drop_in and not test_abi and not distributed" 2>&1 | tail -22 | tee gpurun_out/r05b_pytest_gpu.log
for v in roi_old r5_new; do
  cp ab/$v.so object_detection_tracking_amd/libodt_hip.so
  cd /tmp; rm -rf $R/gpurun_out/prof_roi_$v
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_roi_$v -o b1 -- python $R/bench.py --batch 1 --graph single --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-d7 --profile-steps 1 > $R/gpurun_out/roi_$v.log 2>&1
  cd $R
  python tools/kernel_stats.py gpurun_out/prof_roi_$v | grep -i "roi_\|# " | cut -c1-160 | sed "s/^/[$v] /"
  python - <<PY
import glob, sqlite3
db = sorted(glob.glob("gpurun_out/prof_roi_$v/**/*.db", recursive=True))[-1]
con = sqlite3.connect(db)
rows = con.execute("select name, start, end from kernels where name like '%roi_%' order by start").fetchall()
# per forward: box-head call (300 RoIs, NHWC only), feature call (100 RoIs, NCHW + pooled) [+ mean pass]
import collections
d = collections.defaultdict(list)
seq = [(n.split('(')[0].split('::')[-1][:24], (e - s) / 1e3) for n, s, e in rows]
per = 3 if any('pool_mean' in n for n, _ in seq) else 2
for i, (n, us) in enumerate(seq): d[(i % per, n)].append(us)
for k in sorted(d): print("[$v] call %d %-24s n=%d median %.1f us" % (k[0], k[1], len(d[k]), sorted(d[k])[len(d[k]) // 2]))
PY
  find gpurun_out/prof_roi_$v -name "*.db" -size +20M -delete
done | tee gpurun_out/r05b_roi_align_ab.txt
cp ab/r5_new.so object_detection_tracking_amd/libodt_hip.so
