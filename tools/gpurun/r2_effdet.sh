#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_efficientnet.py -q -m gpu 2>&1 | tail -3
(ODT_EFFDET_SPLIT=0 timeout 200 python tools/bench_efficientdet.py --frame 1080x1920 --no-cpu-baseline 2>/dev/null | tail -1) > gpurun_out/bench_effdet_d7_nosplit.json
(timeout 200 python tools/bench_efficientdet.py --frame 1080x1920 --no-cpu-baseline 2>/dev/null | tail -1) > gpurun_out/bench_effdet_d7_split.json
python - <<'PY'
import json
for f in ("nosplit", "split"):
  try:
    d = json.load(open("gpurun_out/bench_effdet_d7_%s.json" % f)); print(f, "%.1f FPS %.2f ms frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"]))
  except Exception as e: print(f, "failed", e, open("gpurun_out/bench_effdet_d7_%s.json" % f).read()[-300:])
PY
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp; rm -rf $R/gpurun_out/prof_eff
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_eff -o eff -- python $R/tools/bench_efficientdet.py --frame 1080x1920 --no-cpu-baseline --steps 5 --warmup 2 > /dev/null 2>&1
cd $R; python tools/kernel_stats.py gpurun_out/prof_eff | cut -c1-160 | head -24 | tee gpurun_out/kernel_stats_effdet_d7.txt
find gpurun_out/prof_eff -name "*.db" -size +20M -delete
