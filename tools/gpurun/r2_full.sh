#!/bin/bash
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/ -q -m gpu 2>&1 | tail -6 | tee gpurun_out/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 | tee gpurun_out/smoke.log
timeout 120 python tools/tracker_bench.py native 2>&1 | tail -1 | cut -c1-500
timeout 420 python bench.py --steps 20 --warmup 3 2>gpurun_out/bench_err.log | tail -1 > gpurun_out/bench_b8.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_b8.json"))
print("FPS %.1f  split %.1f TF frac %.3f" % (d["value"], d["roofline"]["achieved"], d["roofline"]["frac"]))
print(json.dumps(d["extra"])[:1200]); print(json.dumps(d["handle"])); print(json.dumps(d.get("cpu_baseline"))[:300])
PY
