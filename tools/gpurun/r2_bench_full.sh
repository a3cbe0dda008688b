#!/bin/bash
mkdir -p gpurun_out
timeout 420 python bench.py --steps 20 --warmup 3 2>gpurun_out/bench_err.log | tail -1 > gpurun_out/bench_b8.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_b8.json"))
print("FPS %.1f  split %.1f TF frac %.3f" % (d["value"], d["roofline"]["achieved"], d["roofline"]["frac"]))
print(json.dumps(d["extra"], indent=1)[:2500])
print(json.dumps(d.get("cpu_baseline"), indent=1))
PY
tail -5 gpurun_out/bench_err.log
