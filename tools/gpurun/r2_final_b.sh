#!/bin/bash
# after the stream-priority change: the GPU tests that touch streams / ingest / tracker, smoke, and the full bench line
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ingest.py tests/test_tracker_core.py tests/test_oracle_golden.py tests/test_drop_in.py tests/test_distributed.py -q -m gpu 2>&1 | tail -3 | tee gpurun_out/pytest_gpu_streams.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke | tee gpurun_out/smoke.log
timeout 420 python bench.py --steps 20 --warmup 3 2>gpurun_out/bench_err.log | tail -1 > gpurun_out/bench_b8.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_b8.json"))
print("b8 FPS %.1f  split %.1f TF frac %.3f" % (d["value"], d["roofline"]["achieved"], d["roofline"]["frac"]))
print(json.dumps({k: v for k, v in d["extra"].items() if not isinstance(v, dict)})[:900])
print(d["cpu_baseline"])
PY
