mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_ops.py -m gpu -q -k conv -p no:cacheprovider 2>&1 | tail -3)
(timeout 300 python tools/conv_trace.py conv3 conv1 conv2 2>&1 | grep -v "^\[conv trace\] prologue\|first wave")
(timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/bench_b8.log 2>&1
(timeout 300 python bench.py --batch 1 --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/bench_b1.log 2>&1
(timeout 300 python tools/profile_layers.py --batch 8 --steps 3 2>&1 | tail -45) > gpurun_out/layers_b8.log 2>&1
python - <<'PY'
import json
for f in ("gpurun_out/bench_b8.log", "gpurun_out/bench_b1.log"):
  try:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, "FPS %.2f ms/step %.3f conv TF %.1f frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["frac"]))
  except Exception as e:
    print(f, "ERR", e, open(f).read()[-500:])
PY
head -12 gpurun_out/layers_b8.log; tail -1 gpurun_out/layers_b8.log
