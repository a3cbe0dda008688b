#!/bin/bash
# bf16x3 split-GEMM feasibility probe (tools/experiments/split_gemm.hip); no torch needed.
mkdir -p gpurun_out
B=tools/experiments/build/split_gemm
{
for rep in 1 2; do
for v in "6 1" "6 9" "6 5" "6 13"; do
  timeout 120 $B 65280 256 2304 $v 2 2
done
for v in "6 1" "6 9"; do
  timeout 120 $B 1044480 256 2304 $v 2 2
  timeout 120 $B 65280 256 1024 $v 2 2
done
done
} > gpurun_out/split_gemm.log 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/split_gemm.log"):
  try: d = json.loads(l)
  except Exception: print(l.strip()); continue
  print(d["M"], d["N"], d["K"], "flags", d["flags"], "ms %.4f" % d["ms"], "TF %.1f" % d["effective_f32_TFLOPs"], "err %.2e" % d["max_err_over_sum_abs"])
PY
