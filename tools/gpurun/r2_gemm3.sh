#!/bin/bash
# split_gemm3 structure probe, steady-state clocks (300 back-to-back launches per measurement)
mkdir -p gpurun_out
B=tools/experiments/build/split_gemm3
{
for f in 0 1 2 4 6 8 15; do timeout 60 $B 65280 256 2304 $f 1 300; done
timeout 60 $B 65280 256 2304 0 2 300
for f in 0 6 15; do timeout 60 $B 65280 256 1024 $f 1 600; done
for f in 0 6 15; do timeout 60 $B 65280 1024 256 $f 1 600; done
timeout 60 $B 261120 256 2304 0 1 100
} > gpurun_out/r2_gemm3.log 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r2_gemm3.log"):
  try: d = json.loads(l)
  except Exception: print(l.strip()); continue
  print(d.get("kernel", "split_gemm"), d["M"], d["N"], d["K"], "flags", d["flags"], "ms %.4f" % d["ms"], "TF %.1f" % d["effective_f32_TFLOPs"], "GHz b2b %.2f" % d["ghz_back_to_back"], "| after idle: ms %.4f GHz %.2f" % (d["ms_after_idle"], d["ghz_after_idle"]), "err %.2e" % d["max_err_over_sum_abs"])
PY
