#!/bin/bash
# PMC pass over the stand-alone probes: is the gap to the matrix-pipe peak clock (power) or stalls?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/probe_pmc
run() { # tag, cmd...
  tag=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE \
    --output-format csv -d $R/gpurun_out/probe_pmc/$tag -o p -- "$@" > $R/gpurun_out/probe_pmc/$tag.log 2>&1
}
run g3_f0 $R/tools/experiments/build/split_gemm3 65280 256 2304 0 1
run g3_f15 $R/tools/experiments/build/split_gemm3 65280 256 2304 15 1
run g3_f2 $R/tools/experiments/build/split_gemm3 65280 256 2304 2 1
run peak $R/tools/experiments/build/mfma_bf16_peak
cd $R
python - <<'PY'
import csv, glob, collections, os
for tag in ("g3_f0", "g3_f15", "g3_f2", "peak"):
  base = "gpurun_out/probe_pmc/" + tag
  cc = glob.glob(base + "/**/p_counter_collection.csv", recursive=True)
  kt = glob.glob(base + "/**/p_kernel_trace.csv", recursive=True)
  if not cc or not kt:
    print(tag, "no output", open(base + ".log").read()[-500:]); continue
  dur = {}
  for r in csv.DictReader(open(kt[0])):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
  per = collections.defaultdict(dict)
  name = {}
  for r in csv.DictReader(open(cc[0])):
    per[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"]); name[r["Dispatch_Id"]] = r["Kernel_Name"]
  seen = collections.Counter()
  for d, c in per.items():
    k = name[d][:60]; seen[k] += 1
    if seen[k] not in (2, 5): continue          # one early, one later dispatch of each kernel
    t = dur.get(d, 0)
    if t <= 0: continue
    clk = c.get("GRBM_GUI_ACTIVE", 0) / 8.0 / t / 1e9            # cycles summed over 8 XCDs
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024.0 / max(1.0, c.get("GRBM_GUI_ACTIVE", 0) / 8.0)
    wc = max(1.0, c.get("SQ_WAVE_CYCLES", 0))
    print("%-6s %-60s %.3f ms  clk %.2f GHz  mfma busy %.3f  wait_any %.3f wait_inst_any %.3f active %.3f wait_inst_lds %.3f valu %.3f lds %.3f" % (
      tag, k, t * 1e3, clk, busy, c.get("SQ_WAIT_ANY", 0) / wc, c.get("SQ_WAIT_INST_ANY", 0) / wc, c.get("SQ_ACTIVE_INST_ANY", 0) / wc,
      c.get("SQ_WAIT_INST_LDS", 0) / wc, c.get("SQ_ACTIVE_INST_VALU", 0) / wc, c.get("SQ_ACTIVE_INST_LDS", 0) / wc))
PY
