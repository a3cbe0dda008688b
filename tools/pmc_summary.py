#!/usr/bin/env python
"""Summarise rocprofv3 --pmc passes (separate passes: FETCH_SIZE | WRITE_SIZE | SQ MFMA counters)
collected with tools/gpurun/pmc.sh into profiles/<round>_pmc_summary.{txt,json}."""
import collections, csv, json, os, sys
src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
out = sys.argv[2] if len(sys.argv) > 2 else "profiles/r02_pmc_summary"
def load(tag):
  d = [x for x in os.listdir(src) if x.startswith("pmc_" + tag) and os.path.isdir(os.path.join(src, x))][0]
  return list(csv.DictReader(open(os.path.join(src, d, "pmc_counter_collection.csv"))))
def agg(rows, pred):
  tot = collections.defaultdict(float); disp = set()
  for r in rows:
    if pred(r["Kernel_Name"]):
      tot[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
  return tot, len(disp)
is_conv = lambda k: "conv_igemm_kernel" in k or "conv_split" in k or "conv_h2" in k or "conv_stem_kernel" in k
is_pre = lambda k: "preprocess_kernel" in k
f, nconv = agg(load("FETCH_SIZE"), is_conv)
w, _ = agg(load("WRITE_SIZE"), is_conv)
m, _ = agg(load("SQ_VALU_MFMA"), is_conv)
try:
  i, _ = agg(load("SQ_INSTS"), is_conv)
except IndexError:            # optional fourth pass
  i = collections.defaultdict(float)
is_split = lambda k: ("conv_split" in k or "conv_h2" in k or "conv_stem" in k) and "kernel" in k and "split_weights" not in k
fs, nsplit = agg(load("FETCH_SIZE"), is_split)
ws, _ = agg(load("WRITE_SIZE"), is_split)
ms, _ = agg(load("SQ_VALU_MFMA"), is_split)
_, nfwd = agg(load("FETCH_SIZE"), is_pre)
res = {
  "command": "rocprofv3 --kernel-trace --pmc <set> --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 1",
  "forwards_profiled": nfwd, "conv_launches": nconv,
  "conv_fetch_GB_per_forward_raw": f["FETCH_SIZE"] * 1024 / nfwd / 1e9,
  "conv_write_GB_per_forward_raw": w["WRITE_SIZE"] * 1024 / nfwd / 1e9,
  "conv_hbm_bytes_per_launch_raw": (f["FETCH_SIZE"] + w["WRITE_SIZE"]) * 1024 / nconv,
  "conv_hbm_bytes_per_launch_fetch_x2": (2 * f["FETCH_SIZE"] + w["WRITE_SIZE"]) * 1024 / nconv,
  "mfma_busy_frac_of_active": m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (m["GRBM_GUI_ACTIVE"] / 8.0),
  "mfma_flops_per_forward": i["SQ_INSTS_VALU_MFMA_MOPS_F32"] * 512 / nfwd,
  "lds_bank_conflict_frac": i["SQ_LDS_BANK_CONFLICT"] / max(1.0, i["SQ_LDS_IDX_ACTIVE"]),
  "split_launches": nsplit,
  "split_fetch_GB_per_forward_raw": fs["FETCH_SIZE"] * 1024 / nfwd / 1e9,
  "split_write_GB_per_forward_raw": ws["WRITE_SIZE"] * 1024 / nfwd / 1e9,
  "split_hbm_bytes_per_launch_fetch_x2": (2 * fs["FETCH_SIZE"] + ws["WRITE_SIZE"]) * 1024 / max(1, nsplit),
  "split_mfma_busy_cycles_share_of_conv": ms["SQ_VALU_MFMA_BUSY_CYCLES"] / max(1.0, m["SQ_VALU_MFMA_BUSY_CYCLES"]),
  "note": "FETCH_SIZE on gfx950 under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md HBM section): "
          "raw and fetch-doubled figures both given; WRITE_SIZE uncalibrated. mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES/1024 SIMDs "
          "over GRBM_GUI_ACTIVE/8 XCDs. Algorithmic conv traffic: 6.83 GB/frame * 8 = 54.6 GB/forward (27.3 read + 27.3 write).",
}
# per kernel family: MFMA-busy cycles against CU-busy cycles (same pass), HBM bytes per launch
fams = {"conv_stem_kernel": lambda k: "conv_stem_kernel" in k, "conv_h2k_kernel": lambda k: "conv_h2k_kernel" in k, "conv_h2_kernel": lambda k: "conv_h2_kernel" in k,
        "conv_split3(k)_kernel": lambda k: "conv_split3" in k, "conv_split_kernel (one-stage)": lambda k: "conv_split_kernel" in k,
        "conv_igemm_kernel": lambda k: "conv_igemm_kernel" in k}
res["by_kernel"] = {}
for name, pred in fams.items():
  mm, n = agg(load("SQ_VALU_MFMA"), pred)
  ff, _ = agg(load("FETCH_SIZE"), pred)
  ww, _ = agg(load("WRITE_SIZE"), pred)
  if n == 0:
    continue
  res["by_kernel"][name] = {
      "launches": n,
      "mfma_busy_cycles_per_cu_busy_cycle": mm["SQ_VALU_MFMA_BUSY_CYCLES"] / max(1.0, mm["SQ_BUSY_CU_CYCLES"]),
      "fetch_GB_per_forward_raw": ff["FETCH_SIZE"] * 1024 / nfwd / 1e9, "write_GB_per_forward_raw": ww["WRITE_SIZE"] * 1024 / nfwd / 1e9}
json.dump(res, open(out + ".json", "w"), indent=1)
with open(out + ".txt", "w") as fh:
  for k, v in res.items():
    fh.write("%-40s %s\n" % (k, v))
print(open(out + ".txt").read())
