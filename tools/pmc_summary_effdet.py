#!/usr/bin/env python
"""HBM traffic of the EfficientDet forward from two rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE, collected by
tools/gpurun/r2_effdet_pmc.sh): GB per forward, all kernels and per kernel family.  Units / corrections as in
tools/pmc_summary.py (counter values in KB; FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950)."""
import collections, csv, json, os, sys
src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
out = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/r02_pmc_summary_effdet_d7"
def load(tag):
  d = [x for x in os.listdir(src) if x.startswith("pmceff_" + tag) and os.path.isdir(os.path.join(src, x))][0]
  return list(csv.DictReader(open(os.path.join(src, d, "pmc_counter_collection.csv"))))
def fam(k):
  for key, name in (("mbconv_expand_dw", "MBConv expand + depthwise (one kernel)"), ("dwconv", "depthwise"), ("conv_split", "1x1 conv (bf16x3 split)"), ("conv_igemm", "1x1 conv (exact f32)"),
                    ("split_reduce", "split-K reduce"), ("split_weights", "gate -> weights"), ("scale_weights", "gate -> weights"),
                    ("bifpn_fuse", "BiFPN fusion"), ("se_", "squeeze-excite gate"), ("channel_mean", "squeeze-excite gate"),
                    ("channel_scale", "squeeze-excite scale"), ("eff_", "tail"), ("roi_", "tail"), ("preprocess", "preprocess")):
    if key in k:
      return name
  return "other"
tot = {"FETCH_SIZE": collections.defaultdict(float), "WRITE_SIZE": collections.defaultdict(float)}
nfwd = 0
for tag in ("FETCH_SIZE", "WRITE_SIZE"):
  rows = load(tag)
  if tag == "FETCH_SIZE":
    nfwd = len({r["Dispatch_Id"] for r in rows if "preprocess_rgb" in r["Kernel_Name"]})
  for r in rows:
    if r["Counter_Name"] == tag:
      tot[tag][fam(r["Kernel_Name"])] += float(r["Counter_Value"])
res = {"forwards_profiled": nfwd, "families": {}}
for f in sorted(set(tot["FETCH_SIZE"]) | set(tot["WRITE_SIZE"])):
  res["families"][f] = {"fetch_GB_per_forward_raw": tot["FETCH_SIZE"][f] * 1024 / nfwd / 1e9,
                        "write_GB_per_forward_raw": tot["WRITE_SIZE"][f] * 1024 / nfwd / 1e9}
res["fetch_GB_per_forward_raw"] = sum(v["fetch_GB_per_forward_raw"] for v in res["families"].values())
res["write_GB_per_forward_raw"] = sum(v["write_GB_per_forward_raw"] for v in res["families"].values())
res["hbm_GB_per_forward_fetch_x2"] = 2 * res["fetch_GB_per_forward_raw"] + res["write_GB_per_forward_raw"]
res["note"] = ("raw counter sums (KB -> GB); FETCH_SIZE on gfx950 under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md), "
               "fetch-doubled total given as well; algorithmic traffic of the unfused graph: 29.4 GB per D7 frame, of the best fusion 11.3 GB")
json.dump(res, open(out + ".json", "w"), indent=1)
print(json.dumps(res, indent=1))
