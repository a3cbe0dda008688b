#!/usr/bin/env python
"""LDS-conflict and issue-stall ratios per conv kernel family from the SQ_LDS_* / SQ_WAIT_* --pmc pass
(tools/gpurun/evidence.sh, part `pmc`).  usage: python tools/pmc_by_kernel.py gpurun_out"""
import collections, csv, os, sys
root = sys.argv[1]
d = [x for x in os.listdir(root) if x.startswith("pmc_SQ_LDS_BANK") and os.path.isdir(os.path.join(root, x))][0]
rows = list(csv.DictReader(open(os.path.join(root, d, "pmc_counter_collection.csv"))))
t = collections.defaultdict(lambda: collections.defaultdict(float))
import re
FAMS = ("conv_stem_kernel", "conv_h2k_kernel", "conv_h2_kernel", "conv_split3k_kernel", "conv_split3_kernel", "conv_split_kernel", "conv_igemm_kernel")
for r in rows:
  k = r["Kernel_Name"]
  for fam in FAMS:
    if fam in k:
      # conv_h2k_kernel<TN, TRACE, FUSE, WN>: the launches with the fused 1x1 tail are a family of their own
      if fam == "conv_h2k_kernel" and (re.search(r"conv_h2k_kernel<\d+, (true|false), true", k) or re.search(r"conv_h2k_kernelILi\dELb[01]ELb1", k)):
        fam = "conv_h2k_kernel<fused tail>"
      t[fam][r["Counter_Name"]] += float(r["Counter_Value"]); break
for fam, c in t.items():
  label = fam
  print("%-28s lds_conflict/lds_active %.3f   wait_inst_any/wave_cycles %.3f   %s" %
        (label, c["SQ_LDS_BANK_CONFLICT"] / max(1, c["SQ_LDS_IDX_ACTIVE"]), c["SQ_WAIT_INST_ANY"] / max(1, c["SQ_WAVE_CYCLES"]), dict(c)))
