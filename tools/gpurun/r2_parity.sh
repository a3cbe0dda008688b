#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_parity_report.py tests/test_e2e.py -q -m gpu 2>&1 | tail -15
python - <<'PY'
import json
d = json.load(open("gpurun_out/r02_parity.json"))
for c, modes in d["configs"].items():
  for m, r in modes.items():
    print(c, m, "stage max %.2e" % max(r["stage_max_rel_err"].values()), "props", r["proposals"], "dets", r["detections"], "feat", r.get("fpn_box_feat_max_rel_err"))
PY
