#!/bin/bash
# round 3: the measured parity report (three arithmetic modes, configs #2 and #3 at 1080p) + the split kernels against f64 at model shapes
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_parity_report.py -q -m gpu -x 2>&1 | tail -5 | tee gpurun_out/r3_parity_pytest.log
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03_parity.json'))
for c,v in d['configs'].items():
  for m,r in v.items():
    print(c,m,'stage max %.2e'%max(r['stage_max_rel_err'].values()),'det',r['detections']['count'],r['detections']['unmatched_ours'],r['detections']['unmatched_oracle'],'box %.2e'%r['detections']['max_box_diff_px'],'prob %.2e'%r['detections']['max_prob_diff'],'prop box %.2e'%r['proposals']['max_box_diff_px'], 'feat', r.get('fpn_box_feat_max_rel_err'), 'split', r['split_conv_launches'], 'fp16x2', r['fp16x2_conv_launches'])
PY
