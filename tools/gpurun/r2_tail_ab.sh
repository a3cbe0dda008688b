#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ingest.py tests/test_e2e.py tests/test_drop_in.py -q -m gpu -x 2>&1 | tail -3
for v in "ODT_TAIL_OVERLAP=0" "ODT_TAIL_OVERLAP=1" "ODT_TAIL_OVERLAP=0" "ODT_TAIL_OVERLAP=1"; do
  r8=$(env $v timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f FPS all-conv %.1f TF split %.1f TF' % (d['value'], d['roofline']['all_conv_launches']['achieved'], d['roofline']['achieved']))")
  echo "$v  b8: $r8"
done | tee gpurun_out/tail_overlap_ab.txt
