#!/bin/bash
# round 3, session 2: RPN-head fusion and arena A/Bs (same box), then the whole GPU suite + smoke
mkdir -p gpurun_out
export TMPDIR=/tmp
q() { python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline "$@" 2>>gpurun_out/r3_s2_err.log | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('%-34s fps %.2f  ms/step %.3f  split-family %.1f TF frac %.4f conv_ms %.3f verified %s mem %s' % (sys.argv[1], d['value'], d['ms_per_step'], r['achieved'], r['frac'], r['conv_ms_per_step'], d['verified'], d['handle'].get('memory', {}).get('device_bytes')))" "$TAG"; }
for rep in 1 2; do
  TAG="fuse_rpn_head=1 arena" ODT_FUSE_RPN_HEAD=1 q | tee -a gpurun_out/r3_s2_ab.txt
  TAG="fuse_rpn_head=0 arena" ODT_FUSE_RPN_HEAD=0 q | tee -a gpurun_out/r3_s2_ab.txt
  TAG="fuse_rpn_head=1 keep_taps" ODT_FUSE_RPN_HEAD=1 q --keep-taps | tee -a gpurun_out/r3_s2_ab.txt
done
timeout 300 python tools/profile_layers.py --batch 8 --steps 3 2>&1 | tail -45 > gpurun_out/r3_s2_layers_b8.txt
grep -E "rpn/|conv total" gpurun_out/r3_s2_layers_b8.txt
timeout 2400 python -m pytest tests/ -q -m gpu -x 2>&1 | tail -8 | tee gpurun_out/r3_s2_pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -6 | tee gpurun_out/r3_s2_smoke.log
