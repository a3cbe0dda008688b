#!/bin/bash
# round 3, session 5: pool0 in conv0's epilogue: correctness + same-box A/B
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_e2e.py -q -m gpu -x -k "pool0_fused or multi_r101_b2 or single_r101_odd" 2>&1 | tail -5 | tee gpurun_out/r3_s5_pytest.log
q() { python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline "$@" 2>>gpurun_out/r3_s5_err.log | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('%-16s fps %.2f  ms/step %.3f  split-family %.1f TF frac %.4f conv_ms %.3f verified %s pool %s crc %s' % (sys.argv[1], d['value'], d['ms_per_step'], r['achieved'], r['frac'], r['conv_ms_per_step'], d['verified'], d['handle'].get('pool0_in_conv0_epilogue'), d['verification']['streams'][0]['checksum_crc32']))" "$TAG"; }
for rep in 1 2; do
  TAG="fuse_pool=1" ODT_FUSE_POOL=1 q | tee -a gpurun_out/r3_s5_ab.txt
  TAG="fuse_pool=0" ODT_FUSE_POOL=0 q | tee -a gpurun_out/r3_s5_ab.txt
done
ODT_FUSE_POOL=1 timeout 300 python tools/profile_layers.py --batch 8 --steps 3 2>&1 | tail -45 > gpurun_out/r3_s5_layers.txt
grep -E "conv0|conv total" gpurun_out/r3_s5_layers.txt
