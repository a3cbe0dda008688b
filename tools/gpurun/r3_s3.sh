#!/bin/bash
# round 3, session 3: bf16x3 planes for the conv2 -> conv3 tensors: correctness (bit equality) + same-box A/B
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_e2e.py -q -m gpu -x -k "planes or rpn_head_fused or multi_r101_b2 or single_r101_256" 2>&1 | tail -5 | tee gpurun_out/r3_s3_pytest.log
q() { python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline "$@" 2>>gpurun_out/r3_s3_err.log | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('%-28s fps %.2f  ms/step %.3f  split-family %.1f TF frac %.4f conv_ms %.3f verified %s planes %s crc %s' % (sys.argv[1], d['value'], d['ms_per_step'], r['achieved'], r['frac'], r['conv_ms_per_step'], d['verified'], d['handle'].get('memory', {}).get('bf16x3_plane_tensors'), d['verification']['streams'][0]['checksum_crc32']))" "$TAG"; }
for rep in 1 2; do
  TAG="planes=1" ODT_CONV_PLANES=1 q | tee -a gpurun_out/r3_s3_ab.txt
  TAG="planes=0" ODT_CONV_PLANES=0 q | tee -a gpurun_out/r3_s3_ab.txt
done
for v in 1 0; do
  ODT_CONV_PLANES=$v timeout 300 python tools/profile_layers.py --batch 8 --steps 3 2>&1 | tail -45 > gpurun_out/r3_s3_layers_planes$v.txt
done
paste <(cut -c1-46,60-100 gpurun_out/r3_s3_layers_planes1.txt) <(cut -c78-100 gpurun_out/r3_s3_layers_planes0.txt) | head -30
