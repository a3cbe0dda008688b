#!/bin/bash
# round 3: range record once per workgroup (default) against once per wave (ODT_AMAX_PER_WAVE=1)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_e2e.py -q -m gpu -x -k "fp16x2 or multi_r101_b2" 2>&1 | tail -3
q() { timeout 300 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline --no-d7 "$@" 2>>gpurun_out/r3_h2_14_err.log | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('%-10s fps %.2f  ms/step %.3f  conv_ms %.3f verified %s crc %s' % (sys.argv[1], d['value'], d['ms_per_step'], r['conv_ms_per_step'], d['verified'], d['verification']['streams'][0]['checksum_crc32']))" "$TAG"; }
for rep in 1 2 3; do
  TAG="per_wg" q | tee -a gpurun_out/r3_h2_14_ab.txt
  TAG="per_wave" ODT_AMAX_PER_WAVE=1 q | tee -a gpurun_out/r3_h2_14_ab.txt
done
