#!/bin/bash
# round 3: residual epilogues in 64-row passes with the next pass's residual chunks in flight (ODT_EPI_FINE): parity + per-layer A/B + bench
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops.py tests/test_e2e.py -q -m gpu -x -k "residual or fp16x2 or multi_r101_b2 or rpn_head" 2>&1 | tail -4 | tee gpurun_out/r3_h2_7_pytest.log
for fine in 1 0; do
  ODT_EPI_FINE=$fine timeout 300 python tools/profile_layers.py --batch 8 --steps 3 2>&1 | tail -45 > gpurun_out/r3_h2_7_layers_fine$fine.txt
  echo "fine=$fine: $(tail -1 gpurun_out/r3_h2_7_layers_fine$fine.txt)"
done
q() { timeout 300 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline --no-d7 "$@" 2>>gpurun_out/r3_h2_7_err.log | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('%-10s fps %.2f  ms/step %.3f  conv_ms %.3f frac %.4f of_sustained %.4f verified %s crc %s' % (sys.argv[1], d['value'], d['ms_per_step'], r['conv_ms_per_step'], r['frac'], r.get('frac_of_sustained', 0), d['verified'], d['verification']['streams'][0]['checksum_crc32']))" "$TAG"; }
for rep in 1 2; do
  TAG="fine=1" ODT_EPI_FINE=1 q | tee -a gpurun_out/r3_h2_7_ab.txt
  TAG="fine=0" ODT_EPI_FINE=0 q | tee -a gpurun_out/r3_h2_7_ab.txt
done
