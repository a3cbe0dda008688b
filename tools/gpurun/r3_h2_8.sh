#!/bin/bash
# round 3: fp16x2 on the 64-wide layers (ODT_CONV_H2_N64): parity + per-layer A/B + bench
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops.py tests/test_e2e.py -q -m gpu -x -k "(split and (2/256 or 2/128)) or fp16x2 or multi_r101_b2 or single_r101_odd or preprocess" 2>&1 | tail -4 | tee gpurun_out/r3_h2_8_pytest.log
for n64 in 1 0; do
  ODT_CONV_H2_N64=$n64 timeout 300 python tools/profile_layers.py --batch 8 --steps 3 2>&1 | tail -45 > gpurun_out/r3_h2_8_layers_n64_$n64.txt
  echo "n64=$n64: $(tail -1 gpurun_out/r3_h2_8_layers_n64_$n64.txt)"
  grep -E "^conv0|group0" gpurun_out/r3_h2_8_layers_n64_$n64.txt | awk '{printf "   %-42s %7s %6s\n",$1,$6,$7}'
done
q() { timeout 300 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline --no-d7 "$@" 2>>gpurun_out/r3_h2_8_err.log | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('%-10s fps %.2f  ms/step %.3f  conv_ms %.3f frac %.4f of_sustained %.4f products %.3f verified %s crc %s' % (sys.argv[1], d['value'], d['ms_per_step'], r['conv_ms_per_step'], r['frac'], r.get('frac_of_sustained', 0), r.get('products_per_mac', 0), d['verified'], d['verification']['streams'][0]['checksum_crc32']))" "$TAG"; }
for rep in 1 2; do
  TAG="n64=1" ODT_CONV_H2_N64=1 q | tee -a gpurun_out/r3_h2_8_ab.txt
  TAG="n64=0" ODT_CONV_H2_N64=0 q | tee -a gpurun_out/r3_h2_8_ab.txt
done
