#!/bin/bash
# round 3: conv_h2d_kernel (activations by LDS-DMA) -- parity on the GPU, per-layer profile and same-box A/B against the
# register-prefetch loop (ODT_CONV_H2_ADMA=0)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops.py tests/test_e2e.py -q -m gpu -x -k "(split and 2/256) or fp16x2 or matches_f32_kernel" 2>&1 | tail -6 | tee gpurun_out/r3_h2_2_pytest.log
for adma in 1 0; do
  ODT_CONV_H2_ADMA=$adma timeout 300 python tools/profile_layers.py --batch 8 --steps 3 2>&1 | tail -45 > gpurun_out/r3_h2_2_layers_adma$adma.txt
  tail -1 gpurun_out/r3_h2_2_layers_adma$adma.txt
done
q() { timeout 300 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline --no-d7 "$@" 2>>gpurun_out/r3_h2_2_err.log | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('%-10s fps %.2f  ms/step %.3f  conv_ms %.3f frac %.4f of_sustained %s verified %s fp16x2 launches %s crc %s' % (sys.argv[1], d['value'], d['ms_per_step'], r['conv_ms_per_step'], r['frac'], r.get('frac_of_sustained'), d['verified'], d['handle'].get('fp16x2_split_launches'), d['verification']['streams'][0]['checksum_crc32']))" "$TAG"; }
for rep in 1 2; do
  TAG="adma=1" ODT_CONV_H2_ADMA=1 q | tee -a gpurun_out/r3_h2_2_ab.txt
  TAG="adma=0" ODT_CONV_H2_ADMA=0 q | tee -a gpurun_out/r3_h2_2_ab.txt
done
