#!/bin/bash
# round 3: fp16x2 policy as committed (few-tile layers on 128 x 128 tiles, split-K below res3 at b=1): parity, per-layer profiles
# at b=8 and b=1 (against conv_split_family 3), quick bench lines
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops.py tests/test_e2e.py -q -m gpu -x -k "(split and (2/256 or 2/128 or 3/128)) or fp16x2" 2>&1 | tail -4 | tee gpurun_out/r3_h2_5_pytest.log
timeout 300 python tools/profile_layers.py --batch 8 --steps 3 2>&1 | tail -45 > gpurun_out/r3_h2_5_layers_b8.txt; tail -1 gpurun_out/r3_h2_5_layers_b8.txt
for fam in 2 3; do
  ODT_CONV_SPLIT_PIPE=$fam timeout 300 python tools/profile_layers.py --batch 1 --steps 5 2>&1 | tail -60 > gpurun_out/r3_h2_5_layers_b1_fam$fam.txt
  echo "b=1 family=$fam: $(tail -1 gpurun_out/r3_h2_5_layers_b1_fam$fam.txt)"
done
q() { timeout 300 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline --no-d7 "$@" 2>>gpurun_out/r3_h2_5_err.log | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('%-16s fps %.2f  ms/step %.3f  conv_ms %.3f frac %.4f of_sustained %s verified %s fp16x2 launches %s crc %s' % (sys.argv[1], d['value'], d['ms_per_step'], r['conv_ms_per_step'], r['frac'], r.get('frac_of_sustained'), d['verified'], d['handle'].get('fp16x2_split_launches'), d['verification']['streams'][0]['checksum_crc32']))" "$TAG"; }
TAG="b8 family=2" q | tee -a gpurun_out/r3_h2_5_ab.txt
TAG="b1 family=2" q --batch 1 | tee -a gpurun_out/r3_h2_5_ab.txt
TAG="b1 family=3" ODT_CONV_SPLIT_PIPE=3 q --batch 1 | tee -a gpurun_out/r3_h2_5_ab.txt
