#!/bin/bash
# round 3: A-plane pad fix (LDS store bank conflicts), range slots only for family 2 (EfficientDet back to no epilogue atomics): parity + layers + bench + D7
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops.py tests/test_e2e.py -q -m gpu -x -k "(split and (2/256 or 2/128)) or fp16x2" 2>&1 | tail -4 | tee gpurun_out/r3_h2_6_pytest.log
timeout 300 python tools/profile_layers.py --batch 8 --steps 3 2>&1 | tail -45 > gpurun_out/r3_h2_6_layers_b8.txt; tail -1 gpurun_out/r3_h2_6_layers_b8.txt
q() { timeout 300 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline --no-d7 "$@" 2>>gpurun_out/r3_h2_6_err.log | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('%-16s fps %.2f  ms/step %.3f  conv_ms %.3f frac %.4f of_sustained %s verified %s fp16x2 launches %s crc %s' % (sys.argv[1], d['value'], d['ms_per_step'], r['conv_ms_per_step'], r['frac'], r.get('frac_of_sustained'), d['verified'], d['handle'].get('fp16x2_split_launches'), d['verification']['streams'][0]['checksum_crc32']))" "$TAG"; }
TAG="b8" q | tee -a gpurun_out/r3_h2_6_ab.txt
TAG="b8" q | tee -a gpurun_out/r3_h2_6_ab.txt
TAG="b1" q --batch 1 | tee -a gpurun_out/r3_h2_6_ab.txt
timeout 300 python tools/bench_efficientdet.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('D7 fps %.2f' % d['value'])" | tee -a gpurun_out/r3_h2_6_ab.txt
