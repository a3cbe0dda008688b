#!/bin/bash
# round 3: non-temporal hints on by default: parity subset, bench A/B (FPN b=8, EfficientDet-D7)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops.py tests/test_e2e.py tests/test_efficientnet.py -q -m gpu -x -k "(split and (2/256 or 3/256 or 1)) or fp16x2 or multi_r101_b2 or d0" 2>&1 | tail -4 | tee gpurun_out/r3_h2_11_pytest.log
q() { timeout 300 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline --no-d7 "$@" 2>>gpurun_out/r3_h2_11_err.log | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('%-10s fps %.2f  ms/step %.3f  conv_ms %.3f frac %.4f of_sustained %.4f verified %s crc %s' % (sys.argv[1], d['value'], d['ms_per_step'], r['conv_ms_per_step'], r['frac'], r.get('frac_of_sustained', 0), d['verified'], d['verification']['streams'][0]['checksum_crc32']))" "$TAG"; }
e() { timeout 300 python tools/bench_efficientdet.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('%-10s D7 fps %.2f' % (sys.argv[1], d['value']))" "$TAG"; }
for rep in 1 2; do
  TAG="nt=3" q | tee -a gpurun_out/r3_h2_11_ab.txt
  TAG="nt=0" ODT_CONV_NT=0 q | tee -a gpurun_out/r3_h2_11_ab.txt
done
TAG="nt=3" e | tee -a gpurun_out/r3_h2_11_ab.txt
TAG="nt=0" ODT_CONV_NT=0 e | tee -a gpurun_out/r3_h2_11_ab.txt
TAG="nt=3" e | tee -a gpurun_out/r3_h2_11_ab.txt
TAG="nt=0" ODT_CONV_NT=0 e | tee -a gpurun_out/r3_h2_11_ab.txt
