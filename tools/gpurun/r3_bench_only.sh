#!/bin/bash
# the driver's bench line on the final code
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 900 python bench.py --steps 20 --warmup 5 2>gpurun_out/r03_bench_err.log | tail -1) > gpurun_out/r03_bench_n1.json
python -c "
import json; d=json.load(open('gpurun_out/r03_bench_n1.json')); r=d['roofline']
print('b8 FPS %.2f  frac %.4f  frac_of_sustained %.4f  products/MAC %.3f verified %s' % (d['value'], r['frac'], r.get('frac_of_sustained', 0), r.get('products_per_mac', 0), d['verified']))
e=d['extra']; print({k: (round(v, 2) if isinstance(v, float) else v) for k, v in e.items() if not isinstance(v, dict)})
x=e.get('efficientdet_d7', {}); print('D7', x.get('value'), x.get('roofline', {}).get('frac'), x.get('extra', {}).get('detect_tmot_pipelined_fps'), (x.get('cpu_baseline') or {}).get('value'))"
