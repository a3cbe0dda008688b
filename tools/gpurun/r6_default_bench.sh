#!/bin/bash
# the driver's invocation: python bench.py with no flags -- wall time and the line
mkdir -p gpurun_out
export TMPDIR=/tmp
s=$(date +%s)
(python bench.py 2>gpurun_out/r06_bench_default_flags_err.log | tail -1) > gpurun_out/r06_bench_default_flags.json
e=$(date +%s)
echo "wall seconds: $((e-s))"
python -c "
import json; d=json.load(open('gpurun_out/r06_bench_default_flags.json')); r=d['roofline']; e=d['extra']
print('value %.2f frac %.4f sust %.4f verified %s' % (d['value'], r['frac'], r.get('frac_of_sustained', 0), d['verified']))
print('traffic', r.get('traffic'), (r.get('traffic_source') or '')[:100])
print({k: (round(v, 2) if isinstance(v, float) else v) for k, v in e.items() if not isinstance(v, dict)})
x=e.get('efficientdet_d7', {}); print('D7', x.get('value'), x.get('verified'))
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])"
tail -3 gpurun_out/r06_bench_default_flags_err.log
