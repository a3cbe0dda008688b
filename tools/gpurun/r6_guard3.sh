#!/bin/bash
# Round 6, guard v2 (|max| growth watch, no kernel code): tests on hip, headline vs the round-5 tree and vs the library with the
# producers' counting compiled out (should be the same code now), detect + track / pipelined legs.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$(pwd)
timeout 1500 python -m pytest tests/test_e2e.py tests/test_ingest.py tests/test_drop_in.py tests/test_abi.py -q -m gpu -x -k "guard or auto or ingest or drop or abi or trained_like or stream or describe" 2>&1 | tail -4 | tee gpurun_out/r06j_pytest_guard.log
line() { python -c "
import sys,json; d=json.load(sys.stdin); print('value %.1f  ms %.3f' % (d['value'], d['ms_per_step']))"; }
for rep in 1 2; do
cd $R/wt_r05; echo "[r05 tree] $(timeout 600 python bench.py --steps 20 --warmup 5 --no-d7 --no-cpu-baseline --no-extras --no-nn-matching 2>/dev/null | tail -1 | line)"
cd $R; echo "[r06 final] $(timeout 600 python bench.py --steps 20 --warmup 5 --no-d7 --no-cpu-baseline --no-extras --no-nn-matching 2>/dev/null | tail -1 | line)"
done 2>&1 | tee gpurun_out/r06j_guard_v2_cost_ab.txt
cd $R
(timeout 900 python bench.py --steps 20 --warmup 5 --no-d7 --no-cpu-baseline 2>gpurun_out/r06j_bench_err.log | tail -1) > gpurun_out/r06j_bench_n1.json
python -c "
import json; d=json.load(open('gpurun_out/r06j_bench_n1.json')); e=d['extra']; r=d['roofline']
print('value %.2f frac %.4f' % (d['value'], r['frac']), 'traffic', r.get('traffic'), (r.get('traffic_source') or '')[:120])
print({k: (round(v, 2) if isinstance(v, float) else v) for k, v in e.items() if not isinstance(v, dict)})
print(d['handle'].get('conv_split_family_auto', {}).get('watch'))"
