#!/bin/bash
# round 3 evidence, fp16x2 default: GPU suite + smoke, the full bench line, b=1 on both graphs, per-layer roofline tables,
# sustained-peak probe (incl. the fp16x2 mix), rocprofv3 kernel-trace stats + separate --pmc passes (FPN b=8)
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/ -q -m gpu -x 2>&1 | tail -8 | tee gpurun_out/r03_pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -7 | tee gpurun_out/r03_smoke.log
(timeout 900 python bench.py --steps 20 --warmup 5 2>gpurun_out/r03_bench_err.log | tail -1) > gpurun_out/r03_bench_n1.json
python -c "
import json; d=json.load(open('gpurun_out/r03_bench_n1.json')); r=d['roofline']
print('b8 FPS %.2f  frac %.4f  frac_of_sustained %.4f  products/MAC %.3f verified %s' % (d['value'], r['frac'], r.get('frac_of_sustained', 0), r.get('products_per_mac', 0), d['verified']))
e=d['extra']; print({k: (round(v, 2) if isinstance(v, float) else v) for k, v in e.items() if not isinstance(v, dict)})
x=e.get('efficientdet_d7', {}); print('D7', x.get('value'), x.get('roofline', {}).get('frac'), x.get('extra', {}).get('detect_tmot_pipelined_fps'), (x.get('cpu_baseline') or {}).get('value'))"
(timeout 300 python bench.py --batch 1 --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1) > gpurun_out/r03_bench_n1_b1.json
(timeout 300 python bench.py --batch 1 --graph single --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1) > gpurun_out/r03_bench_n1_b1_single.json
python -c "
import json
for f in ('b1', 'b1_single'):
  d=json.load(open('gpurun_out/r03_bench_n1_%s.json' % f)); print(f, d['config']['graph'], 'FPS %.2f verified %s' % (d['value'], d['verified']))"
(timeout 300 python tools/profile_layers.py --batch 8 --steps 3 2>&1 | tail -45) > gpurun_out/r03_conv_layers_b8.txt 2>&1
python tools/roofline_table.py gpurun_out/r03_conv_layers_b8.txt > gpurun_out/r03_roofline_per_layer_b8.txt; tail -1 gpurun_out/r03_roofline_per_layer_b8.txt
(timeout 300 python tools/profile_layers.py --batch 1 --steps 5 2>&1 | tail -60) > gpurun_out/r03_conv_layers_b1.txt 2>&1; tail -1 gpurun_out/r03_conv_layers_b1.txt
(timeout 120 python tools/probe_sustained.py) > gpurun_out/r03_mfma_sustained_probe.txt 2>&1; cat gpurun_out/r03_mfma_sustained_probe.txt
cd /tmp
rm -rf $R/gpurun_out/prof_r03
(timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r03 -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-d7 --profile-steps 1 2>&1 | tail -2) > $R/gpurun_out/r03_rocprof.log 2>&1
cd $R
python tools/kernel_stats.py gpurun_out/prof_r03 > gpurun_out/r03_kernel_stats_bench_b8_1080p.txt 2>&1
find gpurun_out/prof_r03 -name "*.db" -size +20M -delete
head -16 gpurun_out/r03_kernel_stats_bench_b8_1080p.txt | cut -c1-170
cd /tmp
CMD="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-d7 --profile-steps 1"
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-40)
  rm -rf $R/gpurun_out/pmc_$tag
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_$tag -o pmc -- $CMD > $R/gpurun_out/pmc_$tag.log 2>&1
done
cd $R
rm -rf gpurun_out/pmc_SQ_INSTS*
python tools/pmc_summary.py gpurun_out gpurun_out/r03_pmc_summary_split > gpurun_out/r03_pmc_summary.log 2>&1
cat gpurun_out/r03_pmc_summary_split.txt | cut -c1-200
python - <<'PY' > gpurun_out/r03_pmc_lds_wait_by_kernel.txt 2>&1
import csv, os, collections
d = [x for x in os.listdir('gpurun_out') if x.startswith('pmc_SQ_LDS_BANK') and os.path.isdir(os.path.join('gpurun_out', x))][0]
rows = list(csv.DictReader(open(os.path.join('gpurun_out', d, 'pmc_counter_collection.csv'))))
t = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
  k = r['Kernel_Name']
  for fam in ('conv_h2k_kernel', 'conv_h2_kernel', 'conv_split3k_kernel', 'conv_split3_kernel', 'conv_split_kernel', 'conv_igemm_kernel'):
    if fam in k:
      t[fam][r['Counter_Name']] += float(r['Counter_Value']); break
for fam, c in t.items():
  print('%-22s lds_conflict/lds_active %.3f   wait_inst_any/wave_cycles %.3f   %s' % (fam, c['SQ_LDS_BANK_CONFLICT'] / max(1, c['SQ_LDS_IDX_ACTIVE']), c['SQ_WAIT_INST_ANY'] / max(1, c['SQ_WAVE_CYCLES']), dict(c)))
PY
cat gpurun_out/r03_pmc_lds_wait_by_kernel.txt | cut -c1-220
find gpurun_out -name "*.csv" -size +20M -delete; find gpurun_out -name "*.db" -size +20M -delete
