#!/bin/bash
# The round's evidence from ONE box, in budget-sized parts (each optional):
#   ROUND=r04 PARTS="tests bench b1 layers probe rocprof pmc d7pmc" bash tools/gpurun/evidence.sh
#   tests   : the whole -m gpu suite + smoke()
#   bench   : the full bench line (headline + extras + EfficientDet-D7 leg + CPU baseline)
#   b1      : b = 1 on both graphs (bench lines of their own)
#   layers  : per-layer tables (b = 8, b = 1) + per-layer roofline table
#   probe   : what the matrix pipe of this box sustains on the kernels' MFMA mixes
#   rocprof : rocprofv3 --kernel-trace --stats of the bench command
#   pmc     : the separate --pmc passes of the bench command (FETCH_SIZE | WRITE_SIZE | MFMA busy | LDS / wait)
#   d7pmc   : HBM traffic counters of the EfficientDet-D7 forward (two --pmc passes) + its kernel trace
#   n2      : the N-rank launch path on the one GPU of the box (two ranks over gloo sharing device 0)
# Everything lands in gpurun_out/<ROUND>_*; copy what should be judged into profiles/.
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-$(pwd)}
N=${ROUND:-r05}
PARTS=${PARTS:-"tests bench b1 layers probe rocprof pmc"}
export TMPDIR=/tmp
has() { case " $PARTS " in *" $1 "*) return 0;; *) return 1;; esac; }
if has tests; then
  timeout 2400 python -m pytest tests/ -q -m gpu -x 2>&1 | tail -8 | tee gpurun_out/${N}_pytest_gpu.log
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -7 | tee gpurun_out/${N}_smoke.log
fi
if has bench; then
  (timeout 1200 python bench.py --steps 20 --warmup 5 2>gpurun_out/${N}_bench_err.log | tail -1) > gpurun_out/${N}_bench_n1.json
  python -c "
import json; d=json.load(open('gpurun_out/${N}_bench_n1.json')); r=d['roofline']
print('b8 FPS %.2f  frac %.4f  frac_of_sustained %.4f  products/MAC %.3f verified %s' % (d['value'], r['frac'], r.get('frac_of_sustained', 0), r.get('products_per_mac', 0), d['verified']))
e=d['extra']; print({k: (round(v, 2) if isinstance(v, float) else v) for k, v in e.items() if not isinstance(v, dict)})
b=e.get('b1_single_graph', {}); print('b1 single', e.get('b1_single_graph_fps'), b.get('verified'), (b.get('roofline') or {}).get('frac'))
x=e.get('efficientdet_d7', {}); print('D7', x.get('value'), x.get('verified'), x.get('roofline', {}).get('frac'), x.get('extra', {}).get('detect_tmot_pipelined_fps'), (x.get('cpu_baseline') or {}).get('value'))"
fi
if has b1; then
  (timeout 300 python bench.py --batch 1 --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1) > gpurun_out/${N}_bench_n1_b1.json
  (timeout 300 python bench.py --batch 1 --graph single --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1) > gpurun_out/${N}_bench_n1_b1_single.json
  python -c "
import json
for f in ('b1', 'b1_single'):
  d=json.load(open('gpurun_out/${N}_bench_n1_%s.json' % f)); print(f, d['config']['graph'], 'FPS %.2f verified %s frac %.3f' % (d['value'], d['verified'], d['roofline']['frac']))"
fi
if has layers; then
  (timeout 300 python tools/profile_layers.py --batch 8 --steps 3 2>&1 | tail -45) > gpurun_out/${N}_conv_layers_b8.txt 2>&1
  python tools/roofline_table.py gpurun_out/${N}_conv_layers_b8.txt > gpurun_out/${N}_roofline_per_layer_b8.txt; tail -1 gpurun_out/${N}_roofline_per_layer_b8.txt
  (timeout 300 python tools/profile_layers.py --batch 1 --steps 5 2>&1 | tail -60) > gpurun_out/${N}_conv_layers_b1.txt 2>&1; tail -1 gpurun_out/${N}_conv_layers_b1.txt
fi
if has probe; then
  (timeout 120 python tools/probe_sustained.py) > gpurun_out/${N}_mfma_sustained_probe.txt 2>&1; cat gpurun_out/${N}_mfma_sustained_probe.txt
fi
if has rocprof; then
  cd /tmp; rm -rf $R/gpurun_out/prof_${N}
  (timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${N} -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-d7 --profile-steps 1 2>&1 | tail -2) > $R/gpurun_out/${N}_rocprof.log 2>&1
  cd $R
  python tools/kernel_stats.py gpurun_out/prof_${N} > gpurun_out/${N}_kernel_stats_bench_b8_1080p.txt 2>&1
  find gpurun_out/prof_${N} -name "*.db" -size +20M -delete
  head -16 gpurun_out/${N}_kernel_stats_bench_b8_1080p.txt | cut -c1-170
fi
if has pmc; then
  cd /tmp
  CMD="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-d7 --profile-steps 1"
  for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"; do
    tag=$(echo $c | tr ' ' '_' | cut -c1-40)
    rm -rf $R/gpurun_out/pmc_$tag
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_$tag -o pmc -- $CMD > $R/gpurun_out/pmc_$tag.log 2>&1
  done
  cd $R
  python tools/pmc_summary.py gpurun_out gpurun_out/${N}_pmc_summary_split > gpurun_out/${N}_pmc_summary.log 2>&1
  cat gpurun_out/${N}_pmc_summary_split.txt | cut -c1-200
  python tools/pmc_by_kernel.py gpurun_out > gpurun_out/${N}_pmc_lds_wait_by_kernel.txt 2>&1; cut -c1-220 gpurun_out/${N}_pmc_lds_wait_by_kernel.txt
  find gpurun_out -name "*.csv" -size +20M -delete; find gpurun_out -name "*.db" -size +20M -delete
fi
if has d7pmc; then
  cd /tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $R/gpurun_out/pmceff_$c
    timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmceff_$c -o pmc -- python $R/tools/bench_efficientdet.py --no-cpu-baseline --steps 2 --warmup 1 > $R/gpurun_out/pmceff_$c.log 2>&1
  done
  rm -rf $R/gpurun_out/prof_d7
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_d7 -o d7 -- python $R/tools/bench_efficientdet.py --no-cpu-baseline --steps 10 --warmup 2 > $R/gpurun_out/d7_prof.log 2>&1
  cd $R
  python tools/pmc_summary_effdet.py gpurun_out gpurun_out/${N}_pmc_summary_effdet_d7 2>&1 | tail -12
  python tools/kernel_stats.py gpurun_out/prof_d7 > gpurun_out/${N}_kernel_stats_efficientdet_d7.txt 2>&1
  find gpurun_out -name "*.csv" -size +20M -delete; find gpurun_out -name "*.db" -size +20M -delete
fi
if has n2; then
  (timeout 600 python bench.py --gpus 2 --dist-backend gloo --device 0 --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-d7 2>gpurun_out/${N}_bench_n2_err.log | tail -1) > gpurun_out/${N}_bench_n2_gloo_one_gpu.json
  python -c "
import json; d=json.load(open('gpurun_out/${N}_bench_n2_gloo_one_gpu.json'))
print('n_gpus %d ranks_seen %s total FPS %.2f per rank %s verified %s' % (d['n_gpus'], d['ranks_seen'], d['value'], [round(v, 1) for v in d['per_rank_fps']], d['verified']))"
fi
