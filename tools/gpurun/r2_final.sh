#!/bin/bash
# round-2 validation: GPU suite, smoke, bench b=8 (with extras + CPU baseline) and b=1, layer tables, rocprofv3 kernel stats + PMC passes
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/ -q -m gpu 2>&1 | tail -4 | tee gpurun_out/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke | tee gpurun_out/smoke.log
timeout 420 python bench.py --steps 20 --warmup 3 2>gpurun_out/bench_err.log | tail -1 > gpurun_out/bench_b8.json
(timeout 300 python bench.py --batch 1 --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1) > gpurun_out/bench_b1.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_b8.json"))
print("b8 FPS %.1f  split %.1f TF frac %.3f" % (d["value"], d["roofline"]["achieved"], d["roofline"]["frac"]))
print(json.dumps({k: v for k, v in d["extra"].items() if not isinstance(v, dict)})[:900])
d1 = json.load(open("gpurun_out/bench_b1.json")); print("b1 FPS %.1f" % d1["value"])
PY
(timeout 300 python tools/profile_layers.py --batch 8 --steps 3 2>&1 | tail -45) > gpurun_out/layers_b8.txt 2>&1
(timeout 300 python tools/profile_layers.py --batch 1 --steps 3 2>&1 | tail -45) > gpurun_out/layers_b1.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_r02
(timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r02 -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --profile-steps 1 2>&1 | tail -2) > $R/gpurun_out/rocprof.log 2>&1
cd $R
python tools/kernel_stats.py gpurun_out/prof_r02 > gpurun_out/kernel_stats_b8.txt 2>&1
find gpurun_out/prof_r02 -name "*.db" -size +20M -delete
head -12 gpurun_out/kernel_stats_b8.txt | cut -c1-170
cd /tmp
CMD="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --profile-steps 1"
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-40)
  rm -rf $R/gpurun_out/pmc_$tag
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_$tag -o pmc -- $CMD > $R/gpurun_out/pmc_$tag.log 2>&1
done
cd $R
python tools/pmc_summary.py gpurun_out gpurun_out/r02_pmc_summary_split > gpurun_out/pmc_summary.log 2>&1
find gpurun_out -name "*.db" -size +20M -delete; find gpurun_out -name "*.csv" -size +20M -delete
grep -E "fetch_GB|write_GB|mfma_busy|hbm_bytes_per_launch_fetch" gpurun_out/r02_pmc_summary_split.txt | cut -c1-120
# ---- EfficientDet-D7 (config #5): bench incl. the detect + TMOT leg and the CPU restatement, per-kernel and per-layer tables
(timeout 600 python tools/bench_efficientdet.py --steps 20 2>gpurun_out/effdet_err.log | tail -1) > gpurun_out/bench_efficientdet_d7.json
python -c "
import json; d=json.load(open('gpurun_out/bench_efficientdet_d7.json')); print('D7 FPS %.1f ms %.2f frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac']), d['extra'], d['cpu_baseline'])"
(timeout 300 python tools/profile_layers.py --effdet efficientdet-d7 --steps 3 2>&1 | tail -45) > gpurun_out/effdet_d7_conv_layers.txt 2>&1
cd /tmp
rm -rf $R/gpurun_out/prof_effdet
(timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_effdet -o eff -- python $R/tools/bench_efficientdet.py --no-cpu-baseline --steps 10 --warmup 2 2>&1 | tail -1) > $R/gpurun_out/effdet_rocprof.log 2>&1
cd $R
python tools/kernel_stats.py gpurun_out/prof_effdet > gpurun_out/kernel_stats_effdet_d7.txt 2>&1
find gpurun_out/prof_effdet -name "*.db" -size +20M -delete
head -8 gpurun_out/kernel_stats_effdet_d7.txt | cut -c1-170
(timeout 280 python tools/profile_detect_track.py 2>&1 | grep -E "arrays=|pieces|idle" ) > gpurun_out/detect_track_pieces.txt 2>&1
cat gpurun_out/detect_track_pieces.txt
