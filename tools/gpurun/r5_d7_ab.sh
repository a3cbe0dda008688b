#!/bin/bash
# Round 5: EfficientDet-D7 @1536 same-box A/B of the fused MBConv front half (ODT_EFFDET_FUSE_MB: 0 off, 1 maps >= MIN, 2 all)
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_efficientnet.py -q -m gpu -x 2>&1 | tail -4
run() { env $1 timeout 300 python tools/bench_efficientdet.py --no-cpu-baseline --steps 30 --warmup 5 2>gpurun_out/d7_err.txt | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%.2f FPS  %.3f ms  verified %s  fused blocks %s  launches %s' % (d['value'], d['ms_per_step'], d['verified'], d['extra'].get('handle',{}).get('mbconv_expand_dw_fused'), d['extra'].get('handle',{}).get('conv_launches')))" || tail -5 gpurun_out/d7_err.txt; }
for rep in 1 2; do for v in ${VARIANTS:-"ODT_EFFDET_FUSE_MB=0,ODT_EFFDET_SE_FUSED=0" "ODT_EFFDET_FUSE_MB=0" "ODT_EFFDET_FUSE_MB=1" "ODT_EFFDET_FUSE_MB=1,ODT_EFFDET_FUSE_MB_MIN=192" "ODT_EFFDET_FUSE_MB=2"}; do
  v=$(echo $v | tr ',' ' ')
  echo "[$v] rep$rep: $(run "$v")"
done; done | tee gpurun_out/${OUT:-r05_effdet_mbconv_fusion_ab}.txt
# kernel trace of the default
cd /tmp; rm -rf $R/gpurun_out/prof_d7
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_d7 -o d7 -- python $R/tools/bench_efficientdet.py --no-cpu-baseline --steps 10 --warmup 2 > $R/gpurun_out/d7_prof.log 2>&1
cd $R
python tools/kernel_stats.py gpurun_out/prof_d7 > gpurun_out/r05_kernel_stats_efficientdet_d7.txt 2>&1
head -24 gpurun_out/r05_kernel_stats_efficientdet_d7.txt | cut -c1-150
find gpurun_out/prof_d7 -name "*.db" -size +20M -delete
