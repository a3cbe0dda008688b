mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_b1
timeout 600 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_b1 -o b1 -- python $GRAFT_REPO_ROOT/bench.py --batch 1 --steps 20 --warmup 3 --no-cpu-baseline --profile-steps 1 2>&1 | tail -2
cd $GRAFT_REPO_ROOT
python tools/kernel_stats.py gpurun_out/prof_b1 | head -30
find gpurun_out/prof_b1 -name "*.db" -size +30M -delete
