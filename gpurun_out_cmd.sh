mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -q --timeout 400 -p no:cacheprovider 2>&1 | tail -60) > gpurun_out/pytest_gpu.log 2>&1
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5) > gpurun_out/smoke.log 2>&1
(timeout 600 python bench.py --steps 10 --warmup 2 2>&1 | tail -5) > gpurun_out/bench_n1.log 2>&1
cd /tmp && export TMPDIR=/tmp
(timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r01 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 1 2>&1 | tail -5) > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
ls -R gpurun_out | head -40
tail -5 gpurun_out/pytest_gpu.log; cat gpurun_out/smoke.log; cat gpurun_out/bench_n1.log
rocm-smi --showmeminfo vram | head -8; nproc
