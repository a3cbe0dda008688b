mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -q --timeout 400 -p no:cacheprovider 2>&1 | tail -30) > gpurun_out/pytest_gpu.log 2>&1
(timeout 300 python bench.py --batch 1 --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -3) > gpurun_out/bench_b1.log 2>&1
(timeout 300 python tools/profile_layers.py --batch 1 --steps 5 2>&1 | tail -45) > gpurun_out/layers_b1.log 2>&1
tail -5 gpurun_out/pytest_gpu.log; cat gpurun_out/bench_b1.log; tail -3 gpurun_out/layers_b1.log
