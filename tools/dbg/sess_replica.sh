cd /root/repo
timeout 600 python -m pytest tests/test_gpu_replica.py -m gpu -x -q 2>&1 | tail -5
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -3 gpurun_out/bench_default.err; cat gpurun_out/bench_default.json
timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-ref-f32 --layers 2>gpurun_out/bench_long.err | tee gpurun_out/bench_long.json; grep layer gpurun_out/bench_long.err | head -30
timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-ref-f32 --inflight 1 | tee gpurun_out/bench_long_if1.json
