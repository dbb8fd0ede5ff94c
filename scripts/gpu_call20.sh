#!/bin/bash
mkdir -p gpurun_out/c20
timeout 900 python -m pytest tests/test_bf16_gpu.py -x -q 2>&1 | tail -25 > gpurun_out/c20/pytest.txt
cat gpurun_out/c20/pytest.txt
timeout 600 python bench.py --precision bf16 --steps 10 --warmup 3 > gpurun_out/c20/bench_bf16.json 2> gpurun_out/c20/bench_bf16.err
tail -3 gpurun_out/c20/bench_bf16.err; cat gpurun_out/c20/bench_bf16.json | head -c 3000
