#!/bin/bash
mkdir -p gpurun_out/c30
timeout 600 python -m pytest tests/test_bf16_gpu.py -x -q -k "entrypoint" 2>&1 | tail -8
timeout 300 python bench.py --workload infer512 --steps 3 --warmup 1 --precision bf16 2>/dev/null | tail -1 | tee gpurun_out/c30/bench_bf16_infer512.json
timeout 300 python bench.py --workload infer512 --steps 3 --warmup 1 2>/dev/null | tail -1 | tee gpurun_out/c30/bench_infer512.json
