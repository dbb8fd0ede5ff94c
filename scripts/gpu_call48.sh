#!/bin/bash
mkdir -p gpurun_out/c48
timeout 900 python -m pytest tests/test_bf16_gpu.py -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/c48/pytest_bf16.txt
timeout 300 python scripts/bf16_s2_time.py 2>&1 | tail -5 | tee gpurun_out/c48/bf16_s2_time.txt
timeout 600 python bench.py --precision bf16 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/c48/bench_bf16.json
python3 -c "import json; j=json.load(open('gpurun_out/c48/bench_bf16.json')); print(j['value'], j['ms_per_step'], j['parity_check'])"
