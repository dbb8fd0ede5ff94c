#!/bin/bash
mkdir -p gpurun_out/c26
timeout 600 python -m pytest tests/test_bf16_gpu.py -x -q 2>&1 | tail -5
timeout 300 python scripts/bf16_time.py 2>&1 | grep -E "^wgrad" | tee gpurun_out/c26/time.txt
