#!/bin/bash
mkdir -p gpurun_out/c18
timeout 600 python -m pytest tests/test_bf16_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/c18/pytest.txt
cat gpurun_out/c18/pytest.txt
timeout 300 python scripts/bf16_time.py 2>&1 | tee gpurun_out/c18/time.txt
