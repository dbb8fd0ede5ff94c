#!/bin/bash
# bf16 mode, first contact: parity of the forward / input-gradient kernel, then its time against the fp32 F(4,3) kernel
mkdir -p gpurun_out/c17
timeout 600 python -m pytest tests/test_bf16_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/c17/pytest.txt
cat gpurun_out/c17/pytest.txt
timeout 300 python scripts/bf16_time.py 2>&1 | tee gpurun_out/c17/time.txt
