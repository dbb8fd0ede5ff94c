#!/bin/bash
# soak: the kernel-level suites repeated (rare races in the counted-wait pipelines show up as rare failures)
mkdir -p gpurun_out/c61
for i in 1 2 3 4 5; do timeout 600 python -m pytest tests/test_bf16_gpu.py tests/test_conv_gpu.py tests/test_ops_gpu.py -x -q -m gpu 2>&1 | tail -1 | tee -a gpurun_out/c61/soak.txt; done
