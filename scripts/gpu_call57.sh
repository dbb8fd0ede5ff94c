#!/bin/bash
export TMPDIR=/tmp; export HSA_ENABLE_IPC_MODE_LEGACY=0; R=$PWD; O=$R/gpurun_out/c57; mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python scripts/bn_time.py 2>&1 | tail -12 | tee $O/bn_time.txt
