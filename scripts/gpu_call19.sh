#!/bin/bash
mkdir -p gpurun_out/c19
for lib in pesr_amd/libpesr_hip.so exp/libwb256.so exp/libwb384.so; do
  echo "== $lib"
  PESR_HIP_LIB=$lib timeout 300 python scripts/bf16_time.py 2>&1 | grep wgrad
done | tee gpurun_out/c19/time.txt
