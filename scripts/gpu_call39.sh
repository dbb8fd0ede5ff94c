#!/bin/bash
mkdir -p gpurun_out/c39
for lib in pesr_amd/libpesr_hip.so exp/libx4same3.so exp/libx4same4.so exp/libx4same5.so pesr_amd/libpesr_hip.so; do
  echo "== $lib"
  PESR_HIP_LIB=$lib timeout 300 python scripts/wgrad4_time.py 2>&1 | grep "32x32x2"
done | tee gpurun_out/c39/time.txt
