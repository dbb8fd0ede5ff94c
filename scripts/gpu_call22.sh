#!/bin/bash
mkdir -p gpurun_out/c22
for lib in pesr_amd/libpesr_hip.so exp/libb16fakew.so exp/libb16fakex.so exp/libb16fakewx.so exp/libb16st8.so exp/libb16st4.so pesr_amd/libpesr_hip.so; do
  echo "== $lib"
  PESR_HIP_LIB=$lib timeout 300 python scripts/bf16_time.py 2>&1 | grep -E "^fwd 16x48x48 256->256|^fwd 16x96x96 256->1024|dgrad" | sed 's/fp32 F(4,3):[^|]*|//'
done | tee gpurun_out/c22/time.txt
