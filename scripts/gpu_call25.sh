#!/bin/bash
mkdir -p gpurun_out/c25
for lib in pesr_amd/libpesr_hip.so exp/libb16noread.so exp/libb16nomfma.so exp/libb16now.so exp/libb16noreadnow.so; do
  echo "== $lib"
  PESR_HIP_LIB=$lib timeout 300 python scripts/bf16_time.py 2>&1 | grep -E "^fwd 16x48x48 256->256|^fwd 16x96x96 256->1024" | sed 's/fp32 F(4,3):[^|]*|//'
done | tee gpurun_out/c25/time.txt
