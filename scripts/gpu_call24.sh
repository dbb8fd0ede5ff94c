#!/bin/bash
mkdir -p gpurun_out/c24
timeout 300 python -m pytest tests/test_bf16_gpu.py -x -q -k "kernel or pixel" 2>&1 | tail -3
for lib in pesr_amd/libpesr_hip.so exp/libb16fxd2.so exp/libb16prio.so exp/libb16fxd2prio.so pesr_amd/libpesr_hip.so; do
  echo "== $lib"
  PESR_HIP_LIB=$lib timeout 300 python scripts/bf16_time.py 2>&1 | grep -E "^fwd 16x48x48 256->256|^fwd 16x96x96 256->1024|dgrad.*bf16" | sed 's/fp32 F(4,3):[^|]*|//'
done | tee gpurun_out/c24/time.txt
