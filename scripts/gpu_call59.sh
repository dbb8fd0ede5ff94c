#!/bin/bash
# Linear input gradient: split-N slices x rows per step (exp/liblin_*.so built by scripts/build_variant.sh with -DLD_NS_TARGET / -DLD_U)
mkdir -p gpurun_out/c59
for l in pesr_amd/libpesr_hip.so exp/liblin_ns8u4.so exp/liblin_ns8u8.so exp/liblin_ns4u8.so exp/liblin_ns4u16.so exp/liblin_ns32u4.so exp/liblin_ns32u2.so exp/liblin_ns16u2.so pesr_amd/libpesr_hip.so; do
echo "== $l" | tee -a gpurun_out/c59/linear_ns.txt
PESR_HIP_LIB=$PWD/$l timeout 120 python scripts/linear_time.py 2>&1 | grep "input gradient" | tee -a gpurun_out/c59/linear_ns.txt
done
