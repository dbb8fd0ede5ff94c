#!/bin/bash
mkdir -p gpurun_out/c46
for q in 3 6 9 12; do
echo "PESR_S2Q=$q" | tee -a gpurun_out/c46/s2_dgrad.txt
PESR_S2Q=$q timeout 600 python scripts/s2_dgrad_time.py pesr_amd/libpesr_hip.so exp/libs2old.so 2>&1 | grep dgrad | tee -a gpurun_out/c46/s2_dgrad.txt
done
