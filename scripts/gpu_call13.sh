#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/c13; O=gpurun_out/c13
timeout 300 python scripts/s2_time.py pesr_amd/libpesr_hip.so exp/libs2_48.so 2>&1 | grep -v amdgpu.ids > $O/s2_time.txt
cat $O/s2_time.txt
