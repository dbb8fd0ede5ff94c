#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/c10; O=gpurun_out/c10
for lib in pesr_amd/libpesr_hip.so exp/libw4x_nofence.so exp/libw4x_prio.so exp/libw4x_stag.so exp/libw4x_stagprio.so exp/libw4x_stagnof.so pesr_amd/libpesr_hip.so; do
  echo "== $lib" >> $O/wgrad_variants.txt
  PESR_HIP_LIB=$PWD/$lib timeout 300 python scripts/wgrad4_time.py 2>&1 | grep "32x32x2\|16x16x4 - " >> $O/wgrad_variants.txt
done
cat $O/wgrad_variants.txt
