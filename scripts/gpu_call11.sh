#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/c11; O=gpurun_out/c11
for lib in pesr_amd/libpesr_hip.so exp/libx4noprio.so exp/libx4prio3.so exp/libx4prioa.so pesr_amd/libpesr_hip.so; do
  echo "== $lib" >> $O/variants.txt
  PESR_HIP_LIB=$PWD/$lib timeout 300 python scripts/wgrad4_time.py 2>&1 | grep "32x32x2 (main" >> $O/variants.txt
done
timeout 300 python scripts/wino4_ab.py fwd pesr_amd/libpesr_hip.so exp/libw4prio.so exp/libw4prio0.so >> $O/variants.txt 2>&1
timeout 300 python scripts/wino4_ab.py skip pesr_amd/libpesr_hip.so exp/libw4prio.so exp/libw4prio0.so >> $O/variants.txt 2>&1
cat $O/variants.txt
