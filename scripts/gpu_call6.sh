#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/c6; O=gpurun_out/c6
for lib in pesr_amd/libpesr_hip.so exp/liblin_one8u8.so exp/liblin_one4u16.so exp/liblin_one8u16.so exp/liblin_one4u8.so exp/liblin_oldu16.so exp/liblin_oldu4.so; do
  echo "== $lib" >> $O/linear_sweep.txt
  PESR_HIP_LIB=$PWD/$lib timeout 300 python scripts/linear_time.py 2>&1 | grep -v amdgpu.ids >> $O/linear_sweep.txt
done
timeout 900 python scripts/layer_times.py > $O/layer_times.txt 2>&1
cat $O/linear_sweep.txt; head -75 $O/layer_times.txt
