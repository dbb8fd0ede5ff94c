#!/bin/bash
mkdir -p gpurun_out/c43
for what in fwd skip; do
  echo "--- $what"
  timeout 300 python scripts/wino4_ab.py $what pesr_amd/libpesr_hip.so exp/libw4nt.so 2>&1 | grep -v amdgpu
done | tee gpurun_out/c43/ab.txt
for lib in pesr_amd/libpesr_hip.so exp/libw4nt.so pesr_amd/libpesr_hip.so exp/libw4nt.so; do
  PESR_HIP_LIB=$lib timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('$lib', j['value'], j['ms_per_step'], j['roofline']['avg_launch_us'])"
done | tee gpurun_out/c43/bench.txt
