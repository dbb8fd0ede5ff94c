#!/bin/bash
mkdir -p gpurun_out/c44
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_bf16_gpu.py tests/test_fullsize_gpu.py -x -q 2>&1 | tail -3
for lib in pesr_amd/libpesr_hip.so exp/libprev.so pesr_amd/libpesr_hip.so; do
  echo "== $lib"
  PESR_HIP_LIB=$lib timeout 300 python scripts/wgrad4_time.py 2>&1 | grep "32x32x2 (main\|direct"
  PESR_HIP_LIB=$lib timeout 300 python scripts/bf16_time.py 2>&1 | grep "^wgrad 16x48x48 256->256" | sed 's/fp32 F(4,3):[^|]*|//'
done | tee gpurun_out/c44/time.txt
for lib in pesr_amd/libpesr_hip.so exp/libprev.so pesr_amd/libpesr_hip.so exp/libprev.so; do
  PESR_HIP_LIB=$lib timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('$lib', j['value'], j['ms_per_step'])"
done | tee gpurun_out/c44/bench.txt
