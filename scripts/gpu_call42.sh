#!/bin/bash
mkdir -p gpurun_out/c42
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_ops_gpu.py -x -q 2>&1 | tail -3
for lib in pesr_amd/libpesr_hip.so exp/libprev.so pesr_amd/libpesr_hip.so exp/libprev.so; do
  echo "== $lib"
  PESR_HIP_LIB=$lib timeout 300 python scripts/phase_times.py 2>&1 | grep -E "D fwd \(sr\)|D bwd|D dgrad|total"
  PESR_HIP_LIB=$lib timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('bench', j['value'], j['ms_per_step'])"
done | tee gpurun_out/c42/ab.txt
