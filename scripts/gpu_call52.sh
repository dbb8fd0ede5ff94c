#!/bin/bash
export TMPDIR=/tmp; export HSA_ENABLE_IPC_MODE_LEGACY=0; R=$PWD; O=$R/gpurun_out/c52; mkdir -p $O
timeout 900 python -m pytest tests/test_bf16_gpu.py -x -q -m gpu 2>&1 | tail -5 | tee $O/pytest_bf16.txt
timeout 300 python scripts/bf16_s2_time.py 2>&1 | grep "s2 " | tee $O/bf16_s2_time.txt
cd /tmp
for rep in 1 2; do
for v in 0 1; do
for g in "" "--hip-graph"; do
PESR_BF16_NO_S2=$v timeout 300 python3 $R/bench.py --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline $g 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('no_s2=$v graph=$g', j['value'], j['ms_per_step'], j.get('host_enqueue_ms'))" | tee -a $O/ab.txt
done; done; done
