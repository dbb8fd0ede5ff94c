#!/bin/bash
# same-box A/B of the bf16 row with / without the stride-2 form of the bf16 forward kernel (eager and hipGraph), interleaved twice
export TMPDIR=/tmp; export HSA_ENABLE_IPC_MODE_LEGACY=0; R=$PWD; O=$R/gpurun_out/c51; mkdir -p $O; cd /tmp
for rep in 1 2; do
for v in 0 1; do
for g in "" "--hip-graph"; do
PESR_BF16_NO_S2=$v timeout 300 python3 $R/bench.py --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline $g 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('no_s2=$v graph=$g', j['value'], j['ms_per_step'], j.get('host_enqueue_ms'))" | tee -a $O/ab.txt
done; done; done
