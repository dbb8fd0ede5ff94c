#!/bin/bash
export TMPDIR=/tmp; export HSA_ENABLE_IPC_MODE_LEGACY=0; R=$PWD; O=$R/gpurun_out/c54; mkdir -p $O
timeout 900 python -m pytest tests/test_bf16_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest_bf16.txt
cd /tmp
for rep in 1 2; do
for v in 64 128; do
for g in "" "--hip-graph"; do
PESR_BF16_MIN_WGS=$v timeout 300 python3 $R/bench.py --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline $g 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('min_wgs=$v graph=$g', j['value'], j['ms_per_step'], j.get('host_enqueue_ms'))" | tee -a $O/ab.txt
done; done; done
timeout 900 python3 $R/bench.py --precision bf16 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_bf16.json
python3 -c "import json; j=json.load(open('$O/bench_bf16.json')); print(j['value'], j['ms_per_step'], j['parity_check']['max_rel_loss_err'], j['parity_check']['tol'], j['parity_check']['ok'], j['parity_check'].get('oracle_one_ulp_noise_floor'))"
