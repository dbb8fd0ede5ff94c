#!/bin/bash
# refresh the round-3 profile set on the final fp32 code + the bf16 row's variants (hipGraph replay, forced data-parallel)
export TMPDIR=/tmp; export HSA_ENABLE_IPC_MODE_LEGACY=0; R=$PWD
bash scripts/profile_round3.sh > gpurun_out/r03_profile.log 2>&1
O=$R/gpurun_out/r03
cd /tmp
timeout 600 python3 $R/bench.py --precision bf16 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_bf16.json
timeout 300 python3 $R/bench.py --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline --hip-graph 2>/dev/null | tail -1 > $O/bench_bf16_hip_graph.json
PESR_FORCE_DP=1 timeout 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 $R/bench.py --gpus 1 --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_bf16_forced_dp.json
timeout 300 python3 $R/bench.py --precision bf16 --workload pretrain --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_bf16_pretrain.json
cd $R
for f in bench bench_hip_graph bench_forced_dp bench_forced_dp_hip_graph bench_pretrain bench_infer512 bench_bf16 bench_bf16_hip_graph bench_bf16_forced_dp bench_bf16_pretrain; do
  python3 -c "import json,sys; j=json.load(open('$O/$f.json')); print('$f', j['value'], j.get('ms_per_step'), j.get('parity_check',{}).get('max_rel_loss_err'), j.get('comm_exposed_ms'))" 2>&1 | tail -1
done
head -12 $O/kernel_trace_by_grid.csv
