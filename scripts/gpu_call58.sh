#!/bin/bash
# final record of the round: full GPU suite, smoke, the whole round-3 profile set (fp32), the bf16 row's variants and per-kernel tables
export TMPDIR=/tmp; export HSA_ENABLE_IPC_MODE_LEGACY=0; R=$PWD; O=$R/gpurun_out/r03; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash scripts/profile_round3.sh > gpurun_out/r03_profile.log 2>&1
cd /tmp
timeout 900 python3 $R/bench.py --precision bf16 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_bf16.json
timeout 300 python3 $R/bench.py --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline --hip-graph 2>/dev/null | tail -1 > $O/bench_bf16_hip_graph.json
PESR_FORCE_DP=1 timeout 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 $R/bench.py --gpus 1 --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_bf16_forced_dp.json
timeout 300 python3 $R/bench.py --precision bf16 --workload pretrain --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_bf16_pretrain.json
timeout 300 python3 $R/bench.py --precision bf16 --workload infer512 --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/bench_bf16_infer512.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bstats -o run -- python3 $R/bench.py --precision bf16 --steps 5 --warmup 2 --no-cpu-baseline > $O/bstats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/bsingle -o run -- python3 $R/bench.py --precision bf16 --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-events > $O/bsingle.log 2>&1
cd $R
python3 scripts/summarize_profiles.py trace $(find $O/bsingle -name "*kernel_trace.csv") 6 $O/bf16_kernel_trace_by_grid.csv 2
cp $(find $O/bstats -name "*kernel_stats.csv") $O/bf16_kernel_stats.csv 2>/dev/null
rm -rf $O/bstats $O/bsingle
for f in bench bench_hip_graph bench_forced_dp bench_forced_dp_hip_graph bench_pretrain bench_infer512 bench_bf16 bench_bf16_hip_graph bench_bf16_forced_dp bench_bf16_pretrain bench_bf16_infer512; do
  python3 -c "import json,sys; j=json.load(open('$O/$f.json')); print('$f', j['value'], j.get('ms_per_step'), j.get('parity_check',{}).get('max_rel_loss_err'), j.get('parity_check',{}).get('ok'), j.get('comm_exposed_ms'))" 2>&1 | tail -1
done
python3 -c "import json; j=json.load(open('$O/bench.json')); print(j['roofline']['frac'], [ (k['kernel'][:40], k['frac']) for k in j.get('roofline_kernels',[])])"
cat $O/phase_times.txt
