#!/bin/bash
# round-3 first GPU call: new tests, bench fields, forced-DP eager vs graph, host profile, CU contention
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/c1; O=gpurun_out/c1
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_dp_gpu.py tests/test_model_gpu.py -m gpu -x -q -k "dp or rank or hipgraph or captured or replay" -s > $O/pytest_new.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_new.txt
timeout 600 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
PESR_FORCE_DP=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_forcedp.json 2> $O/bench_forcedp.err
PESR_FORCE_DP=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --hip-graph > $O/bench_forcedp_graph.json 2> $O/bench_forcedp_graph.err
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --hip-graph > $O/bench_graph.json 2> $O/bench_graph.err
timeout 600 python scripts/host_profile.py > $O/host_profile.txt 2>&1
PESR_FORCE_DP=1 timeout 600 python scripts/host_profile.py > $O/host_profile_dp.txt 2>&1
timeout 600 python scripts/cu_contention.py > $O/cu_contention.txt 2>&1
tail -3 $O/pytest_new.txt; cat $O/bench.json | head -c 1500; echo; tail -5 $O/cu_contention.txt
