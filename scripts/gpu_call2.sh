#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/c2; O=gpurun_out/c2
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 2400 python -m pytest tests -m gpu -q -x --durations=15 > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
timeout 600 python bench.py --steps 20 --warmup 5 --lr 5e-5 --no-cpu-baseline > $O/bench_lr5e-5.json 2> $O/bench_lr5e-5.err
tail -4 $O/pytest_gpu.txt; head -c 2500 $O/bench.json; echo; python - <<'PY'
import json
for f in ("bench","bench_lr5e-5"):
    d=json.load(open(f"gpurun_out/c2/{f}.json")); print(f, d["value"], d["ms_per_step"], d["losses"], d["host_enqueue_ms"])
PY
