#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/c5; O=gpurun_out/c5
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python scripts/linear_time.py > $O/linear_time.txt 2>&1
timeout 2400 python -m pytest tests -m gpu -q --durations=5 > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
cat $O/linear_time.txt; grep -n "passed\|failed\|FAILED\|^E  " $O/pytest_gpu.txt | head -20
python - <<'PY'
import json
d=json.load(open("gpurun_out/c5/bench.json")); print(d["value"], d["ms_per_step"], d["losses"], d["host_enqueue_ms"], d["step_issued_frac"])
PY
