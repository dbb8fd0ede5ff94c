#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/c15; O=gpurun_out/c15
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 2400 python -m pytest tests -m gpu -q --durations=5 > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
timeout 300 python scripts/aten_residue.py 2>&1 | grep -v "amdgpu.ids" | tail -22 > $O/aten_residue.txt
grep -n "passed\|failed\|FAILED\|^E  " $O/pytest_gpu.txt | head; cat $O/aten_residue.txt
python - <<'PY'
import json
d=json.load(open("gpurun_out/c15/bench.json")); print(d["value"], d["ms_per_step"], d["losses"], d["host_enqueue_ms"], d["parity_check"]["max_rel_loss_err"])
PY
