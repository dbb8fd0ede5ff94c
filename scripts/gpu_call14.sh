#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/c14; O=gpurun_out/c14
timeout 300 python scripts/s2_dgrad_time.py exp/libno48.so pesr_amd/libpesr_hip.so 2>&1 | grep -v amdgpu.ids > $O/s2_time.txt
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_ops_gpu.py -m gpu -q -x > $O/pytest_conv.txt 2>&1; echo "rc=$?" >> $O/pytest_conv.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
cat $O/s2_time.txt; tail -3 $O/pytest_conv.txt
python - <<'PY'
import json
d=json.load(open("gpurun_out/c14/bench.json")); print(d["value"], d["ms_per_step"])
PY
