#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/c9; O=gpurun_out/c9
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python scripts/wgrad4_time.py > $O/wgrad4_time.txt 2>&1
echo "== interleaved accumulators" >> $O/wgrad4_time.txt; PESR_HIP_LIB=$PWD/exp/libw4x_il.so timeout 300 python scripts/wgrad4_time.py 2>&1 | grep "32x32x2" >> $O/wgrad4_time.txt
timeout 2400 python -m pytest tests -m gpu -q --durations=5 > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
PESR_HIP_LIB=$PWD/exp/libw4x_il.so timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_il.json 2> $O/bench_il.err
cat $O/wgrad4_time.txt; grep -n "passed\|failed\|FAILED\|^E  " $O/pytest_gpu.txt | head
python - <<'PY'
import json
for f in ("bench","bench_il"):
    try:
        d=json.load(open(f"gpurun_out/c9/{f}.json")); rk=d.get("roofline_kernels",[{},{}])
        print(f, d["value"], d["ms_per_step"], "fwd", d["roofline"]["avg_launch_us"], "wgrad", rk[1].get("avg_launch_us"), rk[1].get("frac"), d.get("parity_check",{}).get("max_rel_loss_err"))
    except Exception as e: print(f, "FAILED", e); print(open(f"gpurun_out/c9/{f}.err").read()[-800:])
PY
