#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/c8; O=gpurun_out/c8
timeout 900 python -m pytest tests/test_conv_gpu.py -m gpu -q -x -k "wgrad" > $O/pytest_wgrad.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_wgrad.txt
timeout 300 python scripts/wgrad4_time.py > $O/wgrad4_time.txt 2>&1
for v in w4x_s2 w4x_s4; do echo "== $v" >> $O/wgrad4_time.txt; PESR_HIP_LIB=$PWD/exp/lib$v.so timeout 300 python scripts/wgrad4_time.py 2>&1 | grep "32x32x2" >> $O/wgrad4_time.txt; done
PESR_WGRAD_WINO4X=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_w4x.json 2> $O/bench_w4x.err
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_base.json 2> $O/bench_base.err
tail -5 $O/pytest_wgrad.txt; cat $O/wgrad4_time.txt
python - <<'PY'
import json
for f in ("bench_base","bench_w4x"):
    try:
        d=json.load(open(f"gpurun_out/c8/{f}.json")); rk=d.get("roofline_kernels",[{},{}])
        print(f, d["value"], d["ms_per_step"], "wgrad", rk[1].get("avg_launch_us"), rk[1].get("kernel","")[:40], d["parity_check"] if "parity_check" in d else "")
    except Exception as e: print(f, "FAILED", e); print(open(f"gpurun_out/c8/{f}.err").read()[-800:])
PY
