#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/c7; O=gpurun_out/c7
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests/test_conv_gpu.py tests/test_ops_gpu.py tests/test_model_gpu.py -m gpu -q -k "other_kernel or 5x5 or linear or hipgraph or captured or replay" > $O/pytest_sel.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_sel.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_eager.json 2> $O/bench_eager.err
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --hip-graph > $O/bench_graph_events.json 2> $O/bench_graph_events.err
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --hip-graph --no-graph-events > $O/bench_graph_noev.json 2> $O/bench_graph_noev.err
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --hip-graph --event-every 1 > $O/bench_graph_events_all.json 2> $O/bench_graph_events_all.err
tail -3 $O/pytest_sel.txt
python - <<'PY'
import json
for f in ("bench_eager","bench_graph_events","bench_graph_noev","bench_graph_events_all"):
    try:
        d=json.load(open(f"gpurun_out/c7/{f}.json"))
        r=d.get("roofline",{}); rk=d.get("roofline_kernels",[{},{}])
        print(f, d["value"], d["ms_per_step"], "fwd", r.get("avg_launch_us"), r.get("launches_timed"), "dgrad", rk[0].get("avg_launch_us"), "wgrad", rk[1].get("avg_launch_us") if len(rk)>1 else None)
    except Exception as e:
        print(f, "FAILED", e); print(open(f"gpurun_out/c7/{f}.err").read()[-1500:])
PY
