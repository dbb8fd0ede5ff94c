#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/c4; O=gpurun_out/c4
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 2400 python -m pytest tests -m gpu -q --durations=8 -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
timeout 600 python scripts/phase_times.py > $O/phase_times.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-events > $GRAFT_REPO_ROOT/$O/bench_prof.json 2> $GRAFT_REPO_ROOT/$O/bench_prof.err
cd $GRAFT_REPO_ROOT
ls -la $O/prof/* | head; 
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/c4/prof/**/*kernel_stats.csv", recursive=True)
print(f)
if f:
    rows = list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("total ms", tot/1e6, "launches", sum(int(r["Calls"]) for r in rows))
    for r in rows[:45]:
        print(f'{float(r["TotalDurationNs"])/1e6:9.3f} ms {int(r["Calls"]):6d} {float(r["AverageNs"])/1e3:9.1f} us  {r["Name"][:90]}')
    at = [r for r in rows if "at::native" in r["Name"] or "rocclr" in r["Name"]]
    print("ATen/rocclr:", sum(int(r["Calls"]) for r in at), "launches", sum(float(r["TotalDurationNs"]) for r in at)/1e6, "ms")
    for r in at: print("   ", r["Calls"], float(r["AverageNs"])/1e3, r["Name"][:110])
PY
tail -3 $O/pytest_gpu.txt; cat $O/phase_times.txt | tail -14
python - <<'PY'
import json
d=json.load(open("gpurun_out/c4/bench.json")); print(d["value"], d["ms_per_step"], d["losses"], d["host_enqueue_ms"], d["step_issued_frac"])
PY
