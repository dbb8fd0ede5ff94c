#!/bin/bash
mkdir -p gpurun_out/c31
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee gpurun_out/c31/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['parity_check']['max_rel_loss_err'], j['roofline']['frac'], j['cpu_baseline']['value'])"
