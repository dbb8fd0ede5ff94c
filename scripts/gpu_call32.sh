#!/bin/bash
mkdir -p gpurun_out/c32
timeout 900 python -m pytest tests/test_bf16_gpu.py -x -q 2>&1 | tail -5
timeout 600 python bench.py --precision bf16 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/c32/bench_bf16.json
python3 -c "import json; j=json.load(open('gpurun_out/c32/bench_bf16.json')); print(j['value'], j['ms_per_step'], j['parity_check']['max_rel_loss_err'], j['step_flops']['by_kernel_family_tflop'])"
