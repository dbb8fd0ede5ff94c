#!/bin/bash
# final record of the round: full GPU suite, smoke, the bench line, and the bf16 step's per-kernel tables on the final code
export TMPDIR=/tmp; export HSA_ENABLE_IPC_MODE_LEGACY=0; R=$PWD; O=$R/gpurun_out/c47; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
cd /tmp
timeout 600 python3 $R/bench.py 2>/dev/null | tail -1 > $O/bench.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 $R/bench.py --precision bf16 --steps 5 --warmup 2 --no-cpu-baseline > $O/stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/single -o run -- python3 $R/bench.py --precision bf16 --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-events > $O/single.log 2>&1
cd $R
python3 scripts/summarize_profiles.py trace $(find $O/single -name "*kernel_trace.csv") 6 $O/bf16_kernel_trace_by_grid.csv 2
cp $(find $O/stats -name "*kernel_stats.csv") $O/bf16_kernel_stats.csv 2>/dev/null
rm -rf $O/stats $O/single
python3 -c "import json; j=json.load(open('$O/bench.json')); print(j['value'], j['ms_per_step'], j['parity_check']['max_rel_loss_err'], j['roofline']['frac'])"
head -12 $O/bf16_kernel_trace_by_grid.csv | cut -c1-160
