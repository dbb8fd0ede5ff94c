#!/bin/bash
# full GPU suite after the bf16 mode went in + where the bf16 step's time goes (rocprofv3 --stats, single-stream trace per grid)
export TMPDIR=/tmp; export HSA_ENABLE_IPC_MODE_LEGACY=0; R=$PWD; O=$R/gpurun_out/c21; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 $R/bench.py --precision bf16 --steps 5 --warmup 2 --no-cpu-baseline > $O/stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/single -o run -- python3 $R/bench.py --precision bf16 --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-events > $O/single.log 2>&1
cd $R
python3 scripts/summarize_profiles.py trace $(find $O/single -name "*kernel_trace.csv") 6 $O/bf16_kernel_trace_by_grid.csv 2
cp $(find $O/stats -name "*kernel_stats.csv") $O/bf16_kernel_stats.csv 2>/dev/null
rm -rf $O/stats $O/single
head -45 $O/bf16_kernel_trace_by_grid.csv
