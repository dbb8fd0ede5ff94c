#!/bin/bash
export TMPDIR=/tmp; export HSA_ENABLE_IPC_MODE_LEGACY=0; R=$PWD; O=$R/gpurun_out/c55; mkdir -p $O; cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/single -o run -- python3 $R/bench.py --precision bf16 --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-events > $O/single.log 2>&1
cd $R
python3 scripts/summarize_profiles.py trace $(find $O/single -name "*kernel_trace.csv") 6 $O/bf16_kernel_trace_by_grid.csv 2
rm -rf $O/single
head -45 $O/bf16_kernel_trace_by_grid.csv | cut -c1-140
