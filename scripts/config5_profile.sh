# BASELINE config 5 (G forward on 4 x 512x512 LR tiles): bench line + rocprofv3 kernel trace, condensed to the HBM-bound
# kernels' GB/s (profiles/r02_config5_hbm_kernels.csv).  Run on the GPU box from the repo root.
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r02c5; mkdir -p $O; cd /tmp
timeout 600 python3 $R/bench.py --workload infer512 --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/bench_infer512.json
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o run -- python3 $R/bench.py --workload infer512 --steps 2 --warmup 1 > $O/trace.log 2>&1
cd $R
python3 scripts/summarize_profiles.py trace $(find $O/trace -name "*kernel_trace.csv") 2 $O/kernel_trace_by_grid.csv 1
python3 scripts/config5_table.py $O/kernel_trace_by_grid.csv $O/config5_hbm_kernels.csv
cat $O/config5_hbm_kernels.csv; cut -c1-400 $O/bench_infer512.json
