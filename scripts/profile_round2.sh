# Round-2 profile set (run on the GPU box from the repo root): bench line, rocprofv3 --stats, single-stream kernel trace
# condensed per (kernel, grid), and four separate --pmc passes over the G-body-shape kernels (scripts/profile_w4.py).
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r02; mkdir -p $O; cd /tmp
timeout 600 python3 $R/bench.py --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/bench.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/single -o run -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-events > $O/single.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o run -- python3 $R/scripts/profile_w4.py > $O/pmc1.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o run -- python3 $R/scripts/profile_w4.py > $O/pmc2.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_sq -o run -- python3 $R/scripts/profile_w4.py > $O/pmc3.log 2>&1
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_grbm -o run -- python3 $R/scripts/profile_w4.py > $O/pmc4.log 2>&1
cd $R
python3 scripts/summarize_profiles.py trace $(find $O/single -name "*kernel_trace.csv") 3 $O/kernel_trace_by_grid.csv 2
python3 scripts/summarize_profiles.py pmc $O/k1_pmc_summary.csv $(find $O/pmc_* -name "*counter_collection.csv")
cp $(find $O/stats -name "*kernel_stats.csv") $O/kernel_stats.csv 2>/dev/null
head -45 $O/kernel_trace_by_grid.csv; cut -c1-300 $O/bench.json
