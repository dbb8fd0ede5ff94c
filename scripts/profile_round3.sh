# Round-3 profile set (run on the GPU box from the repo root; scripts/README.md): bench lines (GAN eager = the headline, hipGraph
# replay, forced data-parallel eager / captured, pretrain, config 5), rocprofv3 --stats, a single-stream kernel trace condensed per
# (kernel, grid), four separate --pmc passes over the G-body-shape kernels, three over the HBM-bound kernels, phase / layer times.
export TMPDIR=/tmp; export HSA_ENABLE_IPC_MODE_LEGACY=0; R=$PWD; O=$R/gpurun_out/r03; mkdir -p $O; cd /tmp
timeout 900 python3 $R/bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench.json
timeout 300 python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --hip-graph 2>/dev/null | tail -1 > $O/bench_hip_graph.json
PESR_FORCE_DP=1 timeout 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29521 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_forced_dp.json
PESR_FORCE_DP=1 timeout 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29522 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --hip-graph 2>/dev/null | tail -1 > $O/bench_forced_dp_hip_graph.json
timeout 300 python3 $R/bench.py --workload pretrain --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_pretrain.json
timeout 300 python3 $R/bench.py --workload infer512 --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/bench_infer512.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/single -o run -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-events > $O/single.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o run -- python3 $R/scripts/profile_w4.py > $O/pmc1.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o run -- python3 $R/scripts/profile_w4.py > $O/pmc2.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_sq -o run -- python3 $R/scripts/profile_w4.py > $O/pmc3.log 2>&1
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_grbm -o run -- python3 $R/scripts/profile_w4.py > $O/pmc4.log 2>&1
for c in FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/hbm_$c -o run -- python3 $R/scripts/hbm_kernels_pmc.py > $O/hbm_$c.log 2>&1
done
cd $R
timeout 300 python3 scripts/phase_times.py 2>&1 | grep -v amdgpu.ids > $O/phase_times.txt
timeout 600 python3 scripts/layer_times.py 2>&1 | grep -v amdgpu.ids > $O/layer_times.txt
timeout 300 python3 scripts/linear_time.py 2>&1 | grep -v amdgpu.ids > $O/linear_time.txt
timeout 300 python3 scripts/rgb_layer_time.py 2>&1 | grep -v amdgpu.ids > $O/rgb_layer_time.txt
timeout 300 python3 scripts/host_profile.py 2>&1 | grep -v amdgpu.ids | head -40 > $O/host_profile.txt
python3 scripts/summarize_profiles.py trace $(find $O/single -name "*kernel_trace.csv") 6 $O/kernel_trace_by_grid.csv 2   # 2 warm-up + 3 timed + 3 host-enqueue steps
python3 scripts/summarize_profiles.py pmc $O/k1_pmc_summary.csv $(find $O/pmc_* -name "*counter_collection.csv")
python3 scripts/summarize_profiles.py pmc $O/hbm_pmc_summary.csv $(find $O/hbm_* -name "*counter_collection.csv")
cp $(find $O/stats -name "*kernel_stats.csv") $O/kernel_stats.csv 2>/dev/null
rm -rf $O/stats $O/single $O/pmc_* $O/hbm_FETCH_SIZE $O/hbm_WRITE_SIZE $O/hbm_GRBM_GUI_ACTIVE
head -30 $O/kernel_trace_by_grid.csv; cut -c1-600 $O/bench.json; echo; cat $O/phase_times.txt; grep -v "at::native" $O/hbm_pmc_summary.csv | head -60
