# Round-5 profile set (run on the GPU box from the repo root, through scripts/gpu_job.sh): the headline bench line (with the side
# measurements of configs 2 and 5), the forced one-rank data-parallel line (schedule calibration incl. the captured step), hipGraph
# replay, rocprofv3 --stats, single-stream trace per (kernel, grid), four separate --pmc passes over the G-body kernels, three over
# the HBM-bound kernels, phase / layer / Linear / RGB-layer times.  Outputs -> gpurun_out/r05/ (copy what is to be judged into profiles/).
export TAG=r05
bash scripts/gpu_job.sh bench bench --steps 20 --warmup 5
bash scripts/gpu_job.sh dp1 bench_forced_dp --steps 20 --warmup 5 --no-cpu-baseline --no-side
bash scripts/gpu_job.sh bench bench_hip_graph --steps 20 --warmup 5 --no-cpu-baseline --no-side --hip-graph
bash scripts/gpu_job.sh bench bench_lr5e-5 --steps 20 --warmup 5 --no-cpu-baseline --no-side --lr 5e-5
bash scripts/gpu_job.sh bench bench_split_bf16 --steps 20 --warmup 5 --precision split-bf16
bash scripts/gpu_job.sh bench bench_bf16 --steps 20 --warmup 5 --precision bf16
bash scripts/gpu_job.sh stats
bash scripts/gpu_job.sh trace
bash scripts/gpu_job.sh pmc k1 scripts/profile_w4.py
bash scripts/gpu_job.sh hbm hbm scripts/hbm_kernels_pmc.py
bash scripts/gpu_job.sh py phase_times scripts/phase_times.py
bash scripts/gpu_job.sh py layer_times scripts/layer_times.py
bash scripts/gpu_job.sh py linear_time scripts/linear_time.py
bash scripts/gpu_job.sh py rgb_layer_time scripts/rgb_layer_time.py
bash scripts/gpu_job.sh py split_bf16_kernel_times scripts/bf16x3_time.py
# (profiles/r05_wgrad_phases.txt: in-kernel stamps of the 12-wave weight-gradient kernel, a -DX4_STAMPS build of scripts/diag/conv3x3_wgrad_wino4_diag.hip:
#  bash scripts/build_variant.sh x4st conv3x3_wgrad_wino4_diag.hip -DX4_STAMPS; bash scripts/gpu_job.sh py wgrad_phases scripts/x4_phases.py exp/libx4st.so)
