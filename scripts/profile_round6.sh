# Round-6 profile set (run on the GPU box from the repo root, through scripts/gpu_job.sh): the headline bench line (with the side
# measurements of configs 2 and 5), the forced one-rank data-parallel line, hipGraph replay, the optional precision rows, rocprofv3 --stats,
# single-stream trace per (kernel, grid) cut at optimizer steps, four separate --pmc passes over the G-body kernels, three over the HBM-bound
# kernels, the TCC request counters of the Linear kernels, phase / layer / Linear / RGB-layer / BatchNorm-fusion times.
# Outputs -> gpurun_out/r06/ (copy what is to be judged into profiles/).
export TAG=r06
bash scripts/gpu_job.sh bench bench --steps 20 --warmup 5
bash scripts/gpu_job.sh dp1 bench_forced_dp --steps 20 --warmup 5 --no-cpu-baseline --no-side
bash scripts/gpu_job.sh bench bench_hip_graph --steps 20 --warmup 5 --no-cpu-baseline --no-side --hip-graph
bash scripts/gpu_job.sh bench bench_lr5e-5 --steps 20 --warmup 5 --no-cpu-baseline --no-side --lr 5e-5
bash scripts/gpu_job.sh bench bench_split_bf16 --steps 20 --warmup 5 --precision split-bf16
bash scripts/gpu_job.sh bench bench_bf16 --steps 20 --warmup 5 --precision bf16
PESR_BN_FUSE=0 bash scripts/gpu_job.sh bench bench_bn_unfused --steps 20 --warmup 5 --no-cpu-baseline --no-side
bash scripts/gpu_job.sh stats
bash scripts/gpu_job.sh trace
bash scripts/gpu_job.sh pmc k1 scripts/profile_w4.py
bash scripts/gpu_job.sh hbm hbm scripts/hbm_kernels_pmc.py
bash scripts/gpu_job.sh tcc linear scripts/linear_pmc.py
bash scripts/gpu_job.sh py phase_times scripts/phase_times.py
bash scripts/gpu_job.sh py layer_times scripts/layer_times.py
bash scripts/gpu_job.sh py linear_time scripts/linear_time.py
bash scripts/gpu_job.sh py rgb_layer_time scripts/rgb_layer_time.py
bash scripts/gpu_job.sh py bn_fuse_time scripts/bn_fuse_time.py
