# debugging aid (round 5): bench.py with four ranks time-sharing cuda:0 (test hook); stacks of hung ranks after 90 s
cd /root/repo
export PESR_DP_BACKEND=gloo PESR_DP_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0 PESR_DUMP_STACKS_AFTER=90
A="bench.py --gpus 4 --steps 2 --warmup 2 --batch 4 --patch_size 24 --num_channels 64 --num_blocks 2 --calib-steps 1"
run() { n=$1; shift; ( "$@" timeout 150 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node=4 --master-addr 127.0.0.1 --master-port $((29540 + RANDOM % 50)) $A $EXTRA > gpurun_out/run4_$n.out 2> gpurun_out/run4_$n.err ); echo "$n rc=$?"; grep -c "Timeout (0:01:30)" gpurun_out/run4_$n.err; grep -h "comm.py\|step.py\|bench.py\", line" gpurun_out/run4_$n.err | sort | uniq -c | head -8; tail -c 300 gpurun_out/run4_$n.out; echo; }
EXTRA="" run a env
EXTRA="" run b env
EXTRA="" run c env
