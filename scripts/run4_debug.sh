# debugging aid (round 5): bench.py with N ranks (default 4) time-sharing cuda:0 (test hook); stacks of hung ranks after 120 s
#   bash scripts/run4_debug.sh [N] [runs]
cd /root/repo
N=${1:-4}; RUNS=${2:-3}
export PESR_DP_BACKEND=gloo PESR_DP_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0 PESR_DUMP_STACKS_AFTER=120
A="bench.py --gpus $N --steps 2 --warmup 2 --batch 4 --patch_size 24 --num_channels 64 --num_blocks 2 --calib-steps 1"
for n in $(seq 1 $RUNS); do
  timeout 240 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node=$N --master-addr 127.0.0.1 --master-port $((29540 + RANDOM % 50)) $A > gpurun_out/runN_$n.out 2> gpurun_out/runN_$n.err
  echo "run $n rc=$?"; grep -c "Timeout (0:02:00)" gpurun_out/runN_$n.err
  grep -h "comm.py\|step.py\|bench.py\", line" gpurun_out/runN_$n.err | sort | uniq -c | head -8
  python3 - gpurun_out/runN_$n.out <<'P'
import json, sys
l = [x for x in open(sys.argv[1]).read().splitlines() if x.startswith("{")]
if l:
    d = json.loads(l[-1]); p = d["dp_policy"]
    print(d["n_gpus"], d["value"], p["chosen"], p["transport"], p["peer_candidate"], {k: round(v, 1) for k, v in p["ms_per_step"].items()})
P
done
