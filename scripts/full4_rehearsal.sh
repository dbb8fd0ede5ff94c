# Rehearsal (round 5): bench.py at the FULL model size with four ranks time-sharing cuda:0, gradients over comm.PeerCopy (test hook; NOT a scaling measurement)
cd /root/repo
export PESR_DP_BACKEND=gloo PESR_DP_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0 PESR_DUMP_STACKS_AFTER=200 PESR_DP_TRANSPORT=peer
timeout 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node=4 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 4 --steps 3 --warmup 2 --calib-steps 1 > gpurun_out/full4.out 2> gpurun_out/full4.err
echo rc=$?
grep -h "Timeout\|comm.py\|step.py" gpurun_out/full4.err | sort | uniq -c | head
python3 - <<'P'
import json
l = [x for x in open("gpurun_out/full4.out").read().splitlines() if x.startswith("{")]
if l:
    d = json.loads(l[-1]); p = d["dp_policy"]
    print(d["n_gpus"], d["value"], d["ms_per_step"], p["chosen"], p["transport"], p["peer_candidate"], {k: round(v, 1) for k, v in p["ms_per_step"].items()})
    open("gpurun_out/full4.json", "w").write(l[-1] + "\n")
P
