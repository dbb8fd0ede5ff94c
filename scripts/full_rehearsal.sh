#!/bin/bash
# Rehearsal (round 6; rounds 5's full4_rehearsal.sh generalised): bench.py at the FULL model size with N ranks time-sharing cuda:0
# (test hook; NOT a scaling measurement): gradients over comm.PeerCopy (transport=peer) or over gloo (transport=torch).
#   scripts/full_rehearsal.sh <ranks> <peer|torch> [extra bench.py flags]
# Writes gpurun_out/full<N>_<transport>.{out,err,json}.
N=${1:-8}; TR=${2:-peer}; shift; shift
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PESR_DP_BACKEND=gloo PESR_DP_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0 PESR_DUMP_STACKS_AFTER=${PESR_DUMP_STACKS_AFTER:-500} PESR_DP_TRANSPORT=$TR
tag=full${N}_${TR}
timeout ${REHEARSAL_TIMEOUT:-600} python3 -m torch.distributed.run --nnodes=1 --nproc-per-node=$N --master-addr 127.0.0.1 --master-port $((29500 + N)) \
    bench.py --gpus $N --steps 3 --warmup 2 --calib-steps 1 "$@" > gpurun_out/$tag.out 2> gpurun_out/$tag.err
echo "$tag rc=$?"
grep -h "Timeout\|comm.py\|step.py\|Error" gpurun_out/$tag.err | sort | uniq -c | head
python3 - "$tag" <<'P'
import json, sys
tag = sys.argv[1]
l = [x for x in open(f"gpurun_out/{tag}.out").read().splitlines() if x.startswith("{")]
if l:
    d = json.loads(l[-1]); p = d["dp_policy"]
    print(d["n_gpus"], d["value"], d["ms_per_step"], p["chosen"], p["transport"], p.get("peer_candidate"), {k: round(v, 1) for k, v in p.get("ms_per_step", {}).items()})
    print("bringup:", d.get("dp_bringup")); print("collectives:", d.get("dp_collectives")); print("engine:", d.get("peer_copy_engine")); print("replicas:", d.get("replica_check"))
    open(f"gpurun_out/{tag}.json", "w").write(l[-1] + "\n")
P
