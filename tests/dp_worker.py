"""One rank of the multi-GPU parity job (started by tests/test_dp_gpu.py under torch.distributed.run, one process per GPU).

Each rank builds the same networks, takes its contiguous shard of the global batch (what nn.DataParallel's scatter does,
reference train.py:114-118), runs GAN steps through Trainer(world_size=N) with the RCCL bucketed all-reduce, and rank 0
writes losses, parameters and gradients for the parent test to compare with the CPU oracle's full-batch step."""
import argparse
import os
import sys
import warnings

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

# "small": the GAN step as trained; "pretrain": the L1 step (well conditioned: pins 1/N and the shard layout tightly);
# "tv": a GAN step whose generator gradient is the TV term alone (alpha_tv = 1, everything else 0) - TV is a SUM over the
# global batch (reference train.py:137-140), so this pins the x world_size of pesr_amd/step.py
# "policy": the bucket schedule the run uses (Trainer.set_dp_policy) - each of the three is held to the same parity once
CONFIGS = {"small": dict(C=64, depth=2, ps=24, B=4, steps=2, kind="gan", alphas={}, policy="overlap"),
           "pretrain": dict(C=64, depth=2, ps=24, B=4, steps=2, kind="pretrain", alphas={}, policy="defer_g"),
           "tv": dict(C=64, depth=2, ps=24, B=4, steps=1, kind="gan", alphas=dict(alpha_tv=1.0, alpha_vgg=0.0, alpha_gan=0.0, alpha_l1=0.0),
                      policy="defer_all")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--config", default="small")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    world, rank, local = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"])
    backend = os.environ.get("PESR_DP_BACKEND", "nccl")
    if os.environ.get("PESR_DP_SHARE_GPU") == "1":
        local = 0            # N ranks time-share ONE GPU (gloo moves the buckets through host memory): everything of the
        #                      data-parallel path except RCCL itself, on a box where only one GPU is visible
        assert backend == "gloo", "RCCL refuses two ranks on one device"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(backend)
    warnings.filterwarnings("ignore", message=".*pretrained vgg19.*")
    from helpers import dis_sd, gen_sd, vgg_sd
    from model import Discriminator, Generator, VGG
    from oracle import detrand
    from pesr_amd.optim import FlatAdam
    from pesr_amd.step import Trainer
    C, depth, ps, B = cfg["C"], cfg["depth"], cfg["ps"], cfg["B"]
    G = Generator({"num_channels": C, "depth": depth, "res_scale": 0.1}); G.load_state_dict(gen_sd(C, depth)); G.to(dev)
    D = Discriminator({"patch_size": ps, "spectral_norm": False}); D.load_state_dict(dis_sd(ps)); D.to(dev)
    V = VGG(); V.load_state_dict(vgg_sd()); V.to(dev)
    oG = FlatAdam(G.parameters(), lr=5e-5, bucket_bytes=64 << 10)      # small buckets: several all-reduces per backward
    oD = FlatAdam(D.parameters(), lr=5e-5, bucket_bytes=256 << 10)
    assert oG.buckets.enabled and oD.buckets.enabled and len(oG.buckets.bounds) > 2
    tr = Trainer(G, D, V, oG, oD, world_size=world, **cfg["alphas"])
    tr.set_dp_policy(cfg["policy"])
    gan = cfg["kind"] == "gan"
    losses = []
    for it in range(cfg["steps"]):
        lr = detrand.image_batch((B * world, 3, ps, ps), 700 + it)
        hr = detrand.image_batch((B * world, 3, 4 * ps, 4 * ps), 800 + it)
        sh = slice(rank * B, (rank + 1) * B)
        log = (tr.gan_step if gan else tr.pretrain_step)(lr[sh].to(dev), hr[sh].to(dev))
        keys = ("l1", "vgg", "g", "tv", "d") if gan else ("l1",)
        t = torch.stack([log[k].float() for k in keys])
        dist.all_reduce(t)                       # as train.py logs them: mean-type terms averaged, the TV sum summed
        t = t / world
        if gan:
            t[keys.index("tv")] *= world
        losses.append(t.cpu())
        if it == 0:      # gradients of the FIRST step (averaged over ranks = the full-batch ones): the tight comparison
            g0 = {"G": {k: (p.grad * oG.last_scale).cpu() for k, p in G.named_parameters()},
                  "D": {k: (p.grad * oD.last_scale).cpu() for k, p in D.named_parameters()} if gan else {}}
    # replicas must hold bit-identical parameters (same all-reduced gradients, same Adam)
    for opt in ((oG, oD) if gan else (oG,)):
        mine = opt.flat.flat_p.clone()
        ref = mine.clone()
        dist.broadcast(ref, 0)
        assert torch.equal(mine, ref), "replicas diverged"
    if rank == 0:
        torch.save({"losses": torch.stack(losses),
                    "G": {k: v.cpu() for k, v in G.state_dict().items()}, "D": {k: v.cpu() for k, v in D.state_dict().items()},
                    "G.grad": g0["G"], "D.grad": g0["D"],
                    "world": world, "transport": oG.buckets.transport.name, "policy": tr.dp_policy,
                    "launches": {"G": oG.buckets.launches, "D": oD.buckets.launches, "G.buckets": len(oG.buckets.bounds),
                                 "D.buckets": len(oD.buckets.bounds)}}, args.out)
    dist.barrier()
    from pesr_amd import comm
    comm.close_transports()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
