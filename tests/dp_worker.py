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

CONFIGS = {"small": dict(C=64, depth=2, ps=24, B=4, steps=2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--config", default="small")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    world, rank, local = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev)
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
    tr = Trainer(G, D, V, oG, oD, world_size=world)
    losses = []
    for it in range(cfg["steps"]):
        lr = detrand.image_batch((B * world, 3, ps, ps), 700 + it)
        hr = detrand.image_batch((B * world, 3, 4 * ps, 4 * ps), 800 + it)
        sh = slice(rank * B, (rank + 1) * B)
        log = tr.gan_step(lr[sh].to(dev), hr[sh].to(dev))
        keys = ("l1", "vgg", "g", "tv", "d")
        t = torch.stack([log[k].float() for k in keys])
        dist.all_reduce(t)                       # as train.py logs them: mean-type terms averaged, the TV sum summed
        t = t / world
        t[keys.index("tv")] *= world
        losses.append(t.cpu())
        if it == 0:      # gradients of the FIRST step (averaged over ranks = the full-batch ones): the tight comparison
            g0 = {"G": {k: (p.grad * oG.last_scale).cpu() for k, p in G.named_parameters()},
                  "D": {k: (p.grad * oD.last_scale).cpu() for k, p in D.named_parameters()}}
    # replicas must hold bit-identical parameters (same all-reduced gradients, same Adam)
    for opt in (oG, oD):
        mine = opt.flat.flat_p.clone()
        ref = mine.clone()
        dist.broadcast(ref, 0)
        assert torch.equal(mine, ref), "replicas diverged"
    if rank == 0:
        torch.save({"losses": torch.stack(losses),
                    "G": {k: v.cpu() for k, v in G.state_dict().items()}, "D": {k: v.cpu() for k, v in D.state_dict().items()},
                    "G.grad": g0["G"], "D.grad": g0["D"],
                    "world": world}, args.out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
