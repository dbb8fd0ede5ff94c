#!/usr/bin/env python3
"""Headline benchmark: HR-patches/sec of the x4 SR GAN train step (48 -> 192) on N MI355X of one node.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one full GAN train step of reference train.py:194-259 (G forward, 4 D forwards, 2 VGG passes,
RSGAN + focal losses, D and G backward, both Adam updates) on a per-GPU batch of 16 synthetic
DIV2K-shaped crops (LR 48x48, HR 192x192, integers 0..255 as fp32) already resident in HBM.  Weak scaling:
per-GPU batch fixed, global batch 16*N, gradients all-reduced over RCCL inside the optimizers.
Rank 0 prints ONE JSON line.  See DESIGN.md "Measurement" for the roofline / cpu_baseline definitions.
"""
import argparse
import json
import os
import sys
import time
import warnings

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md: dense fp32-input MFMA
GFLOP_PER_PATCH = {"gan": 844.10, "pretrain": 694.69}   # SURVEY.md 8(d), necessary work per HR patch
K1_GFLOP = 2.0 * 16 * 48 * 48 * 256 * 256 * 9 / 1e9     # body conv 256->256 @48x48, batch 16: 43.487 GFLOP per launch


def k1_hbm_traffic_bytes(kernel_substr):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (bench.py cannot read PMCs
    live): 2 x FETCH_SIZE (gfx950 correction, MI355X_MICROARCH.md "HBM") + WRITE_SIZE, both in KiB in the profile."""
    import csv
    path = os.path.join(ROOT, "profiles", "r01_final3_k1_pmc_summary.csv")
    try:
        fetch = write = None
        for r in csv.DictReader(open(path)):
            if kernel_substr in r["kernel"]:
                if r["counter"] == "FETCH_SIZE":
                    fetch = float(r["mean_per_launch"])
                elif r["counter"] == "WRITE_SIZE":
                    write = float(r["mean_per_launch"])
        return int((2 * fetch + write) * 1024) if fetch and write else None
    except OSError:
        return None


def build(args, device, world):
    from model import Discriminator, Generator, VGG
    from pesr_amd.optim import FlatAdam
    from pesr_amd.step import Trainer
    torch.manual_seed(0)
    opt = {"patch_size": args.patch_size, "num_channels": args.num_channels, "depth": args.num_blocks,
           "res_scale": 0.1, "spectral_norm": False}
    G = Generator(opt).to(device)
    D = vgg = oD = None
    if args.workload == "gan":
        D = Discriminator(opt).to(device)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            vgg = VGG().to(device)
        oD = FlatAdam(D.parameters(), lr=5e-5, betas=(0.9, 0.999))
    oG = FlatAdam([p for p in G.parameters() if p.requires_grad], lr=5e-5, betas=(0.9, 0.999))
    if world > 1:   # identical replicas: broadcast rank 0's initial weights once
        dist.broadcast(oG.flat.flat_p, 0)
        if oD is not None:
            dist.broadcast(oD.flat.flat_p, 0)
    return Trainer(G, D, vgg, oG, oD, world_size=world), G, D, vgg


def synth_batch(batch, patch, seed, device):
    g = torch.Generator().manual_seed(seed)
    lr = torch.randint(0, 256, (batch, 3, patch, patch), generator=g).float()
    hr = torch.randint(0, 256, (batch, 3, 4 * patch, 4 * patch), generator=g).float()
    return lr.to(device), hr.to(device)


def host_cores() -> int:
    """Cores this process may really use: CPU affinity capped by the cgroup CPU quota (the GPU box exposes 256
    logical CPUs under a 16-CPU quota; oversubscribing torch's pool there is 50x slower)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(args, G, D, vgg):
    """The same train step on the host cores through the CPU oracle (oracle/step.py), on a bounded sample."""
    from oracle import step as OS
    nthreads = host_cores()
    torch.set_num_threads(nthreads)
    B = args.cpu_batch
    cfg = {"depth": args.num_blocks, "res_scale": 0.1, "learning_rate": 5e-5}
    g_sd = {k: v.detach().cpu().clone() for k, v in G.state_dict().items()}
    d_sd = {k: v.detach().cpu().clone() for k, v in D.state_dict().items()} if D is not None else None
    v_sd = {k: v.detach().cpu().clone() for k, v in vgg.state_dict().items()} if vgg is not None else None
    st = OS.TrainState(g_sd, d_sd, v_sd, cfg)
    lr, hr = synth_batch(B, args.patch_size, 4321, "cpu")
    step = OS.gan_step if args.workload == "gan" else OS.pretrain_step
    t0 = time.perf_counter()
    step(st, lr, hr)
    dt = time.perf_counter() - t0
    return {"value": B / dt, "unit": "patches/s", "cores": nthreads, "kind": "port",
            "sample": f"1 {args.workload} step of the CPU oracle (torch {torch.__version__} CPU ops) at batch {B} "
                      f"(1/{16 // B} of a GPU step's batch), full model size, {dt:.1f} s"}


def bench_infer512(args, device):
    """BASELINE config 5 (side measurement, 1 GPU): G forward, no_grad, 4 x 512x512 LR -> 2048x2048; 105.39 TFLOP per batch."""
    from model import Generator
    torch.manual_seed(0)
    G = Generator({"num_channels": 256, "depth": 32, "res_scale": 0.1}).to(device)
    g = torch.Generator().manual_seed(5)
    x = torch.randint(0, 256, (4, 3, 512, 512), generator=g).float().to(device)
    with torch.no_grad():
        for _ in range(max(1, args.warmup)):
            G(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            y = G(x)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    tf = 105.3875 / dt
    print(json.dumps({"metric": "LR tiles/sec (x4 SR generator forward, 512x512 LR tiles, batch 4)", "value": round(4 / dt, 3),
                      "unit": "tiles/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt * 1e3, 2),
                      "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                      "config": {"workload": "BASELINE config 5: G forward no_grad, 4x3x512x512 -> 4x3x2048x2048, 256 ch x 32 blocks"},
                      "step_tflops_per_gpu": round(tf, 2), "step_frac_of_mfma_peak": round(tf / PEAK_F32_MFMA_TFLOPS, 4),
                      "out_checksum": float(y.double().sum())}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=["gan", "pretrain", "infer512"], default="gan",
                    help="gan = BASELINE config 3/4 (the headline metric); pretrain = config 2; infer512 = config 5 "
                         "(G forward on 4 x 512x512 LR tiles, no_grad) - the latter two are side measurements")
    ap.add_argument("--batch", type=int, default=16, help="per-GPU batch")
    ap.add_argument("--patch_size", type=int, default=48)
    ap.add_argument("--num_channels", type=int, default=256)
    ap.add_argument("--num_blocks", type=int, default=32)
    ap.add_argument("--cpu_batch", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    under_launcher = "RANK" in os.environ and "MASTER_PORT" in os.environ
    if world > 1 or under_launcher:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from pesr_amd import ops
    if args.workload == "infer512":
        return bench_infer512(args, device)
    trainer, G, D, vgg = build(args, device, world)
    lr, hr = synth_batch(args.batch, args.patch_size, 1234 + rank, device)
    step = trainer.gan_step if args.workload == "gan" else trainer.pretrain_step

    for _ in range(args.warmup):
        step(lr, hr)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    if not args.no_kernel_events:
        ops.KERNEL_EVENTS.enable(shape=(args.batch, args.patch_size, args.patch_size, args.num_channels, args.num_channels, 1))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        logs = step(lr, hr)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kern_ms, kern_n = ops.KERNEL_EVENTS.drain()
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank != 0:
        dist.destroy_process_group()
        return

    global_batch = args.batch * world
    value = args.steps * global_batch / elapsed
    flop_patch = GFLOP_PER_PATCH[args.workload] * 1e9
    out = {
        "metric": "HR-patches/sec (x4 SR GAN train step, 48->192)",
        "value": round(value, 3), "unit": "patches/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": ("full GAN phase (G + D + VGG + RSGAN focal loss), " if args.workload == "gan"
                                else "pretrain phase (L1 only), ") +
                               f"per-GPU batch {args.batch}, LR {args.patch_size}x{args.patch_size} -> HR "
                               f"{4 * args.patch_size}x{4 * args.patch_size}, {args.num_channels} ch x {args.num_blocks} blocks",
                   "global_batch": global_batch, "parallelism": f"dp{world}"},
        "step_tflops_per_gpu": round(value / world * flop_patch / 1e12, 2),
        "step_frac_of_mfma_peak": round(value / world * flop_patch / (PEAK_F32_MFMA_TFLOPS * 1e12), 4),
        "losses": {k: float(v) for k, v in logs.items()},
    }
    if kern_n:
        ach = K1_GFLOP * (args.batch / 16) * (args.patch_size / 48) ** 2 * (args.num_channels / 256) ** 2 / kern_ms  # TFLOP/s
        from pesr_amd import ops as _ops
        wino = _ops.wino_eligible(args.batch, args.patch_size, args.patch_size, args.num_channels, args.num_channels)
        kname = "conv3x3_wino_kernel" if wino else "conv3x3_mfma_kernel<1, 8, 9, 2, 1, 2"
        out["roofline"] = {"kernel": (kname if wino else "conv3x3_mfma_kernel<1,8,9,2,1,2>") +
                                     " forward (G body 256->256 @48x48, 65 launches/step)",
                           "bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                           "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4), "traffic": k1_hbm_traffic_bytes(kname),
                           "traffic_note": "HBM bytes/launch (2*FETCH_SIZE + WRITE_SIZE) from the separate rocprofv3 --pmc passes in "
                                           "profiles/r01_final3_k1_pmc_summary.csv (mean of fwd and dgrad launches); algorithmic: 78-116 MB",
                           "launches_timed": kern_n, "avg_launch_us": round(kern_ms * 1e3, 2)}
        if wino:   # `achieved` counts the conv's ALGORITHMIC flops; the Winograd kernel executes 2/3 of them
            out["roofline"]["note"] = ("1-D Winograd F(2,3): the kernel issues 2/3 of the direct conv's MFMA flops, so achieved/peak may "
                                       "exceed 1; matrix-pipe utilisation = mfma_util")
            out["roofline"]["mfma_util"] = round(ach * (2.0 / 3.0) / PEAK_F32_MFMA_TFLOPS, 4)
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(args, G, D, vgg)
    print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
