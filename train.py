#!/usr/bin/env python3
"""Training entry point (counterpart of reference train.py) on the MI355X-native model package.

Same flags and defaults as the reference (train.py:21-77) plus `--synthetic N` (random DIV2K-shaped crops instead of
the DIV2K cache, which is out of scope here).  One process per GPU: run plainly for one GPU, or under
`python -m torch.distributed.run --nproc-per-node N train.py ...` for N (RCCL gradient all-reduce inside the
optimizers replaces nn.DataParallel; --batch_size stays the GLOBAL batch as in the reference, each rank takes 1/N).
tensorboardX is optional.  Checkpoints keep the reference's format: torch.save(G.state_dict()).
"""
import argparse
import json
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.optim.lr_scheduler as lr_scheduler
from torch.utils.data import DataLoader
from torch.utils.data.distributed import DistributedSampler


def str2bool(x):
    return str(x).lower() == "true"


def build_parser():
    p = argparse.ArgumentParser(description="PIRM 2018")
    p.add_argument("--scale", type=int, default=4)
    p.add_argument("--train_dataset", type=str, default="DIV2K")
    p.add_argument("--valid_dataset", type=str, default="PIRM")
    p.add_argument("--num_valids", type=int, default=10)
    p.add_argument("--num_channels", type=int, default=256)
    p.add_argument("--num_blocks", type=int, default=32)
    p.add_argument("--res_scale", type=float, default=0.1)
    p.add_argument("--phase", type=str, default="train", help="pretrain or train")
    p.add_argument("--pretrained_model", type=str, default="")
    p.add_argument("--batch_size", type=int, default=16)
    p.add_argument("--learning_rate", type=float, default=5e-5)
    p.add_argument("--lr_step", type=int, default=120)
    p.add_argument("--num_epochs", type=int, default=200)
    p.add_argument("--num_repeats", type=int, default=20)
    p.add_argument("--patch_size", type=int, default=24)
    p.add_argument("--check_point", type=str, default="check_point/my_model")
    p.add_argument("--snapshot_every", type=int, default=10)
    p.add_argument("--gan_type", type=str, default="RSGAN")
    p.add_argument("--GP", type=str2bool, default=False)
    p.add_argument("--spectral_norm", type=str2bool, default=False)
    p.add_argument("--focal_loss", type=str2bool, default=True)
    p.add_argument("--fl_gamma", type=float, default=1)
    p.add_argument("--alpha_vgg", type=float, default=50)
    p.add_argument("--alpha_gan", type=float, default=1)
    p.add_argument("--alpha_tv", type=float, default=1e-6)
    p.add_argument("--alpha_l1", type=float, default=0)
    # additions
    p.add_argument("--synthetic", type=int, default=0, help="train on N synthetic samples per epoch instead of a dataset folder")
    p.add_argument("--vgg_weights", type=str, default="",
                   help="state_dict file of torchvision's pretrained vgg19 (torch.save(torchvision.models.vgg19(pretrained=True)"
                        ".state_dict(), path)); REQUIRED for the GAN phase unless torchvision itself is importable")
    p.add_argument("--allow_random_vgg", type=str2bool, default=False,
                   help="GAN phase without pretrained VGG19 weights: seeded random features (benchmarks / smoke runs only - the "
                        "perceptual loss is then NOT the reference's)")
    p.add_argument("--max_iters", type=int, default=0, help="stop each epoch after this many iterations (smoke runs)")
    p.add_argument("--hip_graph", type=str, default="auto", choices=["auto", "true", "false", "True", "False"],
                   help="capture the train step into a hipGraph after two eager iterations and replay it (same results bit for "
                        "bit; the host no longer issues ~1000 launches per step).  auto = on for a single-GPU run; under "
                        "torch.distributed the captured step (RCCL all-reduces included) is one candidate of --dp_policy auto")
    p.add_argument("--dp_peer_candidate", type=str, default="true",
                   help="--dp_policy auto also times the CU-free gradient exchange over peer memory (pesr_amd/comm.py PeerCopy), after a "
                        "rehearsal in child processes; it replaces RCCL only if it is faster")
    p.add_argument("--dp_policy", type=str, default="auto", choices=["auto", "overlap", "defer_g", "defer_all"],
                   help="under torch.distributed: how the gradient all-reduces are scheduled against the backward kernels.  auto = "
                        "measured during the first iterations (Trainer.calibrate_dp_policy: eager overlap, G's exchange deferred behind "
                        "its backward pass, both deferred, and - unless --hip_graph false - the captured step) and the fastest kept")
    p.add_argument("--precision", type=str, default="fp32", choices=["fp32", "bf16", "split-bf16"],
                   help="fp32 (default): the reference's arithmetic.  bf16: OPTIONAL mixed-precision mode - the stride-1 3x3 convs with "
                        "32-multiple input / 64-multiple output channels run on the bf16 MFMA (both operands rounded to bf16, fp32 "
                        "accumulation, fp32 tensors and optimizer): ~2.4x the step rate, NOT the reference's numerics.  split-bf16: "
                        "OPTIONAL - forward and input gradient of the stride-1 convs with Cout % 128 == 0 as THREE bf16 products per "
                        "multiply (hi / lo operand split): 4e-6 of the output maximum, inside the fp32 kernels' own tolerances")
    p.add_argument("--gpu_pipeline", type=str2bool, default=False,
                   help="keep the uint8 training images in HBM and crop/augment on the GPU (pesr_amd.input_pipeline)")
    return p


def make_loaders(args, rank, world, need_train=True):
    from data import FolderSRDataset, SyntheticSRDataset
    train_set = None
    if args.synthetic:
        train_set = SyntheticSRDataset(args.synthetic, args.patch_size)
        val_set = SyntheticSRDataset(min(args.num_valids, 2), args.patch_size, seed=99)
    else:
        if need_train:      # (the GPU input pipeline replaces the host training loader altogether)
            train_set = FolderSRDataset(os.path.join("data/origin/train", args.train_dataset), args.patch_size, args.num_repeats, True)
        val_set = FolderSRDataset(os.path.join("data/origin/valid", args.valid_dataset), None, 1, False, fixed_length=10)
    sampler = train_loader = None
    if train_set is not None:
        sampler = DistributedSampler(train_set, world, rank, shuffle=True, drop_last=True) if world > 1 else None
        train_loader = DataLoader(train_set, batch_size=args.batch_size // world, shuffle=sampler is None, sampler=sampler,
                                  num_workers=4, pin_memory=True, drop_last=True, persistent_workers=True)   # (forked once: a fork of a
        # process with a GPU context's mappings takes seconds, and without this flag every epoch forks its four workers anew)
    val_loader = DataLoader(val_set, batch_size=1, shuffle=False, num_workers=1, pin_memory=True)
    return train_loader, val_loader, sampler


class GpuLoader:
    """Iterable with the DataLoader's contract (yields (lr, hr) batches) on top of GpuPatchSampler, with the reference's epoch
    composition (reference data.py:60-77 + DataLoader(shuffle=True, drop_last=True), train.py:96-97): every image appears
    exactly num_repeats times per epoch in a fresh random order; under torch.distributed the epoch's permutation (seeded by
    the epoch, identical on all ranks) is dealt round-robin to the ranks like DistributedSampler does, so no image is drawn
    twice into one global batch position and every sample is seen by exactly one rank."""

    def __init__(self, sampler, batch, patch, n_images, num_repeats, rank, world, seed=1):
        import random
        self.sampler, self.batch, self.patch = sampler, batch, patch
        self.n, self.rep, self.rank, self.world, self.seed = n_images, num_repeats, rank, world, seed
        self.epoch = 0
        self.rng = random.Random(seed * 7919 + rank)          # crop origins / augmentation: per-rank stream

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        return (self.n * self.rep) // (self.batch * self.world)

    def epoch_indices(self):
        import random
        order = list(range(self.n)) * self.rep
        random.Random(self.seed * 1000003 + self.epoch).shuffle(order)
        usable = len(self) * self.batch * self.world            # drop_last
        return order[self.rank:usable:self.world]

    def __iter__(self):
        idx = self.epoch_indices()
        for b in range(len(self)):
            picks = self.sampler.draw_for(idx[b * self.batch:(b + 1) * self.batch], self.patch, self.rng, augment=True)
            yield self.sampler.assemble(picks, self.patch, nhwc=True)


def make_gpu_loader(args, rank, world, device):
    import glob
    from PIL import Image
    from pesr_amd.input_pipeline import GpuPatchSampler
    root = os.path.join("data/origin/train", args.train_dataset)
    lr_paths = sorted(glob.glob(os.path.join(root, "LR", "*.png")))
    lrs = [np.asarray(Image.open(p).convert("RGB")) for p in lr_paths]
    hrs = [np.asarray(Image.open(os.path.join(root, "HR", os.path.basename(p))).convert("RGB")) for p in lr_paths]
    return GpuLoader(GpuPatchSampler(lrs, hrs, device), args.batch_size // world, args.patch_size, len(lrs), args.num_repeats,
                     rank, world)


def build_vgg(args, device, rank, world):
    """The reference's perceptual loss uses torchvision's PRETRAINED vgg19 (reference model/vgg.py:8); training against anything
    else silently produces a different model.  Order of preference: --vgg_weights file; torchvision's own pretrained weights if
    that package is importable; else refuse, unless --synthetic / --allow_random_vgg ask for seeded random features - which are
    then made identical on every rank (seeded construction + broadcast from rank 0), so all replicas optimise the same loss."""
    import warnings
    from model import VGG
    if args.vgg_weights:
        return VGG(args.vgg_weights).to(device)
    # Only "the package or its model-zoo file is not there" counts as unavailable (ImportError; URLError / OSError of the
    # download); anything else - a corrupt cache, out of memory - is an error, not a reason to train on random features.
    import urllib.error
    sd, why = None, ""
    try:
        import torchvision
        sd = torchvision.models.vgg19(pretrained=True).state_dict()
    except (ImportError, urllib.error.URLError, OSError) as e:
        why = type(e).__name__
    if world > 1:
        # every rank must take the same branch: the pretrained weights are used only if ALL ranks have them
        ok = torch.tensor([1 if sd is not None else 0], device=device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            sd = None
    if sd is not None:
        import tempfile
        with tempfile.NamedTemporaryFile(suffix=".pt") as f:
            torch.save(sd, f.name)
            return VGG(f.name).to(device)
    if not (args.synthetic or args.allow_random_vgg):
        raise SystemExit("train.py: the GAN phase needs torchvision's pretrained vgg19 weights (reference model/vgg.py:8): pass "
                         "--vgg_weights <state_dict file>, or --allow_random_vgg true for a run whose perceptual loss is NOT the "
                         f"reference's (torchvision's weights unavailable on at least one rank: {why or 'another rank'})")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        state = torch.random.get_rng_state()
        torch.manual_seed(20180917)
        vgg = VGG().to(device)
        torch.random.set_rng_state(state)
    if world > 1:
        for p in vgg.parameters():
            dist.broadcast(p.data, 0)
    if rank == 0:
        print("WARNING: VGG19 features are RANDOM (seeded): the perceptual loss of this run is not the reference's")
    return vgg


def check_limits(args, world):
    """Shape limits of the HIP path, checked up front with a clear message instead of a PESR_EINVAL deep inside a step."""
    if args.batch_size % world:
        raise SystemExit(f"train.py: --batch_size {args.batch_size} must be a multiple of the {world} ranks")
    if args.phase != "pretrain" and args.patch_size % 4:
        raise SystemExit("train.py: the GAN phase needs --patch_size % 4 == 0 (HR patches must survive four stride-2 stages and "
                         "VGG's four 2x2 max-pools, which this implementation does for even sizes only)")


def main(argv=None):
    args = build_parser().parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # TEST HOOK (tests/test_dp_gpu.py): PESR_DP_SHARE_GPU=1 + PESR_DP_BACKEND=gloo lets the ranks time-share cuda:0 over gloo, so
    # that the multi-rank branches of this script run on a one-GPU box
    backend = os.environ.get("PESR_DP_BACKEND", "nccl")
    if os.environ.get("PESR_DP_SHARE_GPU") == "1":
        assert backend == "gloo", "RCCL refuses two ranks on one device"
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    check_limits(args, world)
    if rank == 0:
        print("_____________YOUR SETTINGS_____________")
        for k, v in vars(args).items():
            print("%20s: %s" % (k, v))

    from model import Discriminator, Generator, VGG
    from pesr_amd import ops as _ops
    _ops.set_precision(args.precision)
    from pesr_amd.optim import FlatAdam
    from pesr_amd.step import Trainer
    from utils import compute_PSNR

    gpu_pipe = args.gpu_pipeline and not args.synthetic
    train_loader, val_loader, sampler = make_loaders(args, rank, world, need_train=not gpu_pipe)
    if gpu_pipe:
        train_loader = sampler = make_gpu_loader(args, rank, world, device)      # (it has set_epoch like a DistributedSampler)
    opt = {"patch_size": args.patch_size, "num_channels": args.num_channels, "depth": args.num_blocks,
           "res_scale": args.res_scale, "spectral_norm": args.spectral_norm}
    G = Generator(opt)
    if args.pretrained_model:
        G.load_state_dict(torch.load(args.pretrained_model, map_location="cpu"))
    G = G.to(device)
    gan = args.phase != "pretrain"
    D = Discriminator(opt).to(device) if gan else None
    vgg = build_vgg(args, device, rank, world) if gan else None

    optim_G = FlatAdam([p for p in G.parameters() if p.requires_grad], lr=args.learning_rate, betas=(0.9, 0.999))
    optim_D = FlatAdam(D.parameters(), lr=args.learning_rate, betas=(0.9, 0.999)) if gan else None
    if world > 1:  # replicas start identical
        dist.broadcast(optim_G.flat.flat_p, 0)
        if gan:
            dist.broadcast(optim_D.flat.flat_p, 0)
    scheduler_G = lr_scheduler.StepLR(optim_G, step_size=args.lr_step, gamma=0.5)
    scheduler_D = lr_scheduler.StepLR(optim_D, step_size=args.lr_step, gamma=0.5) if gan else None
    trainer = Trainer(G, D, vgg, optim_G, optim_D, gan_type=args.gan_type, focal_loss=args.focal_loss, fl_gamma=args.fl_gamma,
                      alpha_vgg=args.alpha_vgg, alpha_gan=args.alpha_gan, alpha_tv=args.alpha_tv, alpha_l1=args.alpha_l1,
                      world_size=world, gradient_penalty=args.GP)

    check_point = os.path.join(args.check_point, args.phase)
    tb = None
    if rank == 0:
        os.makedirs(check_point, exist_ok=True)
        try:
            from tensorboardX import SummaryWriter
            tb = SummaryWriter(check_point)
        except ImportError:
            pass
    best_psnr = 0.0
    keys = ("l1", "vgg", "g", "tv", "d") if gan else ("l1",)
    graphed, graph_shapes, eager_at_shape = None, None, 0
    use_graph = device.type == "cuda" and (world == 1 if args.hip_graph == "auto" else str2bool(args.hip_graph))
    # data-parallel schedule: fixed by flag, or measured once (inside the first epoch that is long enough for it)
    dp_on = optim_G.buckets.enabled
    dp_calibrate = dp_on and args.dp_policy == "auto"
    if dp_on and args.dp_policy != "auto":
        trainer.set_dp_policy(args.dp_policy)
    CALIB_STEPS = 3
    peer_cand = str2bool(args.dp_peer_candidate) and world > 1
    # iterations the calibration consumes at most (its own worst-case count: eager candidates, the peer-memory candidate, the capture's
    # batch and the replays - ADVICE r05: the peer candidate was not counted and the calibration could run off the end of an epoch)
    calib_need = Trainer.calibration_batches("gan" if gan else "pretrain", CALIB_STEPS, graph=args.hip_graph == "auto", peer_candidate=peer_cand) + 2

    for epoch in range(1, args.num_epochs + 1):
        # The reference calls scheduler.step() at epoch START (train.py:156,185-186); under its pinned torch 0.4 the
        # constructor leaves last_epoch = -1, so epoch e trains at lr*0.5^((e-1)//lr_step) (first halving: epoch lr_step+1).
        # On torch >= 1.1 the constructor already counts one step, so the same schedule is: step at the END of every epoch.
        cur_lr = optim_G.param_groups[0]["lr"]
        if sampler is not None:
            sampler.set_epoch(epoch)
        running = torch.zeros(len(keys), device=device)      # accumulated on the device: one host sync per epoch
        iters = 0
        batch_iter = iter(train_loader)
        n_iters = len(train_loader) if not args.max_iters else min(len(train_loader), args.max_iters)
        consumed = [0]

        def next_batch():
            a, b = next(batch_iter)
            consumed[0] += 1
            return a.to(device, non_blocking=True), b.to(device, non_blocking=True)

        while consumed[0] < n_iters:
            try:
                lr_img, hr_img = next_batch()
            except StopIteration:
                break
            if graphed is not None and lr_img.shape == graph_shapes[0] and hr_img.shape == graph_shapes[1]:
                logs = graphed(lr_img, hr_img)
            else:
                logs = trainer.gan_step(lr_img, hr_img) if gan else trainer.pretrain_step(lr_img, hr_img)
                if dp_calibrate and iters >= 1 and n_iters - consumed[0] >= calib_need:
                    # every rank is at the same iteration of equally long loaders: the calibration's collectives line up.  Its
                    # steps are real training steps (their losses are not added to this epoch's averages).
                    info = trainer.calibrate_dp_policy("gan" if gan else "pretrain", next_batch, steps=CALIB_STEPS, peer_candidate=peer_cand,
                                                       graph=args.hip_graph == "auto",
                                                       will_capture=args.hip_graph != "auto" and str2bool(args.hip_graph))
                    dp_calibrate = False
                    if rank == 0:
                        print("data-parallel schedule:", info)
                        # a run is reproduced bit for bit with an explicit --dp_policy (the bucket schedule decides RCCL's
                        # reduction order): the choice is kept next to the checkpoints
                        with open(os.path.join(check_point, "dp_policy.json"), "w") as fh:
                            json.dump({"dp_policy": trainer.dp_policy, **info}, fh, indent=1)
                    if info["chosen"].startswith("graph"):
                        graphed, graph_shapes = trainer.dp_step, (lr_img.shape, hr_img.shape)
                    # (--hip_graph true / false: the flag decides and only the bucket schedule was measured; the capture below
                    # then runs with that schedule)
                if use_graph and graphed is None and not dp_calibrate:
                    # capture after TWO eager steps at these shapes: the second one has seen every weight packing the first one
                    # created (some only in its backward pass), so nothing is allocated or uploaded under capture
                    shapes = (lr_img.shape, hr_img.shape)
                    eager_at_shape = eager_at_shape + 1 if shapes == graph_shapes else 1
                    graph_shapes = shapes
                    if eager_at_shape >= 2:
                        graphed = (trainer.capture_gan_step if gan else trainer.capture_pretrain_step)(lr_img, hr_img)
            running += torch.stack([logs[k].float() for k in keys])
            iters += 1
        if dp_calibrate:
            # No epoch will get longer: with too few iterations left behind the second step for the measurement (short epochs
            # or a small --max_iters) the schedule stays the default one - said once, and --hip_graph true can capture now.
            dp_calibrate = False
            if rank == 0:
                print("data-parallel schedule: an epoch has %d iterations, the measurement needs %d behind the second one - keeping "
                      "'%s' (choose with --dp_policy)" % (n_iters, calib_need, trainer.dp_policy))
        if world > 1:
            dist.all_reduce(running)
            running /= world
            if gan:  # the TV term is a sum over the global batch
                running[keys.index("tv")] *= world
        avg = (running / max(iters, 1)).tolist()
        if rank == 0:
            print("Epoch [%d/%d] lr %g  " % (epoch, args.num_epochs, cur_lr) + "  ".join("%s %.4f" % kv for kv in zip(keys, avg)))
            if tb is not None:
                tb.add_scalar("Learning rate", cur_lr, epoch)
                names = {"l1": "L1 Loss", "vgg": "VGG Loss", "g": "G Loss", "tv": "TV Loss", "d": "D Loss"}
                for k, v in zip(keys, avg):
                    tb.add_scalar(names[k] if gan else "Pretrain Loss", v, epoch)

        # validation on rank 0 (full images, batch 1, no_grad), reference train.py:281-295
        if rank == 0:
            psnr = []
            with torch.no_grad():
                for lr_img, hr_img in val_loader:
                    sr = G(lr_img.to(device))
                    psnr.append(compute_PSNR(hr_img.to(device), sr))
            val_psnr = float(np.mean(psnr)) if psnr else 0.0
            print("Finish valid [%d/%d]. PSNR: %.4fdB" % (epoch, args.num_epochs, val_psnr))
            if tb is not None:
                tb.add_scalar("Validate PSNR", val_psnr, epoch)
            if not gan and val_psnr > best_psnr:
                best_psnr = val_psnr
                torch.save(G.state_dict(), os.path.join(check_point, "best_model.pt"))
            elif gan and epoch % args.snapshot_every == 0:
                torch.save(G.state_dict(), os.path.join(check_point, "model_%d.pt" % epoch))
        scheduler_G.step()
        if gan:
            scheduler_D.step()
        if world > 1:
            dist.barrier()
    if world > 1:
        from pesr_amd import comm
        comm.close_transports()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
