"""CPU, world_size 2 (gloo): the data-parallel path - flat gradient buckets all-reduced from backward hooks, and the
loss scaling that reproduces nn.DataParallel's full-batch losses (mean-type losses averaged, TV summed)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

from helpers import gen_sd
from oracle import detrand
from oracle import model as OM
from oracle import step as OS


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pesr_amd.optim import FlatParams, GradBuckets
        torch.set_num_threads(2)
        sd = gen_sd(16, 1)
        params = [torch.nn.Parameter(v.clone()) for v in sd.values()]
        names = list(sd.keys())
        flat = FlatParams(params)
        buckets = GradBuckets(flat, bucket_bytes=8 << 10)      # many small buckets -> exercises the hook path
        assert buckets.enabled and len(buckets.bounds) > 3
        leaves = dict(zip(names, params))
        B = 4
        lr = detrand.image_batch((B, 3, 8, 8), 11); hr = detrand.image_batch((B, 3, 32, 32), 12)
        sh = slice(rank * B // world, (rank + 1) * B // world)           # contiguous shard, as DataParallel's scatter
        for _ in range(2):                                               # two rounds: reset() must re-arm the hooks
            flat.zero_grad(); buckets.reset()
            sr = OM.generator_forward(leaves, lr[sh], 1, 0.1)
            loss = F.l1_loss(sr, hr[sh]) + OS.tv_loss(sr) * 1e-3 * world   # local mean + (sum-type term) x world
            loss.backward()
            scale = buckets.finish()
            assert scale == 1.0 / world
        g_avg = flat.flat_g * scale
        # replicas hold identical averaged gradients
        other = [torch.zeros_like(g_avg) for _ in range(world)]
        dist.all_gather(other, g_avg)
        assert all(torch.equal(o, other[0]) for o in other)
        if rank == 0:
            q.put((names, [p.shape for p in params], flat.offsets, g_avg.clone()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gradients_equal_full_batch():
    world = 2
    ctx = mp.get_context("spawn")
    result = None
    for attempt in range(3):        # the rendezvous port is picked before the workers bind it: retry if somebody else took it
        port, q = _free_port(), ctx.Queue()
        procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        try:
            result = q.get(timeout=150)
        except Exception:
            result = None
        for p in procs:
            p.join(60)
            if p.is_alive():
                p.kill()
        if result is not None and all(p.exitcode == 0 for p in procs):
            break
        result = None
    assert result is not None, "the 2-rank gloo job failed three times"
    names, shapes, offsets, g_avg = result
    # single-process full-batch reference: the reference computes its losses on the gathered batch (train.py:240-242)
    sd = gen_sd(16, 1)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    lr = detrand.image_batch((4, 3, 8, 8), 11); hr = detrand.image_batch((4, 3, 32, 32), 12)
    sr = OM.generator_forward(leaves, lr, 1, 0.1)
    (F.l1_loss(sr, hr) + OS.tv_loss(sr) * 1e-3).backward()
    for n, shp, off in zip(names, shapes, offsets):
        ref = leaves[n].grad
        got = g_avg[off:off + ref.numel()].view(shp)
        assert torch.allclose(got, ref, rtol=1e-4, atol=1e-6 * ref.abs().max().item()), n


def test_flat_params_views_and_zero_grad():
    from pesr_amd.optim import FlatParams
    ps = [torch.nn.Parameter(torch.randn(3, 5)), torch.nn.Parameter(torch.randn(7))]
    vals = [p.detach().clone() for p in ps]
    flat = FlatParams(ps)
    assert flat.numel == 16 + 8 and all(o % 4 == 0 for o in flat.offsets)
    for p, v in zip(ps, vals):
        assert torch.equal(p.detach(), v) and p.grad is None and p.data_ptr() >= flat.flat_p.data_ptr()
    (ps[0].sum() * 2 + ps[1].sum()).backward()     # plain autograd: gradients arrive as foreign tensors
    flat.attach_grads()                            # ... and are moved into the flat buffer
    assert torch.equal(flat.flat_g[:15], torch.full((15,), 2.0)) and torch.equal(flat.flat_g[16:23], torch.ones(7))
    assert ps[0].grad.data_ptr() == flat.flat_g.data_ptr()
    (ps[0].sum()).backward()                       # a second backward accumulates in place into the flat view
    assert torch.equal(flat.flat_g[:15], torch.full((15,), 3.0))
    flat.zero_grad()                               # drops the gradients; the buffer is zeroed lazily ...
    assert ps[1].grad is None
    flat.finalize_grads()                          # ... for every parameter that got no gradient in the new step
    assert float(flat.flat_g.abs().sum()) == 0.0
    # the fast path of functional.grad_out: one claim per step
    from pesr_amd import functional as PF
    v1 = PF.grad_out(ps[0]); v2 = PF.grad_out(ps[0])
    assert v1 is not None and v1.data_ptr() == flat.flat_g.data_ptr() and v2 is None
    flat.zero_grad()
    assert PF.grad_out(ps[0]) is not None


def _rasgan_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pesr_amd.model.focal_loss import FocalLoss
        from pesr_amd.step import rasgan_d_loss, rasgan_g_loss
        gen = torch.Generator().manual_seed(5)
        B = 3
        pr_all, pf_all = torch.randn(B * world, 1, generator=gen) * 2, torch.randn(B * world, 1, generator=gen) * 2
        sh = slice(rank * B, (rank + 1) * B)
        pr, pf = pr_all[sh].clone().requires_grad_(True), pf_all[sh].clone().requires_grad_(True)
        ones, zeros = torch.ones(B, 1), torch.zeros(B, 1)
        ld = rasgan_d_loss(pr, pf, ones, zeros, world)
        lg = rasgan_g_loss(pr, pf, ones, zeros, FocalLoss(1.0), world)
        (ld + 3.0 * lg).backward()
        # what the optimizers do with the result: the local losses are MEANS over the shard, gradients are averaged over ranks
        out = torch.stack([ld.detach(), lg.detach()])
        dist.all_reduce(out); out /= world
        grads = [torch.zeros(2 * B, 1) for _ in range(world)]
        dist.all_gather(grads, torch.cat([pr.grad, pf.grad]) / world)
        if rank == 0:
            q.put((out, torch.cat([g[:B] for g in grads]), torch.cat([g[B:] for g in grads])))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_rasgan_losses_equal_full_batch():
    """RaSGAN's batch means are over the GLOBAL batch (step._GlobalBatchMean): two ranks with 3 samples each must produce the
    losses and the per-sample logit gradients of one process holding all 6 (oracle-style full-batch evaluation)."""
    world = 2
    ctx = mp.get_context("spawn")
    result = None
    for attempt in range(3):
        port, q = _free_port(), ctx.Queue()
        procs = [ctx.Process(target=_rasgan_worker, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        try:
            result = q.get(timeout=120)
        except Exception:
            result = None
        for p in procs:
            p.join(60)
            if p.is_alive():
                p.kill()
        if result is not None and all(p.exitcode == 0 for p in procs):
            break
        result = None
    assert result is not None, "the 2-rank gloo job failed three times"
    losses, g_pr, g_pf = result
    gen = torch.Generator().manual_seed(5)
    pr = (torch.randn(6, 1, generator=gen) * 2).requires_grad_(True)
    pf = (torch.randn(6, 1, generator=gen) * 2).requires_grad_(True)
    ones, zeros = torch.ones(6, 1), torch.zeros(6, 1)
    bce = F.binary_cross_entropy_with_logits
    ld = 0.5 * (bce(pr - pf.mean(), ones) + bce(pf - pr.mean(), zeros))
    lg = 0.5 * (OS.focal_loss(pr - pf.mean(), zeros, 1.0) + OS.focal_loss(pf - pr.mean(), ones, 1.0))
    (ld + 3.0 * lg).backward()
    assert torch.allclose(losses, torch.stack([ld.detach(), lg.detach()]), rtol=1e-6, atol=1e-7)
    assert torch.allclose(g_pr, pr.grad, rtol=1e-5, atol=1e-8) and torch.allclose(g_pf, pf.grad, rtol=1e-5, atol=1e-8)


def _policy_worker(rank, world, port, q):
    """Two ranks pick the data-parallel schedule by measurement (Trainer.calibrate_dp_policy): rank 1 is made slow under the
    "overlap" schedule only, so the max-over-ranks times must make BOTH ranks choose "defer_g"."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import time
        from pesr_amd import comm
        from pesr_amd.optim import FlatParams, GradBuckets
        from pesr_amd.step import Trainer
        torch.set_num_threads(2)
        sd = gen_sd(16, 1)
        params = [torch.nn.Parameter(v.clone()) for v in sd.values()]
        leaves = dict(zip(sd.keys(), params))
        flat = FlatParams(params)
        buckets = GradBuckets(flat, bucket_bytes=8 << 10)
        assert isinstance(buckets.transport, comm.TorchGroup) and buckets.transport.backend == "gloo" and not buckets.transport.capturable
        nb = len(buckets.bounds)
        assert nb > 3

        class Opt:
            pass
        oG = Opt(); oG.buckets = buckets
        launches = {}

        class T(Trainer):
            def pretrain_step(self, lr, hr):
                flat.zero_grad(); buckets.reset()
                n0 = buckets.launches
                sr = OM.generator_forward(leaves, lr, 1, 0.1)
                F.l1_loss(sr, hr).backward()
                if rank == 1 and buckets.mode == "overlap":
                    time.sleep(0.05)                   # this rank is slow under this schedule only
                assert buckets.finish() == 1.0 / world
                launches.setdefault(buckets.mode, set()).add(buckets.launches - n0)
                return {"l1": torch.zeros(())}

        tr = T(None, optim_G=oG, world_size=world)
        B = 2
        lr = detrand.image_batch((B, 3, 8, 8), 11 + rank); hr = detrand.image_batch((B, 3, 32, 32), 12 + rank)
        info = tr.calibrate_dp_policy("pretrain", lambda: (lr, hr), steps=2)
        assert info["chosen"] == "defer_g" and tr.dp_policy == "defer_g" and buckets.mode == "deferred", info
        assert tr.dp_step == tr.pretrain_step and info["graph_error"] is None and "graph+defer_g" not in info["ms_per_step"]
        assert info["ms_per_step"]["overlap"] >= 50.0 > info["ms_per_step"]["defer_g"], info      # rank 0 sees rank 1's time
        assert launches == {"overlap": {nb}, "deferred": {1}}, launches                         # one collective when deferred
        # the deferred all-reduce moves the same gradients: both schedules give the same averaged gradient
        tr.set_dp_policy("overlap"); tr.pretrain_step(lr, hr); g_a = flat.flat_g.clone()
        tr.set_dp_policy("defer_g"); tr.pretrain_step(lr, hr); g_b = flat.flat_g.clone()
        assert torch.allclose(g_a, g_b, rtol=1e-6, atol=1e-7 * float(g_a.abs().max()))
        q.put((rank, info["chosen"], info["ms_per_step"]))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_dp_policy_selection_agrees():
    world = 2
    ctx = mp.get_context("spawn")
    got = None
    for attempt in range(3):
        port, q = _free_port(), ctx.Queue()
        procs = [ctx.Process(target=_policy_worker, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        try:
            got = [q.get(timeout=150) for _ in range(world)]
        except Exception:
            got = None
        for p in procs:
            p.join(60)
            if p.is_alive():
                p.kill()
        if got is not None and all(p.exitcode == 0 for p in procs):
            break
        got = None
    assert got is not None, "the 2-rank gloo job failed three times"
    assert {g[1] for g in got} == {"defer_g"} and got[0][2] == got[1][2], got      # same choice from the same agreed numbers


def _probe_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pesr_amd import comm
        # the probe itself between two ranks: rank 0's port reaches both, each rank's child starts, finds no GPU here, is reported
        ok, why = comm.probe_direct(torch.device("cpu"))
        assert not ok and "no GPU visible" in why, why
        # make_transport's "auto" with a probe that fails on ONE rank only: both ranks must take torch.distributed (MIN all-reduce)
        # before any of them would have entered ncclCommInitRank (DirectRccl is made to raise if anybody gets there)
        orig_backend = dist.get_backend
        comm.dist.get_backend = lambda group=None: "nccl"
        comm.probe_direct = lambda device, group=None, timeout=None: ((False, "rank 1's child hung") if rank == 1 else (True, ""))

        class _Never:
            def __init__(self, *a, **k):
                raise AssertionError("the communicator was created although a rank's probe failed")
        comm.DirectRccl = _Never
        fake_cuda = type("D", (), {"type": "cuda", "index": 0})()
        real_tensor = torch.tensor
        torch.tensor = lambda *a, **k: real_tensor(*a, **{**k, "device": "cpu"}) if "device" in k else real_tensor(*a, **k)
        try:
            tr = comm.make_transport(fake_cuda, prefer="auto")
        finally:
            torch.tensor = real_tensor
            comm.dist.get_backend = orig_backend
        assert isinstance(tr, comm.TorchGroup) and tr.fallback_reason.startswith("probe: ")
        q.put((rank, tr.fallback_reason))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_direct_transport_probe_falls_back_together():
    """Guard of the never-executed world > 1 branch of the direct RCCL transport (pesr_amd/comm.py `probe_direct`)."""
    world = 2
    ctx = mp.get_context("spawn")
    port, q = _free_port(), ctx.Queue()
    procs = [ctx.Process(target=_probe_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=200) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert got[0][1] == "probe: another rank's probe failed" and got[1][1] == "probe: rank 1's child hung"
