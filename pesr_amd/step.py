"""The two train steps of the reference (train.py:164-173 pretrain, :194-259 GAN), device-side.

Same loss expressions, same order of forward passes (one G forward, four D forwards with their own
BatchNorm batch statistics, two VGG passes), same requires_grad toggling and optimizer steps.  Differences
that do not change results: losses stay on the device (one host sync per log interval instead of five per
step, train.py:262-266); gradients are all-reduced across ranks inside the optimizers (optim.py) instead of
through nn.DataParallel's gather / reduce_add.
Data-parallel loss scaling: every mean-type loss is the local mean and gradients are averaged
over ranks - identical to the reference's global mean for equal shards - but the TV term is a SUM over the
global batch (train.py:137-140), so its local value is multiplied by world_size before the averaging.
"""
from __future__ import annotations

import os

import torch
import torch.nn.functional as F

from . import functional as PF
from .model.basic import nhwc
from .model.focal_loss import FocalLoss
from .model.pesr import Discriminator


class _GlobalBatchMean(torch.autograd.Function):
    """mean of a [B, 1] tensor over the GLOBAL batch of a data-parallel run (every rank holds B samples), differentiable.
    The reference computes its losses on the gathered full batch (nn.DataParallel, train.py:114-118), so a loss term that
    contains a batch mean - RaSGAN's mean(C(x)) - needs it over all ranks.  forward: all-reduce of the local sums;
    backward: every rank's loss depends on the mean, and gradients are AVERAGED over ranks afterwards (optim.GradBuckets), so
    a rank's samples receive sum_over_ranks(dL_r/dm) / (N B)."""

    @staticmethod
    def forward(ctx, x, world):
        import torch.distributed as dist
        ctx.world, ctx.n = world, x.numel()
        s = x.sum()
        if world > 1:
            dist.all_reduce(s)
        return s / (world * x.numel())

    @staticmethod
    def backward(ctx, g):
        import torch.distributed as dist
        g = g.clone()
        if ctx.world > 1:
            dist.all_reduce(g)
        return (g / (ctx.world * ctx.n)).expand(ctx.n, 1), None


def batch_mean(x, world=1):
    return x.mean() if world == 1 else _GlobalBatchMean.apply(x, world)


def rasgan_d_loss(pred_real, pred_fake, target_real, target_fake, world=1):
    """Relativistic AVERAGE standard GAN, discriminator side (Jolicoeur-Martineau 2018, eq. 10 / the published RaSGAN code):
    [BCE(C(x_r) - mean C(x_f), 1) + BCE(C(x_f) - mean C(x_r), 0)] / 2.  An extension: the reference implements SGAN and RSGAN
    only (train.py:210-213), while the task statement names the relativistic-average form."""
    return 0.5 * (F.binary_cross_entropy_with_logits(pred_real - batch_mean(pred_fake, world), target_real) +
                  F.binary_cross_entropy_with_logits(pred_fake - batch_mean(pred_real, world), target_fake))


def rasgan_g_loss(pred_real, pred_fake, target_real, target_fake, loss_fn, world=1):
    """Generator side: the same with the labels swapped; loss_fn = BCE-with-logits or the FocalLoss module, as the reference
    chooses for its other GAN types (train.py:244-253)."""
    return 0.5 * (loss_fn(pred_real - batch_mean(pred_fake, world), target_fake) +
                  loss_fn(pred_fake - batch_mean(pred_real, world), target_real))


def _img(x: torch.Tensor) -> torch.Tensor:
    """logical NCHW image batch -> NHWC-contiguous view/copy for the loss kernels"""
    return nhwc(x).contiguous()


class Trainer:
    def __init__(self, G, D=None, vgg=None, optim_G=None, optim_D=None, *, gan_type="RSGAN", focal_loss=True, fl_gamma=1.0,
                 alpha_vgg=50.0, alpha_gan=1.0, alpha_tv=1e-6, alpha_l1=0.0, world_size=1, gradient_penalty=False):
        self.G, self.D, self.vgg = G, D, vgg
        self.optim_G, self.optim_D = optim_G, optim_D
        self.gan_type, self.use_focal = gan_type, focal_loss
        self.pair_classifier = os.environ.get("PESR_PAIR_CLASSIFIER", "1") != "0"   # D's classifier once per phase on [hr; sr] (gan_step)
        self.f_loss_fn = FocalLoss(fl_gamma)
        self.alpha_vgg, self.alpha_gan, self.alpha_tv, self.alpha_l1 = alpha_vgg, alpha_gan, alpha_tv, alpha_l1
        self.world_size = world_size
        self.gradient_penalty = gradient_penalty       # reference --GP (train.py:216-226), default off
        from . import ops
        if gradient_penalty and ops.PRECISION != "fp32":
            raise ValueError("the gradient penalty (--GP true) is only built for the fp32 arithmetic: the optional bf16 mode has "
                             "no second-order oracle (use --precision fp32)")
        self._targets = {}
        self._graph = None
        self.dp_policy, self.dp_step = "overlap", None

    def _target(self, batch, value, device):
        key = (batch, value, device)
        if key not in self._targets:
            self._targets[key] = torch.full((batch, 1), float(value), device=device)
        return self._targets[key]

    # ---- reference train.py:164-173 -----------------------------------------------------------------
    def pretrain_step(self, lr, hr):
        sr = self.G(lr)
        self.optim_G.zero_grad()
        loss = PF.l1_loss(nhwc(sr), _img(hr))
        loss.backward()
        self.optim_G.step()
        return {"l1": loss.detach()}

    # ---- reference train.py:194-259 -----------------------------------------------------------------
    def _gradient_penalty(self, hr_cl, sr, u):
        """10 * mean((||dD(x_both)/dx_both||_2 - 1)^2), x_both = hr*u + sr*(1-u) a NEW leaf (reference train.py:216-226).  The
        interpolation, the three norms and the square are torch ops on [B,3,H,W] / [B] tensors; everything inside D and its
        first- and second-order backward runs on the HIP kernels (Discriminator.forward_second_order)."""
        B = hr_cl.size(0)
        if u is None:
            u = torch.rand(B, 1, 1, 1, device=hr_cl.device)
        x_both = (hr_cl * u + sr.detach() * (1 - u)).detach().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        out = self.D.forward_second_order(x_both)
        grad = torch.autograd.grad(outputs=out, inputs=x_both, grad_outputs=torch.ones_like(out), retain_graph=True,
                                   create_graph=True, only_inputs=True)[0]
        return 10 * ((grad.norm(2, 1).norm(2, 1).norm(2, 1) - 1) ** 2).mean()

    def gan_step(self, lr, hr, gp_u=None):
        G, D = self.G, self.D
        B, dev = lr.size(0), lr.device
        target_real, target_fake = self._target(B, 1.0, dev), self._target(B, 0.0, dev)
        hr_cl = hr.contiguous(memory_format=torch.channels_last)

        # discriminator phase: hr real, sr fake
        for p in D.parameters():
            p.requires_grad = True
        self.optim_D.zero_grad()
        # The two calls of D keep their own BatchNorm statistics (two passes over the blocks, in the reference's order); the classifier
        # behind them has none, so it runs ONCE on [hr; sr] (Discriminator.classify_pair): one pass over classifier.0's 302 MB instead of
        # two - forward, input gradient and weight gradient - with results bit for bit those of the separate calls
        # (tests/test_model_gpu.py).  (A wrapped D - nn.DataParallel replicas - and batches above 16 take the plain path.)
        pair = self.pair_classifier and isinstance(D, Discriminator) and B <= Discriminator.PAIR_ROWS
        if pair:
            f_real = D.forward_features(hr_cl)
            sr = G(lr)
            pred_real, pred_fake = D.classify_pair(f_real, D.forward_features(sr.detach()))
        else:
            pred_real = D(hr_cl)
            sr = G(lr)
            pred_fake = D(sr.detach())
        if self.gan_type not in ("SGAN", "RSGAN", "RaSGAN"):
            raise ValueError(f"unknown gan_type {self.gan_type}")
        fused = pred_real.is_cuda and B <= 1024 and not (self.gan_type == "RaSGAN" and self.world_size > 1)
        if fused:        # value and both gradients in one kernel (ops.gan_loss) instead of ~25 scalar-sized torch kernels
            total_D_loss = PF.gan_loss(pred_real, pred_fake, self.gan_type, 0)
        elif self.gan_type == "SGAN":
            total_D_loss = F.binary_cross_entropy_with_logits(pred_real, target_real) + \
                F.binary_cross_entropy_with_logits(pred_fake, target_fake)
        elif self.gan_type == "RSGAN":
            total_D_loss = F.binary_cross_entropy_with_logits(pred_real - pred_fake, target_real)
        else:            # RaSGAN under data parallelism: the batch means span the ranks (differentiable all-reduce)
            total_D_loss = rasgan_d_loss(pred_real, pred_fake, target_real, target_fake, self.world_size)
        gp = None
        if self.gradient_penalty:
            gp = self._gradient_penalty(hr_cl, sr, gp_u)
            total_D_loss = total_D_loss + gp
        # with the penalty, classifier.{0,2}.weight get a third contribution (functional.INPLACE_SECOND_USE)
        PF.INPLACE_SECOND_USE = not self.gradient_penalty
        try:
            total_D_loss.backward()
        finally:
            PF.INPLACE_SECOND_USE = True

        # The generator-phase terms that need neither the updated D nor its gradients (reference train.py:238-242) run
        # BEFORE D's optimizer step: on N > 1 GPUs these ~9 ms of VGG kernels cover the all-reduce of D's 321 MB of
        # gradients that optim_D.step() has to wait for.  Same graph, same values; only the launch order differs.
        sr_nhwc, hr_nhwc = nhwc(sr), nhwc(hr_cl)
        l1_loss = PF.l1_loss(sr_nhwc, hr_nhwc) * self.alpha_l1
        vgg_sr, vgg_hr = self.vgg(sr, hr_cl)
        vgg_loss = PF.mse_loss(nhwc(vgg_sr), nhwc(vgg_hr)) * self.alpha_vgg
        tv_local = PF.tv_loss(sr_nhwc) * self.alpha_tv
        self.optim_D.step()

        # generator phase
        for p in D.parameters():
            p.requires_grad = False
        self.optim_G.zero_grad()
        if pair:
            f_fake = D.forward_features(sr)
            with torch.no_grad():      # D's parameters are frozen and hr needs no grad: a pure forward, as in the reference
                f_real = D.forward_features(hr_cl)
            pred_fake, pred_real = D.classify_pair(f_fake, f_real, grad_first_only=True)
        else:
            pred_fake = D(sr)
            with torch.no_grad():      # D's parameters are frozen and hr needs no grad: a pure forward, as in the reference
                pred_real = D(hr_cl)
        loss_fn = self.f_loss_fn if self.use_focal else F.binary_cross_entropy_with_logits
        if fused:
            G_loss = PF.gan_loss(pred_real, pred_fake, self.gan_type, 1, self.use_focal, self.f_loss_fn.gamma, self.alpha_gan)
        else:
            if self.gan_type == "RaSGAN":
                G_loss = rasgan_g_loss(pred_real, pred_fake, target_real, target_fake, loss_fn, self.world_size)
            else:
                G_loss = loss_fn(pred_fake if self.gan_type == "SGAN" else pred_fake - pred_real, target_real)
            G_loss = G_loss * self.alpha_gan
        total_G_loss = l1_loss + vgg_loss + G_loss + tv_local * float(self.world_size)
        total_G_loss.backward()
        self.optim_G.step()
        logs = {"l1": l1_loss.detach(), "vgg": vgg_loss.detach(), "g": G_loss.detach(), "tv": tv_local.detach(),
                "d": total_D_loss.detach()}
        if gp is not None:
            logs["gp"] = gp.detach()
        return logs

    # ---- the GAN step as ONE hipGraph ------------------------------------------------------------------------------------
    def capture_gan_step(self, lr, hr):
        """Capture gan_step(lr, hr) - ~1000 kernel launches - into a hipGraph (torch.cuda.CUDAGraph on ROCm) and return
        gan_step_graphed.  Call it after at least TWO eager gan_steps at the same shapes: first-use work - weight packing,
        workspace growth, the re-pack descriptor tables (the second step's table lists the packings the first step created late,
        in its backward pass) - must not happen under capture.  The capture itself executes nothing.

        What makes the step replayable: every kernel is launched on torch's current stream through the C ABI; the losses stay
        on the device; the two Adam steps read the learning rate and step count from device memory
        (FlatAdam.use_device_state); the packed conv weights are refreshed by launches that are part of the captured sequence.
        Data-parallel runs capture too: the bucketed all-reduces that the autograd hooks launch (optim.GradBuckets) go to the
        transport's communication stream (pesr_amd/comm.py), which forks from the capturing stream at each launch and joins it
        again in FlatAdam.step's wait - they become nodes of the same graph, still concurrent with the backward kernels between
        fork and join.  Every rank must capture (and later replay) the same sequence."""
        if self.gradient_penalty:
            # the penalty's interpolation weights (reference train.py:217, one uniform draw per sample) become a graph INPUT:
            # gan_step_graphed draws them with torch.rand before every replay, or takes the caller's gp_u
            u = torch.rand(lr.size(0), 1, 1, 1, device=lr.device)
            return self._capture("gan", lambda a, b: self.gan_step(a, b, gp_u=u), (self.optim_D, self.optim_G), lr, hr, gp_u=u)
        return self._capture("gan", self.gan_step, (self.optim_D, self.optim_G), lr, hr)

    def capture_pretrain_step(self, lr, hr):
        """The same for pretrain_step (reference train.py:164-173): returns pretrain_step_graphed."""
        return self._capture("pretrain", self.pretrain_step, (self.optim_G,), lr, hr)

    def _capture(self, kind, fn, optims, lr, hr, gp_u=None):
        from . import ops
        for t in self._transports(optims):
            t.begin_capture()       # (the torch.distributed transport refuses: only the direct RCCL transport is capturable)
        for o in optims:
            o.use_device_state()
        st = {"lr": lr.clone(), "hr": hr.clone(), "optims": optims, "gp_u": gp_u}
        steps = [o.steps for o in optims]
        watched, ops.KERNEL_EVENTS.shape = ops.KERNEL_EVENTS.shape, None       # no timing events inside a graph
        torch.cuda.synchronize()
        st["graph"] = torch.cuda.CUDAGraph()
        # The captured repack launches hold raw pointers of their descriptor tables and packed buffers: the record keeps those
        # objects alive for as long as the graph lives, and tells _replay which packings a replay refreshes.
        PF._REPACK_RECORD = st["repacks"] = []
        try:
            with torch.cuda.graph(st["graph"], capture_error_mode="thread_local"):
                st["logs"] = fn(st["lr"], st["hr"])
        finally:
            PF._REPACK_RECORD = None
            ops.KERNEL_EVENTS.shape = watched
            for o, n in zip(optims, steps):
                o.steps = n                                                     # nothing ran: the host count must not move
        if self._graph is None:
            self._graph = {}
        self._graph[kind] = st
        return self.gan_step_graphed if kind == "gan" else self.pretrain_step_graphed

    def _transports(self, optims=None):
        """The distinct gradient-exchange transports (pesr_amd/comm.py) of the given optimizers' buckets."""
        seen = []
        for o in (optims if optims is not None else [o for o in (self.optim_D, self.optim_G) if o is not None]):
            t = o.buckets.transport if o.buckets.enabled else None
            if t is not None and all(t is not u for u in seen):
                seen.append(t)
        return seen

    # ---- data-parallel schedule, chosen by measurement -------------------------------------------------------------------
    DP_POLICIES = {"overlap": ("overlap", "overlap"), "defer_g": ("deferred", "overlap"), "defer_all": ("deferred", "deferred")}

    def set_dp_policy(self, name: str) -> None:
        """Bucket policies of (G, D): "overlap" - both exchanges ride under the backward passes (buckets launched from the
        autograd hooks); "defer_g" - G's 172 MB go as ONE all-reduce after its backward pass, whose 256-workgroup body kernels
        each pay a second round for any CU a collective holds (profiles/r03_cu_contention.txt), D's stay under D's backward and
        the many-workgroup VGG kernels; "defer_all" - both deferred."""
        g_mode, d_mode = self.DP_POLICIES[name]
        if self.optim_G is not None:
            self.optim_G.buckets.set_mode(g_mode)
        if self.optim_D is not None:
            self.optim_D.buckets.set_mode(d_mode)
        self.dp_policy = name

    def set_dp_transport(self, tr) -> None:
        """Every optimizer's gradient exchange moves to transport `tr` (between steps)."""
        for o in (self.optim_D, self.optim_G):
            if o is not None and o.buckets.enabled:
                assert not any(o.buckets._launched)
                o.buckets.transport = tr

    @staticmethod
    def calibration_batches(kind, steps=3, graph=True, candidates=None, peer_candidate=False):
        """Worst-case number of batches calibrate_dp_policy draws from next_batch() (train.py's guard against running off the end
        of an epoch inside the calibration): every eager candidate 1 + steps, the peer-memory candidate the same, the capture one
        batch and its replays 1 + steps."""
        n = len(candidates) if candidates else (3 if kind == "gan" else 2)
        return (n + (1 if peer_candidate else 0)) * (steps + 1) + ((1 + steps + 1) if graph else 0)

    def calibrate_dp_policy(self, kind, next_batch, steps=3, graph=True, candidates=None, margin=0.015, peer_candidate=False,
                            will_capture=False):
        """Pick the data-parallel schedule by timing it (call it inside the warm-up, on every rank, after at least two eager
        steps).  Each eager candidate of DP_POLICIES runs one untimed step (the policy switch) and `steps` timed ones; then,
        if the transport is capturable, the step is captured as a hipGraph with the best eager bucket policy and its replays
        are timed the same way.  peer_candidate (more than one rank, each with its own GPU): the CU-free exchange over peer memory
        (comm.PeerCopy) is first rehearsed in child processes (comm.probe_direct kind "peer": a hang there costs killed children,
        not the job), then timed under the "overlap" schedule as `overlap@peer-copy`; it replaces the transport only if it wins.
        will_capture: the caller is going to capture the step whatever comes out (train.py --hip_graph true): the peer-memory
        candidate - not capturable - is then not timed at all.
        Every rank uses the SLOWEST rank's time per candidate (Transport.host_max), so all ranks
        choose alike.  The first candidate ("overlap") stays unless another one is faster by more than `margin` (1.5 %): the
        bucket schedule decides RCCL's reduction order, i.e. the last bits of a run, and that must not hang on a 0.1 % timing
        race between equally good schedules (a graph replay keeps its eager policy's order: plain "faster" decides).
        ONE wall-clock budget (comm.bringup_budget(), env PESR_DP_BRINGUP_BUDGET, default 300 s from the first transport of the
        process: probes, self-tests and this calibration together): before every candidate the ranks agree on the time spent (MAX
        over ranks); once it exceeds the budget the remaining candidates are skipped, `overlap` on the transport that is up is
        used, and `fallback_reason` says so.  Returns {"chosen", "ms_per_step": {candidate: ms}, "graph_error", "transport",
        "fallback_reason", "bringup"}; afterwards `self.dp_step` is the step function to call (eager method or graph replay).
        These are real optimizer steps on real batches - nothing is thrown away."""
        import time

        from . import comm
        eager = self.gan_step if kind == "gan" else self.pretrain_step
        trs = self._transports((self.optim_D, self.optim_G) if kind == "gan" else (self.optim_G,))
        assert len(trs) == 1, "calibrate_dp_policy: the optimizers must share one enabled transport"
        tr = trs[0]
        cands = list(candidates) if candidates else (["overlap", "defer_g", "defer_all"] if kind == "gan" else ["overlap", "defer_g"])
        t_cal = time.monotonic()
        fallback_reason = getattr(tr, "fallback_reason", None)

        def sync():
            if torch.cuda.is_available():              # (the world-size-2 gloo test of this selection logic runs on the CPU)
                torch.cuda.synchronize()

        def timed(fn):
            fn(*next_batch())                          # untimed: first step under the new schedule
            tr.host_max([0.0])                         # aligns the ranks (and synchronises the device)
            sync()
            t0 = time.perf_counter()
            for _ in range(steps):
                fn(*next_batch())
            sync()
            return 1e3 * (time.perf_counter() - t0) / steps

        def budget_left():
            """Agreed over the ranks: the budget minus the LARGEST time any rank has spent."""
            (spent,) = tr.host_max([comm.bringup_spent()])
            return comm.bringup_budget() - spent

        mine, timed_cands = [], []
        for c in cands:
            if timed_cands and budget_left() <= 0.0:
                fallback_reason = ((fallback_reason + "; ") if fallback_reason else "") + \
                    f"bring-up budget of {comm.bringup_budget():.0f} s spent after candidates {timed_cands}: the rest skipped, overlap kept"
                break
            self.set_dp_policy(c)
            mine.append(timed(eager))
            timed_cands.append(c)
        budget_hit = len(timed_cands) < len(cands)
        agreed = tr.host_max(mine)
        ms = dict(zip(timed_cands, agreed))
        if budget_hit:
            best = cands[0]                            # (not a measured choice any more: the safe default)
        else:
            faster = [c for c in cands[1:] if ms[c] < ms[cands[0]] * (1.0 - margin)]
            best = min(faster, key=lambda c: (ms[c], cands.index(c))) if faster else cands[0]
        self.set_dp_policy(best)
        self.dp_step, chosen, graph_error = eager, best, None
        peer_note = None
        if will_capture and peer_candidate:
            peer_candidate, peer_note = False, "off: the caller captures the step (--hip_graph true) and the peer-memory exchange is not capturable"
        if peer_candidate and tr.world > 1 and tr.name != "peer-copy":
            dev = (self.optim_G or self.optim_D).flat.flat_g.device
            group = (self.optim_G or self.optim_D).buckets.group
            left = budget_left()
            if budget_hit or left < 60.0:
                peer_note = f"skipped: {max(left, 0.0):.0f} s of the bring-up budget left"
            else:
                # (a candidate, not the default: its rehearsal may cost the run 90 s at most - env PESR_DP_PEER_PROBE_TIMEOUT - and
                # never more than the bring-up budget leaves)
                ok_here, why = comm.probe_direct(dev, group, kind="peer", timeout=min(float(os.environ.get("PESR_DP_PEER_PROBE_TIMEOUT", "90")), left - 30.0))
                (bad,) = tr.host_max([0.0 if ok_here else 1.0])
                if bad:
                    peer_note = "probe: " + (why or "another rank's probe failed")
                else:
                    ptr_ = None
                    t_p = time.monotonic()
                    try:
                        ptr_ = comm.PeerCopy(dev, comm.dist.get_rank(group), tr.world, group)
                    except Exception as e:                   # (PeerCopy's constructor raises on every rank together: its failures are agreed)
                        peer_note = f"{type(e).__name__}: {e}"
                    comm.bringup_note("peer-copy transport + self-test", time.monotonic() - t_p, ptr_ is not None, peer_note or "")
                    (bad,) = tr.host_max([0.0 if ptr_ is not None else 1.0])      # (belt and braces: one more agreement on the old transport)
                    if bad and ptr_ is not None:
                        ptr_.close(collective=False)
                        ptr_, peer_note = None, "the peer-memory transport did not come up on another rank"
                    if ptr_ is not None:
                        self.set_dp_transport(ptr_)
                        self.set_dp_policy("overlap")
                        try:
                            (t_peer,) = tr.host_max([timed(eager)])
                        except comm.CommError as e:          # (a placement mismatch found at a bucket's first use: raised on every rank)
                            t_peer, peer_note = None, f"CommError: {e}"
                        if t_peer is not None:
                            ms["overlap@peer-copy"] = t_peer
                        if t_peer is not None and t_peer < ms[best] * (1.0 - margin):
                            chosen, tr = "overlap@peer-copy", ptr_
                            comm.adopt_transport(ptr_)       # (closed with the others at shutdown)
                            graph = False                    # (not capturable)
                        else:
                            self.set_dp_transport(tr)
                            self.set_dp_policy(best)
                            ptr_.close()
        if graph and tr.capturable and not budget_hit and budget_left() > 20.0:
            fn, err = None, None
            try:
                lr, hr = next_batch()
                fn = self.capture_gan_step(lr, hr) if kind == "gan" else self.capture_pretrain_step(lr, hr)
            except Exception as e:                      # (decided together below: every rank still takes part in host_max)
                err = f"{type(e).__name__}: {e}"
            # Agree on the capture BEFORE anybody replays: a replay runs the captured all-reduces, and a rank whose capture failed
            # would not be there to take part in them.
            (bad,) = tr.host_max([0.0 if err is None else 1.0])
            if bad:
                graph_error = err or "the capture failed on another rank"
                if self._graph:
                    self._graph.pop(kind, None)
            else:
                (t_all,) = tr.host_max([timed(fn)])
                ms["graph+" + best] = t_all
                if t_all < ms[best]:
                    self.dp_step, chosen = fn, "graph+" + best
                else:
                    self._graph.pop(kind, None)         # frees the graph's private memory pool
        elif graph and tr.capturable:
            graph_error = "not attempted: bring-up budget spent"
        comm.bringup_note("calibrate_dp_policy", time.monotonic() - t_cal, True, chosen)
        return {"chosen": chosen, "ms_per_step": {k: round(v, 3) for k, v in ms.items()}, "graph_error": graph_error,
                "transport": tr.name, "steps_per_candidate": steps, "margin": margin,
                "peer_candidate": peer_note or ("timed" if "overlap@peer-copy" in ms else "off"),
                "fallback_reason": fallback_reason, "bringup": comm.bringup_log()}

    def _replay(self, kind, lr, hr, gp_u=None):
        assert self._graph and kind in self._graph, f"capture_{kind}_step first"
        st = self._graph[kind]
        for o in st["optims"]:
            o.sync_lr_to_device()
        st["lr"].copy_(lr, non_blocking=True)
        st["hr"].copy_(hr, non_blocking=True)
        if st["gp_u"] is not None:
            if gp_u is None:
                st["gp_u"].uniform_()
            else:
                st["gp_u"].copy_(gp_u, non_blocking=True)
        st["graph"].replay()
        for o in st["optims"]:
            o.steps += 1
            # The replayed Adam kernels rewrote the parameters through raw pointers: every packed layout of these weights is
            # stale now (the epoch is part of the cache keys) ...
            PF.bump_weight_epoch(o.flat.params)
        for jobs, _table, _bufs in st["repacks"]:
            PF.stamp_repacked(jobs)      # ... except the ones the replayed repack launches have just rebuilt.  A packing made
            #                              later for another shape (validation on full images) misses its key and is rebuilt.
        return st["logs"]

    def gan_step_graphed(self, lr, hr, gp_u=None):
        """Replay the captured step on a new batch (same shapes).  Returns the same dict of device scalars as gan_step; they are
        overwritten by the next replay."""
        return self._replay("gan", lr, hr, gp_u)

    def pretrain_step_graphed(self, lr, hr):
        return self._replay("pretrain", lr, hr)
