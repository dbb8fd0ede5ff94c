"""Flat-buffer Adam + bucketed gradient all-reduce (replaces reference train.py:123-128 optim.Adam and
the gradient reduce_add of nn.DataParallel, reference train.py:114-118).

All parameters of a network are re-pointed into ONE contiguous fp32 buffer (and their .grad into a
second one), so the optimizer step is a single HBM-bound kernel over 28 B/parameter and the data-parallel
exchange is a handful of large RCCL all-reduces over xGMI instead of one message per tensor.
"""
from __future__ import annotations

from typing import Callable, List, Optional

import torch
import torch.distributed as dist

from . import functional as PF


def _align4(n: int) -> int:
    return (n + 3) // 4 * 4


class FlatParams:
    """Owns the flat parameter / gradient buffers of a list of nn.Parameters (device-agnostic)."""

    def __init__(self, params: List[torch.nn.Parameter]):
        self.params = [p for p in params]
        assert self.params, "no parameters"
        dev, dt = self.params[0].device, self.params[0].dtype
        self.offsets, off = [], 0
        for p in self.params:
            assert p.device == dev and p.dtype == dt
            self.offsets.append(off)
            off += _align4(p.numel())
        self.numel = off
        self.flat_p = torch.zeros(off, dtype=dt, device=dev)
        self.flat_g = torch.zeros(off, dtype=dt, device=dev)
        for p, o in zip(self.params, self.offsets):
            n = p.numel()
            self.flat_p[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.flat_p[o:o + n].view(p.shape)
            p.grad = None
        import weakref
        me = weakref.ref(self)      # the factories must not keep this object (and its buffers) alive through the parameters

        def factory(o, n, shape):
            def make():
                owner = me()
                return None if owner is None else owner.flat_g[o:o + n].view(shape)
            return make
        for p, o in zip(self.params, self.offsets):
            PF.register_grad_view(p, factory(o, p.numel(), p.shape))

    def attach_one(self, i: int) -> None:
        p, o = self.params[i], self.offsets[i]
        if p.grad is not None and p.grad.data_ptr() != self.flat_g.data_ptr() + o * self.flat_g.element_size():
            view = self.flat_g[o:o + p.numel()].view(p.shape)
            view.copy_(p.grad)
            p.grad = view

    def attach_grads(self) -> None:
        """Make the flat gradient buffer hold every parameter's gradient: a .grad that already is a view of its slice
        (the fast path of functional.grad_out, or autograd's in-place accumulation into it) needs nothing; a foreign
        tensor is copied in; a missing gradient leaves the slice at zero."""
        for i in range(len(self.params)):
            self.attach_one(i)

    def zero_grad(self) -> None:
        """Gradients are dropped (p.grad = None) and every flat slice may be claimed again.  The flat buffer itself is NOT
        zeroed here: the first gradient of a step overwrites its slice (functional.grad_out, or attach_one's copy), and
        finalize_grads() zeroes the slices of parameters that received none - zeroing 172 + 321 MB per step for slices that
        are about to be overwritten was two fill kernels of pure HBM traffic."""
        for p in self.params:
            p.grad = None
        PF.release_grad_views(self.params)

    def finalize_grads(self) -> None:
        """Before the optimizer kernel / all-reduce reads flat_g: foreign gradients are copied in, and a parameter that got no
        gradient in this step contributes zeros (torch-0.4 semantics of the reference: it still takes an Adam step)."""
        for i, p in enumerate(self.params):
            if p.grad is None:
                if not PF.grad_view_claimed(p):
                    o = self.offsets[i]
                    self.flat_g[o:o + p.numel()].zero_()
            else:
                self.attach_one(i)


class GradBuckets:
    """Bucketed, backward-overlapped gradient all-reduce over a FlatParams.

    The flat gradient buffer is cut into contiguous buckets of ~bucket_bytes.  A post-accumulate-grad hook on
    every parameter counts its bucket down; when the last gradient of a bucket lands, that slice is
    all-reduced (SUM) asynchronously on the transport's communication stream (pesr_amd/comm.py: RCCL over xGMI
    on a GPU node - the direct communicator or ProcessGroupNCCL -, gloo in the CPU tests).  Parameters are laid
    out in registration order while backward produces gradients in reverse, so buckets complete from the tail
    of the buffer while the MFMA kernels of earlier layers still run.  `finish()` launches whatever is left,
    waits, and returns 1/world_size (the averaging factor that the caller folds into the optimizer kernel).

    `mode` is the bucket policy, switchable between steps (Trainer.calibrate_dp_policy picks it by measurement):
    "overlap" as above; "deferred": the hooks launch nothing and finish() sends the WHOLE flat buffer as one
    all-reduce - no collective is resident on a CU while the 256-workgroup conv kernels run (one held CU costs
    each of those a second round, profiles/r03_cu_contention.txt), at the price of exposing the transfer."""

    MODES = ("overlap", "deferred")

    def __init__(self, flat: FlatParams, group=None, bucket_bytes: int = 32 << 20, transport=None):
        self.flat, self.group = flat, group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        # PESR_FORCE_DP=1 runs the bucket / hook / all-reduce machinery even with a single rank (lets the RCCL
        # path be exercised on a 1-GPU box)
        import os
        self.enabled = self.world > 1 or (os.environ.get("PESR_FORCE_DP") == "1" and dist.is_available() and dist.is_initialized())
        self.mode = "overlap"
        self.transport = transport
        if self.enabled and self.transport is None:
            from . import comm
            self.transport = comm.get_transport(flat.flat_g.device, group)
        self.bounds, self.members = [], []
        start, acc, cur = 0, 0, []
        esz = flat.flat_g.element_size()
        for i, (p, o) in enumerate(zip(flat.params, flat.offsets)):
            cur.append(i)
            acc = o + _align4(p.numel()) - start
            if acc * esz >= bucket_bytes:
                self.bounds.append((start, start + acc)); self.members.append(cur)
                start, cur = start + acc, []
        if cur:
            self.bounds.append((start, flat.numel)); self.members.append(cur)
        self.bucket_of = {}
        for b, mem in enumerate(self.members):
            for i in mem:
                self.bucket_of[i] = b
        self._pending, self._launched, self._works = [], [], []
        self._hooks = []
        self.launches = 0            # all-reduce calls issued so far (tests count them)
        # measure_exposed = True: bracket finish()'s waits with HIP events on the compute stream; exposed_ms() then reports
        # how long that stream stood still for all-reduces that backward did not cover (bench.py's comm_exposed_ms)
        self.measure_exposed, self._exposed = False, []
        if self.enabled:
            for i, p in enumerate(flat.params):
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))
        self.reset()

    def set_mode(self, mode: str) -> None:
        if mode not in self.MODES:
            raise ValueError(f"bucket policy must be one of {self.MODES}, got {mode!r}")
        assert not any(self._launched), "the bucket policy changes between steps, not inside one"
        self.mode = mode

    def reset(self) -> None:
        self._pending = [sum(1 for i in mem if self.flat.params[i].requires_grad) for mem in self.members]
        self._launched = [False] * len(self.members)
        self._works = []

    def _make_hook(self, i: int) -> Callable:
        def hook(_p):
            self.flat.attach_one(i)          # a gradient that arrived as a foreign tensor is moved into the flat buffer first
            b = self.bucket_of[i]
            self._pending[b] -= 1
            if self._pending[b] == 0 and not self._launched[b] and self.mode == "overlap":
                self._launch(b)
        return hook

    def _all_reduce(self, lo: int, hi: int) -> None:
        if self.flat.flat_g.is_cuda:
            PF.join_side_stream(self.flat.flat_g.device)    # (no-op unless PESR_SIDE_STREAM routes weight gradients to the side stream)
        self.launches += 1
        self._works.append(self.transport.all_reduce_async(self.flat.flat_g[lo:hi]))

    def _launch(self, b: int) -> None:
        self._launched[b] = True
        self._all_reduce(*self.bounds[b])

    def finish(self) -> float:
        if not self.enabled:
            return 1.0
        if self.mode == "deferred" and not any(self._launched):
            self._launched = [True] * len(self.members)
            self._all_reduce(0, self.flat.numel)            # one collective over the whole buffer
        for b in range(len(self.members)):
            if not self._launched[b]:
                self._launch(b)
        timed = self.measure_exposed and self.flat.flat_g.is_cuda and not torch.cuda.is_current_stream_capturing()
        if timed:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        self.transport.wait(self._works)   # (RCCL: the compute stream waits for the communication stream; the host does not block)
        if timed:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self._exposed.append((e0, e1))
        self.reset()
        return 1.0 / self.world

    def exposed_ms(self):
        """-> (mean milliseconds per finish() the compute stream waited for unfinished all-reduces, samples); clears them."""
        ev, self._exposed = self._exposed, []
        if not ev:
            return 0.0, 0
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in ev) / len(ev), len(ev)


class FlatAdam(torch.optim.Optimizer):
    """torch.optim.Adam(betas, eps, no weight decay) semantics on a flat buffer, one fused HIP kernel per step.

    A torch.optim.Optimizer subclass so lr_scheduler.StepLR (reference train.py:127-128) drives
    param_groups[0]['lr'] unchanged.  zero_grad() zeroes (never sets to None): like the reference's torch 0.4,
    a parameter that received no gradient still takes a (zero-gradient) Adam step."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, process_group=None, bucket_bytes=32 << 20):
        params = [p for p in params]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        assert len(self.param_groups) == 1
        self.flat = FlatParams(self.param_groups[0]["params"])
        self.exp_avg = torch.zeros_like(self.flat.flat_p)
        self.exp_avg_sq = torch.zeros_like(self.flat.flat_p)
        self.buckets = GradBuckets(self.flat, process_group, bucket_bytes)
        self.steps = 0
        self.dev_state = None        # use_device_state(): lr and step count in device memory (hipGraph replay)
        self._dev_lr = None
        self.last_scale = 1.0        # the 1/world factor of the latest step (flat_g holds the SUM over ranks)
        PF.bump_weight_epoch(self.flat.params)

    def zero_grad(self, set_to_none: bool = False) -> None:  # noqa: ARG002 (kept for API compatibility)
        self.flat.zero_grad()
        self.buckets.reset()

    def use_device_state(self) -> torch.Tensor:
        """Move the step-dependent Adam scalars to device memory (include/pesr_hip.h pesr_adam_step_dev): from now on step()
        launches nothing whose kernel arguments change from step to step, so a captured step can be replayed.  The host keeps
        counting in self.steps (state_dict, logging); a replay that runs without step() must add to it itself."""
        if self.dev_state is None:
            st = torch.zeros(6, dtype=torch.float32, device=self.flat.flat_p.device)
            st.view(torch.int32)[4] = self.steps & 0x7fffffff
            st.view(torch.int32)[5] = self.steps >> 31
            assert self.steps < (1 << 31)
            self.dev_state = st
            self.sync_lr_to_device()
        return self.dev_state

    def sync_lr_to_device(self) -> None:
        """Write param_groups[0]['lr'] (StepLR) into the device state; a no-op while it is unchanged.  NOT capturable - call it
        between replays."""
        lr = float(self.param_groups[0]["lr"])
        if self.dev_state is not None and lr != self._dev_lr:
            self.dev_state[0:1].fill_(lr)
            self._dev_lr = lr

    @torch.no_grad()
    def step(self, closure: Optional[Callable] = None):
        from . import ops
        assert closure is None
        if self.flat.flat_g.is_cuda:
            PF.join_side_stream(self.flat.flat_g.device)
        self.flat.finalize_grads()
        scale = self.last_scale = self.buckets.finish()
        g = self.param_groups[0]
        self.steps += 1
        if self.dev_state is not None:
            if not torch.cuda.is_current_stream_capturing():
                self.sync_lr_to_device()
            ops.adam_step_dev(self.flat.flat_p, self.flat.flat_g, self.exp_avg, self.exp_avg_sq, self.dev_state, g["betas"][0],
                              g["betas"][1], g["eps"], scale)
        else:
            ops.adam_step(self.flat.flat_p, self.flat.flat_g, self.exp_avg, self.exp_avg_sq, g["lr"], g["betas"][0],
                          g["betas"][1], g["eps"], self.steps, scale)
        PF.bump_weight_epoch(self.flat.params)   # this optimizer's packed conv weights are now stale ...
        PF.repack_all(self.flat.params)          # ... and are refreshed here in one batched launch
