"""Autograd glue over the HIP ops: one torch.autograd.Function per fused block of the reference graph.

Tensors crossing these functions are NHWC-contiguous fp32 GPU tensors ([N, H, W, C]); the nn.Modules in
pesr_amd/model convert from/to the logical NCHW view at the network boundary only.  Nothing here does
arithmetic in torch except scalar-sized glue (gradient fan-in adds are autograd's own).
"""
from __future__ import annotations

import torch
from torch.autograd import Function

from . import ops

# Our fused Adam writes parameters through raw pointers and cannot bump Tensor._version, so every parameter it owns
# carries an explicit epoch (bumped by FlatAdam.step for ITS parameters only - the static VGG weights and the other
# network's weights keep their packed copies).  Part of every packed-weight cache key.  The epoch is an attribute of the
# parameter object itself: it lives and dies with the parameter (a registry keyed by id() would outlive it and hand a
# recycled id a dead optimizer's state).
_EPOCH_ATTR = "_pesr_epoch"
_GRAD_VIEW_ATTR = "_pesr_grad_view"


def weight_epoch(p) -> int:
    return getattr(p, _EPOCH_ATTR, 0)


def bump_weight_epoch(params) -> None:
    for p in params:
        setattr(p, _EPOCH_ATTR, getattr(p, _EPOCH_ATTR, 0) + 1)


# Flat-gradient fast path.  pesr_amd.optim.FlatParams registers, per parameter, a factory of fresh views into its flat
# gradient buffer.  When a backward produces a parameter's FIRST gradient of the step (param.grad is None) the kernel
# writes straight into such a view and returns it; autograd then adopts it as .grad without a copy or an add kernel
# (AccumulateGrad steals a uniquely-referenced, contiguous gradient).  The slice is handed out ONCE per step: a
# parameter used twice in one graph (the Discriminator sees hr and sr in the same backward) gets a temporary for its
# second contribution, which autograd's input buffer adds to the first.  Stored as [factory, claimed] on the parameter
# object (same lifetime argument as the epoch); the factory holds only a weak reference to its FlatParams.
def register_grad_view(param: torch.Tensor, factory) -> None:
    setattr(param, _GRAD_VIEW_ATTR, [factory, False])


def unregister_grad_view(param: torch.Tensor) -> None:
    if hasattr(param, _GRAD_VIEW_ATTR):
        delattr(param, _GRAD_VIEW_ATTR)


def release_grad_views(params) -> None:
    """New step (zero_grad): every registered slice may be claimed again."""
    for p in params:
        e = getattr(p, _GRAD_VIEW_ATTR, None)
        if e is not None:
            e[1] = False


def grad_out_again(param):
    """The parameter's flat-gradient slice AFTER it has been claimed in this step (by grad_out), for a kernel that can
    accumulate into it - the second use of a layer inside one backward pass; else None."""
    if param is None or not param.is_leaf:
        return None
    e = getattr(param, _GRAD_VIEW_ATTR, None)
    if e is None or not e[1]:
        return None
    return e[0]()


def grad_view_claimed(param) -> bool:
    e = getattr(param, _GRAD_VIEW_ATTR, None)
    return bool(e is not None and e[1])


def grad_out(param):
    """A fresh view of the parameter's flat-gradient slice if its first gradient may be written there, else None."""
    if param is None or not param.is_leaf or param.grad is not None:      # (replicas of nn.DataParallel are non-leaf)
        return None
    e = getattr(param, _GRAD_VIEW_ATTR, None)
    if e is None or e[1]:
        return None
    view = e[0]()
    if view is None:          # the owning FlatParams is gone
        return None
    e[1] = True
    return view


_ALL_PACKS = []   # weak registry of every PackedConvWeights (for the batched repack after an optimizer step)


class _PackSlot:
    """The packed layouts of one weight tensor ON ONE DEVICE."""
    __slots__ = ("fwd", "dgrad", "bias", "wfwd", "wdgrad", "w4fwd", "w4dgrad", "bfwd", "bdgrad", "sfwd", "sdgrad", "kf", "kd", "kb", "kwf",
                 "kwd", "k4f", "k4d", "kbf", "kbd", "ksf", "ksd", "wref", "bref")

    def __init__(self):
        self.fwd = self.dgrad = self.bias = self.wfwd = self.wdgrad = self.w4fwd = self.w4dgrad = self.bfwd = self.bdgrad = None
        self.sfwd = self.sdgrad = None
        self.kf = self.kd = self.kb = self.kwf = self.kwd = self.k4f = self.k4d = self.kbf = self.kbd = self.ksf = self.ksd = None
        self.wref = None            # weakref to the weight tensor this slot was last built from
        self.bref = None            # ... and to the bias tensor of its PixelShuffle-permuted copy


class PackedConvWeights:
    """Per-module cache of the kernel-side weight layouts (forward / dgrad packing, PS-permuted bias).

    One slot per device: nn.DataParallel's replicas are shallow copies that SHARE this object while their weights live
    on different GPUs and their forwards run on concurrent threads (SURVEY 8b "Threading") - every replica works on its
    own device's slot, and the lock makes the check-then-pack step atomic."""

    def __init__(self, ps: bool = False):
        import threading
        import weakref
        self.ps = ps
        self._slots = {}            # device index -> _PackSlot
        self._lock = threading.Lock()
        _ALL_PACKS.append(weakref.ref(self))

    def __deepcopy__(self, memo):   # copy.deepcopy(module): a fresh, empty cache (locks cannot be copied)
        return PackedConvWeights(self.ps)

    def __getstate__(self):         # pickling a module (torch.save(model)): drop the device buffers and the lock
        return {"ps": self.ps}

    def __setstate__(self, st):
        self.__init__(st["ps"])

    @staticmethod
    def _key(t: torch.Tensor):
        return (t.data_ptr(), t._version, weight_epoch(t))

    def _slot(self, w) -> _PackSlot:
        idx = w.device.index if w.is_cuda else -1
        sl = self._slots.get(idx)
        if sl is None:
            sl = self._slots[idx] = _PackSlot()
        if sl.wref is None or sl.wref() is not w:
            import weakref
            sl.wref = weakref.ref(w)
        return sl

    def _get(self, w, field, keyf, build):
        with self._lock:
            sl = self._slot(w)
            k = self._key(w)
            if getattr(sl, keyf) != k:
                setattr(sl, field, build())
                setattr(sl, keyf, k)
            return getattr(sl, field)

    def fwd(self, w: torch.Tensor) -> torch.Tensor:
        return self._get(w, "fwd", "kf", lambda: ops.pack_conv3x3(w.detach(), 0, self.ps))

    def dgrad(self, w: torch.Tensor) -> torch.Tensor:
        return self._get(w, "dgrad", "kd", lambda: ops.pack_conv3x3(w.detach(), 1, self.ps))

    def wino_fwd(self, w: torch.Tensor):
        return self._get(w, "wfwd", "kwf", lambda: ops.pack_conv3x3_wino(w.detach(), 0, self.ps))

    def wino_dgrad(self, w: torch.Tensor):
        return self._get(w, "wdgrad", "kwd", lambda: ops.pack_conv3x3_wino(w.detach(), 1, self.ps))

    def wino4_fwd(self, w: torch.Tensor):
        return self._get(w, "w4fwd", "k4f", lambda: ops.pack_conv3x3_wino4(w.detach(), 0, self.ps))

    def wino4_dgrad(self, w: torch.Tensor):
        return self._get(w, "w4dgrad", "k4d", lambda: ops.pack_conv3x3_wino4(w.detach(), 1, self.ps))

    def bf16_fwd(self, w: torch.Tensor):
        return self._get(w, "bfwd", "kbf", lambda: ops.pack_conv3x3_bf16(w.detach(), 0, self.ps))

    def bf16_dgrad(self, w: torch.Tensor):
        return self._get(w, "bdgrad", "kbd", lambda: ops.pack_conv3x3_bf16(w.detach(), 1, self.ps))

    def bf16x3_fwd(self, w: torch.Tensor):
        return self._get(w, "sfwd", "ksf", lambda: ops.pack_conv3x3_bf16x3(w.detach(), 0, self.ps))

    def bf16x3_dgrad(self, w: torch.Tensor):
        return self._get(w, "sdgrad", "ksd", lambda: ops.pack_conv3x3_bf16x3(w.detach(), 1, self.ps))

    def for_fwd(self, w: torch.Tensor, x_shape, stride: int = 1):
        """Packed weights for y = conv(x, w): Winograd F(4,3) where that kernel applies and fills the chip, else F(2,3)
        where THAT applies, else the direct packing.  (ops.PRECISION == "bf16": the bf16 kernel where IT applies, first.)"""
        N, H, W, Cin = x_shape
        assert tuple(w.shape[2:]) == (3, 3), f"PackedConvWeights: 3x3 kernels only, got a weight of shape {tuple(w.shape)}"
        if ops.bf16x3_eligible(N, H, W, Cin, w.shape[0], stride, ps_out=self.ps):
            return self.bf16x3_fwd(w)
        if ops.bf16_eligible(N, H, W, Cin, w.shape[0], stride, ps_out=self.ps):
            return self.bf16_fwd(w)
        if ops.wino4_eligible(N, H, W, Cin, w.shape[0], stride, ps_out=self.ps):
            return self.wino4_fwd(w)
        if ops.wino_eligible(N, H, W, Cin, w.shape[0], stride):
            return self.wino_fwd(w)
        return self.fwd(w)

    def for_dgrad(self, w: torch.Tensor, x_shape, stride: int = 1):
        """Packed weights for dx of y = conv(x, w) with x of NHWC shape x_shape."""
        N, H, W, Cin = x_shape
        assert tuple(w.shape[2:]) == (3, 3), f"PackedConvWeights: 3x3 kernels only, got a weight of shape {tuple(w.shape)}"
        if stride == 1 and ops.bf16x3_eligible(N, H, W, w.shape[0], Cin, 1, ps_in=self.ps):
            return self.bf16x3_dgrad(w)
        if stride == 1 and ops.bf16_eligible(N, H, W, w.shape[0], Cin, 1, ps_in=self.ps):
            return self.bf16_dgrad(w)
        if stride == 2 and not self.ps and ops.bf16_s2_dgrad_eligible(N, H, W, w.shape[0], Cin):
            return self.bf16_dgrad(w)
        if ops.wino4_eligible(N, H, W, w.shape[0], Cin, stride):
            return self.wino4_dgrad(w)
        if ops.wino_eligible(N, H, W, w.shape[0], Cin, stride):
            return self.wino_dgrad(w)
        return self.dgrad(w)

    def bias(self, b):
        if b is None or not self.ps:
            return None if b is None else b.detach()
        with self._lock:
            idx = b.device.index if b.is_cuda else -1
            sl = self._slots.get(idx)
            if sl is None:
                sl = self._slots[idx] = _PackSlot()
            k = self._key(b)
            if sl.kb != k:
                sl.bias = ops.pack_bias_ps(b.detach())
                sl.kb = k
            if sl.bref is None or sl.bref() is not b:
                import weakref
                sl.bref = weakref.ref(b)
            return sl.bias


# Optional side stream for the weight-gradient kernels (off by default): in a backward chain the dgrad kernels
# depend on each other but the wgrad kernels only consume their outputs, so with two streams the workgroups of a wgrad can
# fill the CUs that the previous dgrad frees during its tail.  It paid with the direct kernels; with the Winograd kernels
# the fork / join costs more than the tails it fills.
_SIDE = {}
# PESR_SIDE_STREAM: "0" everything on one stream, "1" every conv block's wgrad on the side stream, "g" only the blocks
# without BatchNorm (the Generator's), "d" only the conv+BN blocks (the Discriminator's)
SIDE_MODE = __import__("os").environ.get("PESR_SIDE_STREAM", "0")
USE_SIDE_STREAM = SIDE_MODE != "0"
_SIDE_OFF = SIDE_MODE == "0"     # (in-place second-use accumulation assumes every weight gradient is written on ONE stream)


def side_stream(device) -> "torch.cuda.Stream":
    s = _SIDE.get(device.index)
    if s is None:
        s = _SIDE[device.index] = torch.cuda.Stream(device=device)
    return s


def join_side_stream(device=None) -> None:
    """Make the current stream wait for everything queued on the side stream (optimizer step, gradient all-reduce)."""
    for idx, s in _SIDE.items():
        if device is None or device.index == idx:
            torch.cuda.current_stream(s.device).wait_stream(s)


_JOIN_QUEUED = [False]


def _join_after_backward():
    _JOIN_QUEUED[0] = False
    join_side_stream()


class _OnSide:
    """with _OnSide(dev, tensors...): the side stream first waits for the work queued so far on the current stream;
    the listed tensors (inputs read on the side stream) are protected from the caching allocator's early reuse.

    Stream safety of the RESULTS: a gradient written into a flat-gradient view (`fast`) is consumed only by code that
    joins the side stream first (FlatAdam.step, the all-reduce buckets, and a callback queued at the end of every
    backward pass).  Anything else - a temporary that autograd itself will add to another contribution on the main
    stream - makes the main stream wait for the side stream on exit (`fast=False`)."""

    def __init__(self, device, *tensors, fast=False, where="g"):
        self.enabled = SIDE_MODE == "1" or SIDE_MODE == where
        self.fast, self.device = fast, device
        if self.enabled:
            if not _JOIN_QUEUED[0]:
                try:
                    torch.autograd.Variable._execution_engine.queue_callback(_join_after_backward)
                    _JOIN_QUEUED[0] = True
                except RuntimeError:      # not inside a backward pass (direct call of a backward in a test)
                    self.fast = False
            self.side = side_stream(device)
            self.side.wait_stream(torch.cuda.current_stream(device))
            for t in tensors:
                if t is not None:
                    t.record_stream(self.side)
            self.ctx = torch.cuda.stream(self.side)

    def __enter__(self):
        if self.enabled:
            self.ctx.__enter__()

    def __exit__(self, *a):
        if self.enabled:
            self.ctx.__exit__(*a)
            if not self.fast:
                torch.cuda.current_stream(self.device).wait_stream(self.side)


_REPACK_TABLES = {}        # key -> device descriptor table, least recently used first (a dict keeps insertion order)
_REPACK_TABLES_MAX = 8
# Set (to a list) by Trainer._capture while a step is being captured into a hipGraph: every repack_all launch under capture
# appends (jobs, table).  The graph holds the table's and the packed buffers' RAW pointers, so the capture state keeps these
# objects alive (an evicted table, or a packed buffer an eager rebuild replaced, would otherwise go back to the caching
# allocator and the next replay would read / write reused memory), and Trainer._replay uses the job list to re-stamp exactly
# the packings the replayed launch refreshed.
_REPACK_RECORD = None

_SLOT_FIELD = {0: ("fwd", "kf"), 1: ("dgrad", "kd"), 2: ("wfwd", "kwf"), 3: ("wdgrad", "kwd"), 4: ("w4fwd", "k4f"), 5: ("w4dgrad", "k4d"),
               6: ("bias", "kb"), 7: ("bfwd", "kbf"), 8: ("bdgrad", "kbd"), 9: ("sfwd", "ksf"), 10: ("sdgrad", "ksd")}      # 6: the PixelShuffle-permuted bias (its source tensor is the slot's bref, not wref)


def _slot_buffer(sl, mode):
    b = getattr(sl, _SLOT_FIELD[mode][0])
    return b if (b is None or torch.is_tensor(b)) else b.t


def stamp_repacked(jobs) -> None:
    """Mark the packings of `jobs` (as recorded by repack_all) as built from their weights' CURRENT contents.  A packing whose
    buffer is no longer the recorded one (an eager rebuild replaced it) is left alone: its key then misses and it is rebuilt
    on next use."""
    for sl, wref, mode, ptr in jobs:
        w = wref()          # (mode 6: the bias tensor)
        buf = _slot_buffer(sl, mode)
        if w is None or buf is None or buf.data_ptr() != ptr:
            continue
        setattr(sl, _SLOT_FIELD[mode][1], PackedConvWeights._key(w))


def repack_all(params) -> None:
    """Refresh, in ONE kernel launch, every packed layout that already exists for the given parameters (called by
    FlatAdam.step right after its Adam kernel, instead of ~2 small pack launches per conv on the next forward/backward)."""
    import weakref

    import numpy as np
    ids = {id(p) for p in params}
    jobs = []
    for ref in list(_ALL_PACKS):
        c = ref()
        if c is None:
            _ALL_PACKS.remove(ref)
            continue
        for sl in list(c._slots.values()):
            w = sl.wref() if sl.wref is not None else None
            if w is None or id(w) not in ids or not w.is_cuda:
                continue
            O, I = w.shape[0], w.shape[1]
            for mode, buf in ((0, sl.fwd), (1, sl.dgrad)):
                if buf is None:
                    continue
                R, Nn = (I, O) if mode == 0 else (O, I)
                jobs.append((sl, w, mode, (w.data_ptr(), buf.data_ptr(), O, I, mode, int(c.ps), (R + 15) // 16 * 16, 16 if Nn <= 16 else (Nn + 63) // 64 * 64)))
            # Winograd packings (batched kernel modes 2 / 3: F(2,3), 4 / 5: F(4,3))
            for mode, wpk in ((2, sl.wfwd), (3, sl.wdgrad), (4, sl.w4fwd), (5, sl.w4dgrad), (7, sl.bfwd), (8, sl.bdgrad), (9, sl.sfwd),
                              (10, sl.sdgrad)):
                if wpk is not None:
                    fwd_like = mode in (2, 4, 7, 9)
                    jobs.append((sl, w, mode, (w.data_ptr(), wpk.t.data_ptr(), O, I, mode, int(c.ps), I if fwd_like else O, O if fwd_like else I)))
    # PixelShuffle-permuted biases of these parameters: refreshed in place too (two tiny launches for the Generator).  Left to the
    # lazy per-forward check they were re-packed on every forward - and a captured step whose capture happened to find the key
    # current (a forward between the last optimizer step and the capture) would never refresh them at all.
    bias_rec = []
    for ref in list(_ALL_PACKS):
        c = ref()
        if c is None:
            continue
        for sl in list(c._slots.values()):
            b = sl.bref() if sl.bref is not None else None
            if b is None or sl.bias is None or id(b) not in ids or not b.is_cuda:
                continue
            ops.pack_bias_ps(b.detach(), out=sl.bias)
            bias_rec.append((sl, sl.bref, 6, sl.bias.data_ptr()))
    if bias_rec:
        if _REPACK_RECORD is not None:
            _REPACK_RECORD.append((bias_rec, None, [sl.bias for sl, _, _, _ in bias_rec]))
        stamp_repacked(bias_rec)
    if not jobs:
        return
    dev = jobs[0][1].device
    key = (tuple(sorted(ids)), tuple(j[3] for j in jobs))
    # One descriptor table per (parameter set, pack layout), kept on the device: the H2D copy that builds it SYNCHRONISES the
    # host with the stream, so it must happen once per optimizer - not once per step (with one shared slot, the G and D
    # optimizers of the GAN step evicted each other's table every step and the host could never run ahead of the GPU).
    table = _REPACK_TABLES.pop(key, None)
    if table is None:
        while len(_REPACK_TABLES) >= _REPACK_TABLES_MAX:           # least recently used goes first
            _REPACK_TABLES.pop(next(iter(_REPACK_TABLES)))
        table = torch.from_numpy(np.array([j[3] for j in jobs], dtype=np.int64)).to(dev)
    _REPACK_TABLES[key] = table                                    # (re-)inserted as the most recently used
    from . import _lib
    _lib.check(_lib.lib().pesr_pack_conv3x3_batched(table.data_ptr(), len(jobs), torch.cuda.current_stream(dev).cuda_stream),
               "pesr_pack_conv3x3_batched")
    rec = [(sl, weakref.ref(w), mode, d[1]) for sl, w, mode, d in jobs]
    if _REPACK_RECORD is not None:
        # under capture nothing ran: the record (it also pins the table and the packed buffers) is used at replay time
        _REPACK_RECORD.append((rec, table, [_slot_buffer(sl, mode) for sl, _, mode, _ in jobs]))
    stamp_repacked(rec)


def _c(t: torch.Tensor) -> torch.Tensor:
    return t if t.is_contiguous() else t.contiguous()


# ------------------------------------------------------------------------------------------------
# generic 3x3 conv (+bias, +ReLU, fused PixelShuffle)          reference model/basic.py:4-7, 56-60
# ------------------------------------------------------------------------------------------------
class Conv3x3Fn(Function):
    """y = act(conv(x, w) + b) [pixel-shuffled].

    relu_in:  x is the output of a ReLU whose backward this function applies to its grad_input
              (mask by x > 0, fused in the dgrad epilogue).
    relu_grad_by_consumer: with act=relu, the consumer of y applies the ReLU mask (it set relu_in);
              otherwise this function masks grad_output itself.
    """

    @staticmethod
    def forward(ctx, x, weight, bias, cache: PackedConvWeights, stride, act, relu_in, relu_grad_by_consumer):
        x = _c(x)
        cout = weight.shape[0]
        y = ops.conv3x3_fwd(x, lambda: cache.for_fwd(weight, x.shape, stride), cache.bias(bias), cout, stride, act=act, ps_out=cache.ps,
                            w_oihw=weight.detach())
        ctx.cache, ctx.stride, ctx.act, ctx.relu_in = cache, stride, act, relu_in
        ctx.mask_here = act == ops.ACT_RELU and not relu_grad_by_consumer
        ctx.has_bias = bias is not None
        ctx.bias_ref = bias
        ctx.save_for_backward(x, weight, y if ctx.mask_here else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, y = ctx.saved_tensors
        gy = _c(gy)
        if ctx.mask_here:
            gy = ops.relu_mask(gy, y)
        cin, ps = x.shape[3], ctx.cache.ps
        dx = dw = db = None
        # dx of a C -> 3 conv is a 3 -> C conv of dy: the HBM-bound direct kernel, no packed weights
        rgb_dgrad = weight.shape[0] == 3 and ctx.stride == 1 and not ctx.relu_in and not ps and cin % 4 == 0 and 256 % (cin // 4) == 0
        # ... and dx of a 3 -> C conv is a C -> 3 conv of dy: the HBM-bound kernel of the C -> 3 forward, on the OIHW weights
        rgb_in_dgrad = not ctx.relu_in and not ps and ops.rgb_in_dgrad_eligible(cin, weight.shape[0], ctx.stride)
        wpd = ctx.cache.for_dgrad(weight, x.shape, ctx.stride) if ctx.needs_input_grad[0] and not (rgb_dgrad or rgb_in_dgrad) else None
        if ctx.needs_input_grad[1]:
            want_b = ctx.has_bias and ctx.needs_input_grad[2]
            outs = dict(dw_out=grad_out(weight), db_out=grad_out(ctx.bias_ref) if want_b else None)
            with _OnSide(gy.device, x, gy, fast=outs["dw_out"] is not None and (outs["db_out"] is not None or not want_b)):
                if cin == 3:
                    dw, db = ops.conv3x3_wgrad_rgb(gy, x, 0, want_bias=want_b, **outs)
                elif weight.shape[0] == 3:
                    dw, db = ops.conv3x3_wgrad_rgb(x, gy, 1, want_bias=want_b, **outs)
                else:
                    dw, db = ops.conv3x3_wgrad(x, gy, ctx.stride, want_bias=want_b, ps_in=ps, **outs)
        if ctx.needs_input_grad[0]:
            if rgb_dgrad:
                dx = ops.conv3x3_rgb_dgrad(gy, weight.detach(), tuple(x.shape))
            elif rgb_in_dgrad:
                dx = ops.conv3x3_rgb_in_dgrad(gy, weight.detach(), tuple(x.shape))
            else:
                dx = ops.conv3x3_dgrad(gy, wpd, tuple(x.shape), ctx.stride, mask=x if ctx.relu_in else None, ps_in=ps)
        return dx, dw, db, None, None, None, None, None


class ConvLReluFn(Function):
    """y = leaky_relu(conv(x, w) + b, slope): BasicBlock(bn=False, act=LeakyReLU) / ResBlock(act=LeakyReLU) - constructor
    branches of reference model/basic.py the reference's own scripts never take."""

    @staticmethod
    def forward(ctx, x, weight, bias, cache, stride, slope):
        x = _c(x)
        y = ops.conv3x3_fwd(x, lambda: cache.for_fwd(weight, x.shape, stride), cache.bias(bias), weight.shape[0], stride,
                            act=ops.ACT_LRELU, slope=slope, ps_out=cache.ps, w_oihw=weight.detach())
        ctx.cache, ctx.stride, ctx.slope, ctx.bias_ref = cache, stride, slope, bias
        ctx.save_for_backward(x, weight, y)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, y = ctx.saved_tensors
        gz = ops.relu_mask(_c(gy), y, slope=ctx.slope)          # sign(y) == sign(z) for a positive slope
        dx = dw = db = None
        if ctx.needs_input_grad[1]:
            want_b = ctx.bias_ref is not None and ctx.needs_input_grad[2]
            if x.shape[3] == 3:
                dw, db = ops.conv3x3_wgrad_rgb(gz, x, 0, want_bias=want_b, dw_out=grad_out(weight), db_out=grad_out(ctx.bias_ref) if want_b else None)
            else:
                dw, db = ops.conv3x3_wgrad(x, gz, ctx.stride, want_bias=want_b, ps_in=ctx.cache.ps, dw_out=grad_out(weight),
                                           db_out=grad_out(ctx.bias_ref) if want_b else None)
        if ctx.needs_input_grad[0]:
            if not ctx.cache.ps and ops.rgb_in_dgrad_eligible(x.shape[3], weight.shape[0], ctx.stride):
                dx = ops.conv3x3_rgb_in_dgrad(gz, weight.detach(), tuple(x.shape))
            else:
                dx = ops.conv3x3_dgrad(gz, ctx.cache.for_dgrad(weight, x.shape, ctx.stride), tuple(x.shape), ctx.stride, ps_in=ctx.cache.ps)
        return dx, dw, db, None, None, None


class ConvKxKFn(Function):
    """y = conv(x, w) + b for an odd kernel size other than 3 (reference model/basic.py:4-7 accepts any; its networks use 3 only):
    the generic, untuned kernels of conv_kxk.hip.  Activations are applied by the caller (ops.relu_mask)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride):
        x = _c(x)
        ctx.stride, ctx.bias_ref = stride, bias
        ctx.save_for_backward(x, weight)
        return ops.conv_kxk_fwd(x, _c(weight.detach()), None if bias is None else bias.detach(), stride)

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gy = _c(gy)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = ops.conv_kxk_dgrad(gy, _c(weight.detach()), tuple(x.shape), ctx.stride)
        if ctx.needs_input_grad[1]:
            want_b = ctx.bias_ref is not None and ctx.needs_input_grad[2]
            dw, db = ops.conv_kxk_wgrad(x, gy, weight.shape[2], ctx.stride, want_bias=want_b, dw_out=grad_out(weight),
                                        db_out=grad_out(ctx.bias_ref) if want_b else None)
        return dx, dw, db, None


def conv3x3(x, weight, bias, cache, stride=1, act=ops.ACT_NONE, relu_in=False, relu_grad_by_consumer=False):
    return Conv3x3Fn.apply(x, weight, bias, cache, stride, act, relu_in, relu_grad_by_consumer)


# ------------------------------------------------------------------------------------------------
# conv + residual add (body tail: `res += x`, reference model/pesr.py:32-33)
# ------------------------------------------------------------------------------------------------
class ConvAddFn(Function):
    """y = conv(x, w) + b + skip"""

    @staticmethod
    def forward(ctx, x, skip, weight, bias, cache):
        x, skip = _c(x), _c(skip)
        y = ops.conv3x3_fwd(x, cache.for_fwd(weight, x.shape), cache.bias(bias), weight.shape[0], 1, skip=skip)
        ctx.cache, ctx.bias_ref = cache, bias
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gy = _c(gy)
        wpd = ctx.cache.for_dgrad(weight, x.shape) if ctx.needs_input_grad[0] else None
        dw = db = None
        if ctx.needs_input_grad[2]:
            o_w, o_b = grad_out(weight), grad_out(ctx.bias_ref)
            with _OnSide(gy.device, x, gy, fast=o_w is not None and o_b is not None):
                dw, db = ops.conv3x3_wgrad(x, gy, 1, want_bias=True, dw_out=o_w, db_out=o_b)
        dx = ops.conv3x3_dgrad(gy, wpd, tuple(x.shape), 1) if ctx.needs_input_grad[0] else None
        return dx, gy, dw, db, None


# ------------------------------------------------------------------------------------------------
# ResBlock: x + res_scale * conv2(relu(conv1(x)))             reference model/basic.py:33-52
# ------------------------------------------------------------------------------------------------
class ResBlockFn(Function):
    """Four MFMA kernels forward+backward per conv pair, no standalone elementwise pass:
    forward : r = relu(conv1(x)+b1) ; y = res_scale*(conv2(r)+b2) + x        (ReLU / scale / skip in epilogues)
    backward: dr = res_scale * dgrad2(gy) masked by r>0 ; dx = dgrad1(dr) + gy  (mask / fan-in add in epilogues)
    """

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, c1: PackedConvWeights, c2: PackedConvWeights, res_scale):
        x = _c(x)
        C = w1.shape[0]
        r = ops.conv3x3_fwd(x, c1.for_fwd(w1, x.shape), b1.detach(), C, 1, act=ops.ACT_RELU)
        y = ops.conv3x3_fwd(r, c2.for_fwd(w2, x.shape), b2.detach(), C, 1, alpha=res_scale, skip=x)
        ctx.c1, ctx.c2, ctx.res_scale = c1, c2, res_scale
        ctx.b1_ref, ctx.b2_ref = b1, b2
        ctx.save_for_backward(x, r, w1, w2)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, r, w1, w2 = ctx.saved_tensors
        gy = _c(gy)
        s = ctx.res_scale
        need_w = ctx.needs_input_grad[1] or ctx.needs_input_grad[3]
        dw1 = db1 = dw2 = db2 = None
        wpd2, wpd1 = ctx.c2.for_dgrad(w2, x.shape), ctx.c1.for_dgrad(w1, x.shape)      # (packing happens on the main stream)
        if need_w:
            o_w, o_b = grad_out(w2), grad_out(ctx.b2_ref)
            with _OnSide(gy.device, r, gy, fast=o_w is not None and o_b is not None):
                dw2, db2 = ops.conv3x3_wgrad(r, gy, 1, alpha=s, dw_out=o_w, db_out=o_b)
        dr = ops.conv3x3_dgrad(gy, wpd2, tuple(r.shape), 1, alpha=s, mask=r)
        if need_w:
            o_w, o_b = grad_out(w1), grad_out(ctx.b1_ref)
            with _OnSide(gy.device, x, dr, fast=o_w is not None and o_b is not None):
                dw1, db1 = ops.conv3x3_wgrad(x, dr, 1, dw_out=o_w, db_out=o_b)
        dx = ops.conv3x3_dgrad(dr, wpd1, tuple(x.shape), 1, skip=gy) if ctx.needs_input_grad[0] else None
        return dx, dw1, db1, dw2, db2, None, None, None


# ------------------------------------------------------------------------------------------------
# MeanShift (trainable 1x1 conv 3->3)                         reference model/basic.py:9-17
# ------------------------------------------------------------------------------------------------
class MeanShiftFn(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, x_nchw, y_nchw):
        x = _c(x)
        y = ops.meanshift_fwd(x, weight.detach(), bias.detach(), x_nchw, y_nchw)
        ctx.x_nchw, ctx.y_nchw, ctx.bias_ref = x_nchw, y_nchw, bias
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        if ctx.y_nchw:  # gradient arrives in the (NCHW) layout of y; the kernel wants NHWC
            gy = gy.permute(0, 2, 3, 1)
        gy = _c(gy)
        need_dx = ctx.needs_input_grad[0]
        dx, dw, db = ops.meanshift_bwd(gy, x, weight, ctx.x_nchw, need_dx, dw_out=grad_out(weight), db_out=grad_out(ctx.bias_ref))
        if need_dx and ctx.x_nchw:
            dx = dx.permute(0, 3, 1, 2)
        return dx, dw, db, None, None


# ------------------------------------------------------------------------------------------------
# Discriminator BasicBlock: conv(no bias) -> BatchNorm2d(train) -> LeakyReLU(0.2)   reference model/basic.py:19-31
# ------------------------------------------------------------------------------------------------
class BnLink:
    """What the block that CONSUMES a BasicBlock's output needs from it (round 6): with z, the saved statistics and the affine
    parameters of the producer's BatchNorm, the consumer's input-gradient kernel writes the gradient of the producer's output already
    multiplied by lrelu'(bn(z)) and leaves the two per-channel sums of the BatchNorm backward in its epilogue (ops.conv3x3_dgrad_bn_sums):
    the producer's backward then runs no reduction pass over the tensor.  The forward fills z .. slope; the consumer's backward fills
    part / g_ptr; the producer's backward uses them once and clears them.  Only ever attached between two blocks whose connecting tensor
    nobody else sees (Discriminator.forward's chain), so the gradient that arrives IS the tensor the consumer wrote (checked by address)."""
    __slots__ = ("z", "stats", "gamma", "beta", "slope", "part", "g_ptr")

    def __init__(self):
        self.z = self.stats = self.gamma = self.beta = self.part = self.g_ptr = None
        self.slope = 0.0


class ConvBnLReluFn(Function):
    """conv (bias optional) -> BatchNorm2d -> activation given by its negative slope (0.2 LeakyReLU, 0 ReLU, 1 none).
    training = True: batch statistics (and the running-stat update) - the Discriminator's only use; training = False: the
    running statistics (nn.BatchNorm2d in .eval(): a constructor branch of reference model/basic.py:26-30 the reference's own
    scripts never take).
    Round 6: where the conv kernel can leave the BatchNorm's sums in its epilogue (ops.conv3x3_fwd_bn_stats: the direct stride-2
    layers and the F(4,3) layers without split-K) the forward runs conv -> finalize -> apply, and the backward takes the sums of
    its reduction pass from the kernel that produced its incoming gradient (BnLink)."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, running_mean, running_var, num_batches, cache, stride, eps, momentum,
                slope, y_nchw, training, prev_link=None, link=None):
        x = _c(x)
        ctx.prev_link, ctx.link = prev_link, None
        if training and bias is None and tuple(weight.shape[2:]) == (3, 3) and ops.conv_rgb_bn_eligible(x.shape[3], weight.shape[0], stride):
            # the Discriminator's features.0: the statistics come out of the conv kernel's epilogue (no pass over z for them)
            z, y, stats = ops.conv_rgb_bn_lrelu_fwd(x, weight.detach(), gamma.detach(), beta.detach(), running_mean, running_var,
                                                    num_batches, eps, momentum, slope, y_nchw)
        else:
            res = None
            if training:
                res = ops.conv3x3_fwd_bn_stats(x, cache.for_fwd(weight, x.shape, stride), None if bias is None else bias.detach(), weight.shape[0], stride)
            if res is not None:
                z, part = res
                y, stats = ops.bn_finalize_apply(z, part, gamma.detach(), beta.detach(), running_mean, running_var, num_batches, eps, momentum,
                                                 slope, y_nchw)
            else:
                z = ops.conv3x3_fwd(x, lambda: cache.for_fwd(weight, x.shape, stride), None if bias is None else bias.detach(), weight.shape[0],
                                    stride, w_oihw=weight.detach())
                if training:
                    y, stats = ops.bn_lrelu_fwd(z, gamma.detach(), beta.detach(), running_mean, running_var, num_batches, eps,
                                                momentum, slope, y_nchw)
                else:
                    stats = ops.bn_eval_stats(running_mean, running_var, eps)
                    y = ops.bn_lrelu_eval_fwd(z, gamma.detach(), beta.detach(), stats, slope, y_nchw)
        if link is not None and training and not y_nchw:
            link.z, link.stats, link.gamma, link.beta, link.slope = z, stats, gamma.detach(), beta.detach(), slope
            ctx.link = link
        ctx.cache, ctx.stride, ctx.slope, ctx.y_nchw, ctx.training, ctx.bias_ref = cache, stride, slope, y_nchw, training, bias
        ctx.save_for_backward(x, z, weight, gamma, beta, stats)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, z, weight, gamma, beta, stats = ctx.saved_tensors
        gy = _c(gy)
        need_p = ctx.needs_input_grad[3] or ctx.needs_input_grad[4]
        bwd = ops.bn_lrelu_bwd if ctx.training else ops.bn_lrelu_eval_bwd
        # the gradient that arrives was written by the consumer's input-gradient kernel, already multiplied by lrelu'(bn(z)), and that
        # kernel left the two sums of the BatchNorm backward (BnLink): no reduction pass
        link, part = ctx.link, None
        if link is not None and link.part is not None:
            if gy.data_ptr() != link.g_ptr or tuple(gy.shape) != tuple(z.shape):
                raise RuntimeError("pesr_amd: the masked gradient a fused input-gradient kernel wrote for this BatchNorm did not arrive as "
                                   "written (its block's output has another consumer?)")
            part, link.part, link.g_ptr = link.part, None, None

            def bwd(z_, gy_, gamma_, beta_, stats_, slope_, nchw_, need_, dgamma_out=None, dbeta_out=None, accumulate=False):   # noqa: F811
                return ops.bn_lrelu_bwd_fused(z_, gy_, part, gamma_, beta_, stats_, need_, dgamma_out=dgamma_out, dbeta_out=dbeta_out,
                                              accumulate=accumulate)
        # SECOND use of this block inside one backward pass (the Discriminator sees hr and sr in one graph, reference
        # train.py:205-214): the parameter gradients are ADDED to the flat-gradient slices the first use wrote and autograd gets
        # nothing to add - its fan-in add_ plus the copy of the sum into the flat buffer were two launches per parameter tensor
        # (see INPLACE_SECOND_USE for when this is allowed)
        again = INPLACE_SECOND_USE and ctx.training and need_p and ctx.needs_input_grad[1] and ctx.bias_ref is None and \
            _SIDE_OFF and all(grad_view_claimed(p) and p.grad is None for p in (gamma, beta, weight))
        if again:
            a_g, a_b, a_w = grad_out_again(gamma), grad_out_again(beta), grad_out_again(weight)
            again = a_g is not None and a_b is not None and a_w is not None
        if again:
            dz, _, _ = bwd(z, gy, gamma.detach(), beta.detach(), stats, ctx.slope, ctx.y_nchw, True, dgamma_out=a_g, dbeta_out=a_b,
                           accumulate=True)
            dgamma = dbeta = None
        else:
            dz, dgamma, dbeta = bwd(z, gy, gamma.detach(), beta.detach(), stats, ctx.slope, ctx.y_nchw, need_p,
                                    dgamma_out=grad_out(gamma) if need_p else None, dbeta_out=grad_out(beta) if need_p else None)
        dx = dw = db = None
        rgb_in_dgrad = ops.rgb_in_dgrad_eligible(x.shape[3], weight.shape[0], ctx.stride)     # Discriminator features.0
        wpd = ctx.cache.for_dgrad(weight, x.shape, ctx.stride) if ctx.needs_input_grad[0] and not rgb_in_dgrad else None
        if again:
            if x.shape[3] == 3:
                ops.conv3x3_wgrad_rgb(dz, x, 0, want_bias=False, dw_out=a_w, accumulate=True)
            else:
                ops.conv3x3_wgrad(x, dz, ctx.stride, want_bias=False, dw_out=a_w, accumulate=True)
        elif ctx.needs_input_grad[1]:
            want_b = ctx.bias_ref is not None and ctx.needs_input_grad[2]
            o_w = grad_out(weight)
            o_b = grad_out(ctx.bias_ref) if want_b else None
            with _OnSide(dz.device, x, dz, fast=o_w is not None and (o_b is not None or not want_b), where="d"):
                if x.shape[3] == 3:
                    dw, db = ops.conv3x3_wgrad_rgb(dz, x, 0, want_bias=want_b, dw_out=o_w, db_out=o_b)
                else:
                    dw, db = ops.conv3x3_wgrad(x, dz, ctx.stride, want_bias=want_b, dw_out=o_w, db_out=o_b)
        if ctx.needs_input_grad[0]:
            prev = ctx.prev_link
            res = None
            if prev is not None and prev.z is not None and not rgb_in_dgrad:
                # x is the output of a BatchNorm + LeakyReLU block: fold that block's backward reductions into this kernel's epilogue
                res = ops.conv3x3_dgrad_bn_sums(dz, wpd, tuple(x.shape), ctx.stride, prev.z, prev.stats, prev.gamma, prev.beta, prev.slope)
            if res is not None:
                dx, prev.part = res
                prev.g_ptr = dx.data_ptr()
            else:
                dx = ops.conv3x3_rgb_in_dgrad(dz, weight.detach(), tuple(x.shape)) if rgb_in_dgrad else ops.conv3x3_dgrad(dz, wpd, tuple(x.shape), ctx.stride)
        return dx, dw, db, dgamma, dbeta, None, None, None, None, None, None, None, None, None, None, None, None


# ------------------------------------------------------------------------------------------------
# spectral normalisation of a conv weight (reference model/basic.py:25; torch.nn.utils.spectral_norm semantics)
# ------------------------------------------------------------------------------------------------
_SN_STAMP = [0]


class SpectralNormFn(Function):
    """w_hat = w / sigma(w), sigma = u^T W v after (in training mode) one power iteration that rewrites the u / v buffers in
    place.  backward: dw = (g - <g, w_hat> u v^T) / sigma with that forward's u, v (constants, as in torch)."""

    @staticmethod
    def forward(ctx, weight, u, v, update, eps):
        w_hat, sigma = ops.spectral_norm_fwd(_c(weight.detach()), u, v, update, eps)
        # (clones: the next forward's power iteration rewrites the buffers - torch clones them for the same reason, so that
        # loss = D(real) - D(fake) can back-propagate through two forward passes)
        ctx.save_for_backward(w_hat, u.clone(), v.clone(), sigma)
        ctx.weight_ref = weight
        return w_hat

    @staticmethod
    def backward(ctx, g):
        w_hat, u, v, sigma = ctx.saved_tensors
        return ops.spectral_norm_bwd(_c(g), w_hat, u, v, sigma, dw_out=grad_out(ctx.weight_ref)), None, None, None, None


def spectral_normalize(weight, u, v, training, eps=1e-12):
    """The normalised weight of a spectral_norm-wrapped conv for THIS forward.  It is a fresh tensor every time; a unique
    (negative) epoch keeps the packed-weight caches from mistaking it for an earlier one whose memory the allocator re-used."""
    w_hat = SpectralNormFn.apply(weight, u, v, bool(training), eps)
    _SN_STAMP[0] += 1
    setattr(w_hat, _EPOCH_ATTR, -_SN_STAMP[0])
    return w_hat


class ScaleAddFn(Function):
    """y = alpha * a + b (`res = body(x).mul(res_scale); res += x`, reference model/basic.py:49-50) - only the un-fused
    ResBlock variants need it; the default ResBlock has it in its second conv's epilogue."""

    @staticmethod
    def forward(ctx, a, b, alpha):
        ctx.alpha = alpha
        return ops.relu_mask(_c(a), None, _c(b), alpha=alpha)

    @staticmethod
    def backward(ctx, gy):
        gy = _c(gy)
        return ops.relu_mask(gy, None, None, alpha=ctx.alpha), gy, None


# ------------------------------------------------------------------------------------------------
# Twice-differentiable building blocks of the Discriminator (gradient penalty, reference train.py:216-226:
# torch.autograd.grad(D(x_both), x_both, create_graph=True) and then .backward() through that gradient).
# Every first-order backward here is itself an autograd Function, so the graph of dD/dx exists; the second-order
# rules are: conv dgrad <-> conv forward / wgrad, Linear likewise, LeakyReLU masks are piecewise constant, and the
# training-mode BatchNorm backward has its own backward kernel (pesr_bn_bwd_bwd).  Un-fused on purpose: autograd has
# to see z between conv and BatchNorm.  Used only for the penalty's extra D forward (Discriminator.forward_second_order).
# ------------------------------------------------------------------------------------------------
class Conv2Fn(Function):
    """z = conv(x, w), no bias."""

    @staticmethod
    def forward(ctx, x, weight, cache, stride):
        x = _c(x)
        z = ops.conv3x3_fwd(x, lambda: cache.for_fwd(weight, x.shape, stride), None, weight.shape[0], stride, w_oihw=weight.detach())
        ctx.cache, ctx.stride = cache, stride
        ctx.save_for_backward(x, weight)
        return z

    @staticmethod
    def backward(ctx, gz):
        x, weight = ctx.saved_tensors
        gz = _c(gz)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = ConvDgrad2Fn.apply(gz, weight, tuple(x.shape), ctx.cache, ctx.stride)
        if ctx.needs_input_grad[1]:
            xd, gd = x.detach(), gz.detach()
            if x.shape[3] == 3:
                dw, _ = ops.conv3x3_wgrad_rgb(gd, xd, 0, want_bias=False)
            else:
                dw, _ = ops.conv3x3_wgrad(xd, gd, ctx.stride, want_bias=False)
        return dx, dw, None, None


class ConvDgrad2Fn(Function):
    """dx = conv_transpose(gz, w): the input gradient of Conv2Fn as a differentiable op.  Its backward for g = dL/d(dx):
    dL/d(gz) = conv(g, w) (the forward conv), dL/dw = wgrad(x := g, dy := gz)."""

    @staticmethod
    def forward(ctx, gz, weight, x_shape, cache, stride):
        gz = _c(gz)
        dx = ops.conv3x3_dgrad(gz, cache.for_dgrad(weight, x_shape, stride), x_shape, stride)
        ctx.cache, ctx.stride = cache, stride
        ctx.save_for_backward(gz, weight)
        return dx

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        gz, weight = ctx.saved_tensors
        g = _c(g)
        l_gz = l_w = None
        if ctx.needs_input_grad[0]:
            l_gz = ops.conv3x3_fwd(g, lambda: ctx.cache.for_fwd(weight, g.shape, ctx.stride), None, weight.shape[0], ctx.stride,
                                   w_oihw=weight.detach())
        if ctx.needs_input_grad[1]:
            if g.shape[3] == 3:
                l_w, _ = ops.conv3x3_wgrad_rgb(gz, g, 0, want_bias=False)
            else:
                l_w, _ = ops.conv3x3_wgrad(g, gz, ctx.stride, want_bias=False)
        return l_gz, l_w, None, None, None


class Bn2Fn(Function):
    """u = gamma * xhat(z) + beta with batch statistics (nn.BatchNorm2d in training mode, running stats updated)."""

    @staticmethod
    def forward(ctx, z, gamma, beta, running_mean, running_var, num_batches, eps, momentum):
        z = _c(z)
        u, stats = ops.bn_lrelu_fwd(z, gamma.detach(), beta.detach(), running_mean, running_var, num_batches, eps, momentum, 1.0, False)
        ctx.save_for_backward(z, gamma, stats)
        return u

    @staticmethod
    def backward(ctx, gu):
        z, gamma, stats = ctx.saved_tensors
        gu = _c(gu)
        dz = dgamma = dbeta = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            dz_plain, dgamma, dbeta = ops.bn_lrelu_bwd(z.detach(), gu.detach(), gamma.detach(), gamma.detach(), stats, 1.0, False, True)
        if ctx.needs_input_grad[0]:
            dz = BnBwd2Fn.apply(gu, z, gamma, stats)
        return dz, dgamma, dbeta, None, None, None, None, None


class BnBwd2Fn(Function):
    """dz = gamma * invstd * (du - mean(du) - xhat * mean(du * xhat)): the training-mode BatchNorm backward as a differentiable
    op (in du, z AND gamma: xhat and invstd depend on z)."""

    @staticmethod
    def forward(ctx, du, z, gamma, stats):
        du = _c(du)
        dz, _, _ = ops.bn_lrelu_bwd(z.detach(), du, gamma.detach(), gamma.detach(), stats, 1.0, False, False)
        ctx.save_for_backward(du, z, gamma, stats)
        return dz

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        du, z, gamma, stats = ctx.saved_tensors
        l_du, l_z, l_ga = ops.bn_bwd_bwd(z, du, _c(g), gamma, stats, ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2])
        return l_du, l_z, l_ga, None


class LRelu2Fn(Function):
    """y = leaky_relu(u); its backward is the (differentiable) mask multiply."""

    @staticmethod
    def forward(ctx, u, slope):
        u = _c(u)
        y = ops.relu_mask(u, u, slope=slope)
        ctx.slope = slope
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        return MaskMul2Fn.apply(gy, y, ctx.slope), None


class MaskMul2Fn(Function):
    """out = gy where ref > 0, slope * gy elsewhere: linear in gy, piecewise constant in ref."""

    @staticmethod
    def forward(ctx, gy, ref, slope):
        ctx.slope = slope
        ctx.save_for_backward(ref)
        return ops.relu_mask(_c(gy), ref, slope=slope)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        (ref,) = ctx.saved_tensors
        return ops.relu_mask(_c(g), ref, slope=ctx.slope), None, None


class Linear2Fn(Function):
    """y = x W^T + b."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x = _c(x)
        ctx.save_for_backward(x, weight)
        return ops.linear_fwd(x, weight.detach(), bias.detach(), ops.ACT_NONE, 0.0)

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gy = _c(gy)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = LinDgrad2Fn.apply(gy, weight)
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            dw, db = ops.linear_wgrad(gy.detach(), x.detach(), want_bias=True)
        return dx, dw, db


class LinDgrad2Fn(Function):
    """dx = gy W; backward for g = dL/d(dx): dL/d(gy) = g W^T, dL/dW = gy^T g."""

    @staticmethod
    def forward(ctx, gy, weight):
        gy = _c(gy)
        ctx.save_for_backward(gy, weight)
        return ops.linear_dgrad(gy, weight.detach())

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        gy, weight = ctx.saved_tensors
        g = _c(g)
        l_gy = ops.linear_fwd(g, weight.detach(), None, ops.ACT_NONE, 0.0) if ctx.needs_input_grad[0] else None
        l_w = ops.linear_wgrad(gy, g, want_bias=False)[0] if ctx.needs_input_grad[1] else None
        return l_gy, l_w


# ------------------------------------------------------------------------------------------------
# Linear (+LeakyReLU)                                          reference model/pesr.py:69-74
# ------------------------------------------------------------------------------------------------
# LinearFn's second use inside one backward pass may accumulate into the flat-gradient slice in place (below).  That is only
# correct while NOTHING ELSE contributes to the same weight in that pass: a third contribution arriving as an ordinary tensor
# (the gradient penalty's LinDgrad2Fn, reference train.py:216-226) makes autograd's input buffer a fresh temporary, and an
# in-place add into the slice would then miss it - whether it does depends on the engine's node order.  Trainer switches the
# fast path off around every backward pass that runs with the penalty.
INPLACE_SECOND_USE = True


class LinearFn(Function):
    """grad_rows (optional): only the first grad_rows rows of x need an input gradient - the Discriminator's classifier on [sr; hr] in the
    generator phase, whose hr half comes from a no_grad pass (Discriminator.classify); the other rows of the returned gradient are
    never read (torch.cat's backward hands them to an input that requires none) and stay uninitialised."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, slope, grad_rows=None):
        x = _c(x)
        y = ops.linear_fwd(x, weight.detach(), bias.detach(), act, slope)
        ctx.act, ctx.slope, ctx.bias_ref, ctx.grad_rows = act, slope, bias, grad_rows
        ctx.save_for_backward(x, weight, y if act != ops.ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, y = ctx.saved_tensors
        gy = _c(gy)
        if ctx.act != ops.ACT_NONE:
            if gy.numel() % 4 == 0:
                gy = ops.relu_mask(gy, y, slope=ctx.slope if ctx.act == ops.ACT_LRELU else 0.0)
            else:  # scalar-sized tail case
                gy = torch.where(y > 0, gy, gy * (ctx.slope if ctx.act == ops.ACT_LRELU else 0.0))
        dx = None
        if ctx.needs_input_grad[0]:
            r = ctx.grad_rows
            if r is not None and 0 < r < gy.shape[0]:
                dx = torch.empty((gy.shape[0], weight.shape[1]), dtype=torch.float32, device=gy.device)
                ops.linear_dgrad(gy[:r], weight.detach(), out=dx[:r])
            else:
                dx = ops.linear_dgrad(gy, weight.detach())
        dw = db = None
        if ctx.needs_input_grad[1]:
            o_w, o_b = grad_out(weight), grad_out(ctx.bias_ref)
            if INPLACE_SECOND_USE and o_w is None and o_b is None and weight.grad is None:
                # second use of this layer in the same backward pass (the Discriminator sees hr and sr in one graph,
                # reference train.py:205-214): accumulate straight into the flat-gradient slices the first use wrote,
                # and hand autograd nothing to add - for classifier.0 that add was a 302 MB elementwise kernel
                a_w, a_b = grad_out_again(weight), grad_out_again(ctx.bias_ref)
                if a_w is not None and a_b is not None:
                    ops.linear_wgrad(gy, x, want_bias=True, dw_out=a_w, db_out=a_b, accumulate=True)
                    return dx, None, None, None, None, None
            dw, db = ops.linear_wgrad(gy, x, want_bias=True, dw_out=o_w, db_out=o_b)
        return dx, dw, db, None, None, None


class SplitRowsFn(Function):
    """t[:n], t[off:off + n] whose backward is ONE buffer (two slicing nodes cost two zero fills, two copies and an add - five launches
    for the [2B, 1] logits of the Discriminator's paired classifier).  Rows outside the two pieces get zero gradients."""

    @staticmethod
    def forward(ctx, t, n, off):
        ctx.n, ctx.off, ctx.rows = n, off, t.shape[0]
        return t[:n].clone(), t[off:off + n].clone()

    @staticmethod
    def backward(ctx, ga, gb):
        if ga is None and gb is None:
            return None, None, None
        ref = ga if ga is not None else gb
        if ctx.off == ctx.n and ctx.rows == 2 * ctx.n and ga is not None and gb is not None:
            return torch.cat([ga, gb]), None, None
        g = ref.new_zeros((ctx.rows,) + tuple(ref.shape[1:]))
        if ga is not None:
            g[:ctx.n] = ga
        if gb is not None:
            g[ctx.off:ctx.off + ctx.n] = gb
        return g, None, None


# ------------------------------------------------------------------------------------------------
# 2x2 max-pool behind a ReLU                                   torchvision vgg19 features (reference model/vgg.py:8-10)
# ------------------------------------------------------------------------------------------------
class MaxPoolFn(Function):
    @staticmethod
    def forward(ctx, x, relu_in):
        x = _c(x)
        ctx.relu_in = relu_in
        ctx.save_for_backward(x)
        return ops.maxpool2x2_fwd(x)

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        return ops.maxpool2x2_bwd(x, _c(gy), ctx.relu_in), None


class VggTailFn(Function):
    """conv4_1 .. conv5_4 (+ pool4) of vgg19 features[:35] (reference model/vgg.py:8-10,19-26) for the sr AND the hr branch in one
    batch: forward on cat(a, b) - one launch per layer over 2B images instead of two over B -, backward on the sr half only (the
    hr branch is the reference's no_grad pass; the weights are frozen, so the backward is a chain of input gradients).  steps:
    [(kind, module, has_relu)] from VGG._steps.  Per layer exactly what Conv3x3Fn / MaxPoolFn do: a conv's ReLU is fused into its
    epilogue and its mask is applied by the consumer's backward (the next conv's input gradient or the pool's)."""

    @staticmethod
    def forward(ctx, a, b, steps):
        B = a.shape[0]
        x = torch.cat([_c(a), _c(b)])
        plan, saved, prev_relu = [], [], False
        for kind, m, has_relu in steps:
            saved.append(x[:B])                     # (the sr half: a contiguous view)
            if kind == "conv":
                plan.append((kind, m, prev_relu, tuple(x[:B].shape)))
                x = ops.conv3x3_fwd(x, m.packed.for_fwd(m.weight, x.shape, 1), m.packed.bias(m.bias), m.out_channels, 1,
                                    act=ops.ACT_RELU if has_relu else ops.ACT_NONE, w_oihw=m.weight.detach())
                prev_relu = has_relu
            else:
                plan.append((kind, m, prev_relu, None))
                x = ops.maxpool2x2_fwd(x)
                prev_relu = False
        ctx.plan = plan
        ctx.save_for_backward(*saved)
        fa, fb = x[:B], x[B:]
        ctx.mark_non_differentiable(fb)
        return fa, fb

    @staticmethod
    def backward(ctx, ga, _gb):
        g = _c(ga)
        for (kind, m, relu_in, shape), xin in zip(reversed(ctx.plan), reversed(ctx.saved_tensors)):
            if kind == "conv":
                g = ops.conv3x3_dgrad(g, m.packed.for_dgrad(m.weight, shape, 1), shape, 1, mask=xin if relu_in else None)
            else:
                g = ops.maxpool2x2_bwd(xin, g, relu_in)
        return g, None, None


# ------------------------------------------------------------------------------------------------
# losses                                                       reference train.py:131-140
# ------------------------------------------------------------------------------------------------
class L1LossFn(Function):
    """nn.L1Loss() (mean) on NHWC 3-channel tensors; the gradient is produced in the same pass."""

    @staticmethod
    def forward(ctx, sr, hr):
        sr, hr = _c(sr), _c(hr)
        out, grad = ops.loss_l1_tv(sr, hr, 1.0 / sr.numel(), 0.0, need_grad=sr.requires_grad)
        ctx.save_for_backward(grad)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None


class TVLossFn(Function):
    """Sum of absolute horizontal + vertical differences (reference train.py:137-140)."""

    @staticmethod
    def forward(ctx, sr):
        sr = _c(sr)
        out, grad = ops.loss_l1_tv(sr, sr, 0.0, 1.0, need_grad=sr.requires_grad)
        ctx.save_for_backward(grad)
        return out[1]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g


class MSELossFn(Function):
    """F.mse_loss(a, b) (mean); only `a` gets a gradient (b is the no_grad branch, reference model/vgg.py:24-26)."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = _c(a), _c(b)
        out, grad = ops.loss_mse(a, b, 2.0 / a.numel(), need_grad=a.requires_grad)
        ctx.save_for_backward(grad)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None


class GanLossFn(Function):
    """The discriminator (side 0) or generator (side 1) loss of reference train.py:210-213 / :244-253 on the [B, 1] logits, scaled
    (alpha_gan): one kernel computes the value and both gradients (ops.gan_loss); backward only scales them."""

    @staticmethod
    def forward(ctx, pred_real, pred_fake, gan_type, side, focal, gamma, scale):
        pr, pf = _c(pred_real), _c(pred_fake)
        out, d_r, d_f = ops.gan_loss(pr, pf, gan_type, side, focal, gamma, scale, need_real=pred_real.requires_grad,
                                     need_fake=pred_fake.requires_grad)
        ctx.save_for_backward(d_r, d_f)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        d_r, d_f = ctx.saved_tensors
        return (None if d_r is None else d_r * g), (None if d_f is None else d_f * g), None, None, None, None, None


def gan_loss(pred_real, pred_fake, gan_type, side, focal=False, gamma=1.0, scale=1.0):
    return GanLossFn.apply(pred_real, pred_fake, gan_type, side, bool(focal), float(gamma), float(scale))


def l1_loss(sr, hr):
    return L1LossFn.apply(sr, hr)


def tv_loss(sr):
    return TVLossFn.apply(sr)


def mse_loss(a, b):
    return MSELossFn.apply(a, b)
