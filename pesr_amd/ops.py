"""Tensor-level wrappers over the C ABI (no autograd here; see functional.py).

All activation tensors handled here are fp32, on a GPU, NHWC-contiguous: shape [N, H, W, C].
(The nn.Module boundary works with logical NCHW tensors in torch.channels_last memory format; a
`.permute(0, 2, 3, 1)` of those is exactly this view - no copies.)
There is no CPU path: a CPU tensor raises.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib

ACT_NONE, ACT_RELU, ACT_LRELU = 0, 1, 2


def _chk(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise _lib.PesrHipError(f"{name}: pesr_amd ops run only on a GPU (HIP) tensor; there is no CPU fallback")
    if t.dtype != torch.float32:
        raise _lib.PesrHipError(f"{name}: expected float32, got {t.dtype}")
    if not t.is_contiguous():
        raise _lib.PesrHipError(f"{name}: expected a contiguous tensor (NHWC physical layout)")


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class _KernelEvents:
    """Optional HIP-event bracket around every launch of ONE conv shape (bench.py's live roofline figures), kept per kind
    ("fwd", "dgrad", "wgrad").  Events are recorded on the stream the kernel is launched on (torch's current stream).
    Not inside a captured step: an event recorded under capture is only a dependency and cannot be timed, and recording it as an
    external event node (hipEventRecordWithFlags(hipEventRecordExternal)) is refused under capture by this ROCm runtime
    (hipErrorInvalidValue; tried in round 3) - bench.py --hip-graph takes its kernel times from eager steps."""

    def __init__(self):
        self.shape, self.pairs, self.every, self._n = None, {}, 1, {}

    def enable(self, shape, every=1):
        """Watch launches of `shape`; bracket every `every`-th launch of each kind (two event records per bracket are not free:
        bracketing all ~200 watched launches of a GAN step cost 2 % of the step)."""
        self.shape, self.pairs, self.every, self._n = tuple(shape), {}, max(1, int(every)), {}

    def begin(self, kind, N, H, W, Cin, Cout, stride):
        """-> an open bracket (or None when this launch is not the watched shape / not sampled); close it with end()."""
        if self.shape is None or self.shape != (N, H, W, Cin, Cout, stride):
            return None
        n = self._n.get(kind, 0)
        self._n[kind] = n + 1
        if n % self.every:
            return None
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
        return (kind, e0)

    def end(self, br):
        if br is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self.pairs.setdefault(br[0], []).append((br[1], e1))

    def drain(self):
        """-> {kind: (average milliseconds per launch, launches)}; disables recording."""
        pairs, self.pairs, self.shape = self.pairs, {}, None
        if not pairs:
            return {}
        torch.cuda.synchronize()
        return {k: (sum(a.elapsed_time(b) for a, b in v) / len(v), len(v)) for k, v in pairs.items()}


KERNEL_EVENTS = _KernelEvents()


class _OpEvents:
    """Optional HIP-event brackets around the HBM-bound ops that have no conv shape key (bench.py's config-5 side measurement:
    the C -> 3 and 3 -> C convs, MeanShift), recorded on the launch stream; off unless enabled."""

    def __init__(self):
        self.on, self.pairs = False, {}

    def enable(self):
        self.on, self.pairs = True, {}

    def begin(self, name):
        if not self.on:
            return None
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
        return (name, e0)

    def end(self, br):
        if br is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self.pairs.setdefault(br[0], []).append((br[1], e1))

    def drain(self):
        """-> {name: (mean milliseconds per launch, launches)}; disables recording."""
        pairs, self.pairs, self.on = self.pairs, {}, False
        if not pairs:
            return {}
        torch.cuda.synchronize()
        return {k: (sum(a.elapsed_time(b) for a, b in v) / len(v), len(v)) for k, v in pairs.items()}


OP_EVENTS = _OpEvents()


class _FlopCount:
    """Optional tally of the matrix-pipe work of a step (bench.py's step_issued_frac): per conv / Linear launch the
    ALGORITHMIC flops of the op and the flops the dispatched kernel ISSUES for them (x 1/2 on the F(4,3) kernels, x 2/3 on
    F(2,3), x 1 on the direct ones; zero padding of the direct kernels' tiles is not counted as issued work)."""

    def __init__(self):
        self.on, self.alg, self.issued, self.by = False, 0.0, 0.0, {}

    def start(self):
        self.on, self.alg, self.issued, self.by = True, 0.0, 0.0, {}

    def stop(self):
        self.on = False
        return {"algorithmic": self.alg, "issued": self.issued, "by_kernel_family": dict(self.by)}

    def add(self, flops, frac, family):
        if self.on:
            self.alg += flops
            self.issued += flops * frac
            a = self.by.setdefault(family, [0.0, 0.0])
            a[0] += flops
            a[1] += flops * frac


FLOPS = _FlopCount()


def _conv_family(wp):
    if isinstance(wp, Bf16x3Packed):
        return 3.0, "split-bf16 (3 bf16 products per multiply)"
    if isinstance(wp, Bf16Packed):
        return 1.0, "bf16"
    if isinstance(wp, Wino4Packed):
        return 0.5, "F(4,3)"
    if isinstance(wp, WinoPacked):
        return 2.0 / 3.0, "F(2,3)"
    return 1.0, "direct"

_workspaces = {}


def workspace(nbytes: int, device: torch.device) -> torch.Tensor:
    """Grow-only per-device scratch buffer (owned by torch's caching allocator)."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


# ------------------------------------------------------------------------------------------------
# weight packing
# ------------------------------------------------------------------------------------------------
def pack_conv3x3(w: torch.Tensor, mode: int, ps: bool = False) -> torch.Tensor:
    """OIHW [O, I, 3, 3] -> packed [9, R/16, Nn, 16] (mode 0: forward, mode 1: dgrad)."""
    _chk(w, "pack_conv3x3.w")
    O, I = w.shape[0], w.shape[1]
    assert tuple(w.shape[2:]) == (3, 3), f"pack_conv3x3: a 3x3 kernel is required, got {tuple(w.shape)}"
    R, Nn = (I, O) if mode == 0 else (O, I)
    nn_pad = 16 if Nn <= 16 else (Nn + 63) // 64 * 64
    out = torch.empty(9 * ((R + 15) // 16 * 16) * nn_pad, dtype=torch.float32, device=w.device)
    _lib.check(_lib.lib().pesr_pack_conv3x3(_p(w), _p(out), O, I, mode, int(ps), _stream()), "pesr_pack_conv3x3")
    return out


def pack_bias_ps(b: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _chk(b, "pack_bias_ps.b")
    if out is None:
        out = torch.empty_like(b)
    assert out.shape == b.shape and out.is_contiguous()
    _lib.check(_lib.lib().pesr_pack_bias_ps(_p(b), _p(out), b.numel(), _stream()), "pesr_pack_bias_ps")
    return out


# ------------------------------------------------------------------------------------------------
# 3x3 conv family
# ------------------------------------------------------------------------------------------------
class WinoPacked:
    """Weights packed for the Winograd F(2,3)-along-x kernel (pesr_pack_conv3x3_wino); conv3x3_fwd / conv3x3_dgrad dispatch on it."""
    __slots__ = ("t",)

    def __init__(self, t: torch.Tensor):
        self.t = t


USE_WINO = __import__("os").environ.get("PESR_WINO", "1") != "0"     # PESR_WINO=0: direct kernel everywhere
# PESR_WGRAD_WINO=0: direct weight-gradient kernel everywhere (passed to the library as the explicit `algo` argument)
USE_WGRAD_WINO = __import__("os").environ.get("PESR_WGRAD_WINO", "1") != "0"
USE_WGRAD_WINO4 = __import__("os").environ.get("PESR_WGRAD_WINO4", "1") != "0"   # =0: F(2,3) weight gradient instead of F(4,3)


def _wg4_plan_ok(N, H, W, Cin, Cout):
    """wg4_plan of csrc/conv3x3_wgrad_wino4.hip: (covered, images per strip)."""
    if W % 4 or Cin % 64 or Cout % 64 or N < 1 or H < 1:
        return False, 1
    xtw, segs_y, side = W // 4, (H + 1) // 2, 1
    if W < 48:
        if xtw < 2 or 12 % xtw:
            return False, 1
        side = 12 // xtw
        groups = (N + side - 1) // side
        if groups * 12 * 8 > N * xtw * 9 or side * H * W * max(Cin, Cout) * 4 >= 1 << 31:
            return False, side
        total = groups * segs_y
    else:
        segs_x = (xtw + 11) // 12
        if segs_x * 12 * 8 > xtw * 9:
            return False, 1
        total = N * segs_x * segs_y
    tiles = (Cout // 64) * (Cin // 32)
    split = max(1, min((256 + tiles - 1) // tiles, total))
    sps = (total + split - 1) // split
    if sps > segs_y:
        sps = (sps + segs_y - 1) // segs_y * segs_y
    split = (total + sps - 1) // sps
    return tiles * split >= 8, side


def wgrad_kernel_for(N, H, W, Cin, Cout):
    """(kernel name, fraction of the algorithmic flops it issues on the matrix pipe) of the stride-1 weight gradient."""
    if USE_WGRAD_WINO and Cin % 64 == 0 and Cout % 64 == 0:
        if USE_WINO4 and USE_WGRAD_WINO4:
            ok, side = _wg4_plan_ok(N, H, W, Cin, Cout)
            if ok and side == 1 and USE_WGRAD_WINO4_16X16:
                return "conv3x3_wgrad_wino4_kernel", 0.5
            if ok and not USE_WGRAD_WINO4_16X16:     # (rows shorter than a strip - images side by side - exist on the 32x32x2 kernel only)
                return ("conv3x3_wgrad_wino4x_kernel", 0.5) if USE_WGRAD_WINO4_1D else ("conv3x3_wgrad_wino4p_kernel", 1.0 / 3.0)
        if W >= 48 and W % 2 == 0 and ((W // 2 + 23) // 24) * 24 * 8 <= (W // 2) * 9:
            return "conv3x3_wgrad_wino_kernel", 2.0 / 3.0
    return "conv3x3_wgrad_kernel", 1.0


def wino_eligible(N: int, H: int, W: int, Cin: int, Cout: int, stride: int = 1) -> bool:
    """The Winograd kernel applies (stride 1, even width, Cout % 128 == 0) AND its 288-pixel x 128-channel tiles fill the
    chip; small layers stay on the direct kernel (smaller tiles, split-K).  Cin / Cout are those of the problem the kernel
    runs (for an input gradient: Cin = the forward conv's Cout and vice versa)."""
    if not USE_WINO or stride != 1 or W % 2 or Cin % 64 or Cout % 128:
        return False
    per_img = (H * (W // 2) + 143) // 144
    wgs = N * per_img * (Cout // 128)
    if wgs >= 192:
        return True
    # fewer tiles: the kernel splits the Cin chunks over up to Cin/64 workgroups per tile - worth it while the 144-x-tile
    # tiles are (nearly) full, i.e. from 24x24 images up
    return per_img * 144 <= 1.15 * H * (W // 2) and wgs * min(8, Cin // 64) >= 192


class Wino4Packed:
    """Weights packed for the Winograd F(4,3)-along-x kernel (pesr_pack_conv3x3_wino4)."""
    __slots__ = ("t",)

    def __init__(self, t: torch.Tensor):
        self.t = t


USE_WINO4 = __import__("os").environ.get("PESR_WINO4", "1") != "0"   # PESR_WINO4=0: no F(4,3) kernel (F(2,3) / direct instead)

_W4_SCORE = {}


def wino4_eligible(N: int, H: int, W: int, Cin: int, Cout: int, stride: int = 1, ps_out: bool = False) -> bool:
    """The F(4,3) kernel applies (stride 1, W % 4 == 0, Cin % 16 == 0, Cout % 64 == 0; Cout % 256 == 0 with a fused
    PixelShuffle store), its 576-pixel x 64-channel tiles (split-K included) give at least 192 workgroups, AND at least 78 %
    of the tile area lies inside the image: it issues 1/2 of the direct conv's MFMAs where F(2,3) issues 2/3, so it wins
    from ~0.75 of F(2,3)'s tile efficiency up.  Cin / Cout are those of the problem the kernel runs."""
    if not (USE_WINO and USE_WINO4) or stride != 1 or W % 4 or Cin % 16 or Cout % 64 or (ps_out and Cout % 256):
        return False
    key = (N, H, W, Cin, Cout, ps_out)
    sc = _W4_SCORE.get(key)
    if sc is None:
        sc = _W4_SCORE[key] = _lib.lib().pesr_conv3x3_wino4_score(N, H, W, Cin, Cout, 0 if ps_out else 1)
    return sc >= 780


def pack_conv3x3_wino4(w: torch.Tensor, mode: int, ps: bool = False) -> Wino4Packed:
    """OIHW [O, I, 3, 3] -> transformed [18, R/16, Nn, 16] (mode 0: forward, mode 1: dgrad; ps: sub-pixel-major O order)."""
    _chk(w, "pack_conv3x3_wino4.w")
    O, I = w.shape[0], w.shape[1]
    assert tuple(w.shape[2:]) == (3, 3), f"pack_conv3x3_wino4: a 3x3 kernel is required, got {tuple(w.shape)}"
    out = torch.empty(18 * O * I, dtype=torch.float32, device=w.device)
    rc = _lib.lib().pesr_pack_conv3x3_wino4(_p(w), _p(out), O, I, mode, int(ps), _stream())
    _lib.check(rc, f"pesr_pack_conv3x3_wino4[{O}x{I},mode{mode}]")
    return Wino4Packed(out)


def pack_conv3x3_wino(w: torch.Tensor, mode: int, ps: bool = False) -> WinoPacked:
    """OIHW [O, I, 3, 3] -> transformed [12, R/16, Nn, 16] (mode 0: forward, mode 1: dgrad; ps: sub-pixel-major O order)."""
    _chk(w, "pack_conv3x3_wino.w")
    O, I = w.shape[0], w.shape[1]
    assert tuple(w.shape[2:]) == (3, 3), f"pack_conv3x3_wino: a 3x3 kernel is required, got {tuple(w.shape)}"
    out = torch.empty(12 * O * I, dtype=torch.float32, device=w.device)
    rc = _lib.lib().pesr_pack_conv3x3_wino(_p(w), _p(out), O, I, mode, int(ps), _stream())
    _lib.check(rc, f"pesr_pack_conv3x3_wino[{O}x{I},mode{mode}]")
    return WinoPacked(out)


# ---- the OPTIONAL bf16-operand mode --------------------------------------------------------------------------------
# PRECISION = "bf16" (set_precision / PESR_PRECISION / train.py --precision): the stride-1 convs with 32-multiple input and
# 64-multiple output channels run on v_mfma_f32_16x16x32_bf16 - operands rounded to bf16, fp32 accumulation, fp32 tensors in HBM.
# Everything else (and everything by default) stays on the fp32 kernels.
PRECISION = __import__("os").environ.get("PESR_PRECISION", "fp32")
_B16_SCORE = {}
_BF16_NO_S2 = __import__("os").environ.get("PESR_BF16_NO_S2", "0") == "1"      # A/B switch of scripts/gpu_call51.sh: stride-2 forwards stay fp32
# layers whose bf16 launch would have fewer workgroups stay on the fp32 kernels (tests lower it; at 64 workgroups the 16x12x12x512
# layers take 47 us against 89 on the F(4,3) kernel, scripts/bf16_small_time.py).  The env form is for A/B runs only.
BF16_MIN_WGS = int(__import__("os").environ.get("PESR_BF16_MIN_WGS", "64"))


PRECISIONS = ("fp32", "bf16", "split-bf16")


def set_precision(p: str) -> None:
    global PRECISION
    if p not in PRECISIONS:
        raise ValueError(f"precision must be one of {PRECISIONS}, got {p!r}")
    PRECISION = p


class Bf16Packed:
    """Weights rounded to bf16 and packed for conv3x3_bf16_kernel (pesr_pack_conv3x3_bf16)."""
    __slots__ = ("t",)

    def __init__(self, t: torch.Tensor):
        self.t = t


# ---- the OPTIONAL split-bf16 mode (round 4) ---------------------------------------------------------------------------
# PRECISION = "split-bf16": forward and input gradient of the stride-1 convs with Cin % 32 == 0 and Cout % 128 == 0 run on the bf16
# MFMA with every operand split into hi + lo bf16 terms and three products per multiply (conv3x3_bf16x3.hip): 3.6 .. 4.7e-6 of the
# output maximum against fp64 - inside the fp32 kernels' own tolerances, so the mode is tested against the fp32 oracle.  Weight
# gradients, stride-2 convs, 64-channel layers and everything else stay on the fp32 kernels.
class Bf16x3Packed:
    """Weights split into hi + lo bf16 planes and packed for conv3x3_bf16x3_kernel (pesr_pack_conv3x3_bf16x3)."""
    __slots__ = ("t",)

    def __init__(self, t: torch.Tensor):
        self.t = t


_B3_SCORE = {}
BF16X3_MIN_WGS = 128


def bf16x3_eligible(N: int, H: int, W: int, Cin: int, Cout: int, stride: int = 1, ps_out: bool = False, ps_in: bool = False) -> bool:
    """PRECISION is "split-bf16" and the split kernel covers the problem it would run (an input gradient: Cin / Cout swapped)."""
    # (ps_out: an n-tile - 256 or 128 channels, the planner's choice - must stay inside one of the four sub-pixel planes)
    if PRECISION != "split-bf16" or stride != 1 or Cin % 32 or Cout % 128 or (ps_out and Cout % 1024) or (ps_in and Cin % 128):
        return False
    key = (N, H, W, Cin, Cout)
    sc = _B3_SCORE.get(key)
    if sc is None:
        sc = _B3_SCORE[key] = _lib.lib().pesr_conv3x3_bf16x3_score(N, H, W, Cin, Cout, BF16X3_MIN_WGS)
    return sc >= 780


def pack_conv3x3_bf16x3(w: torch.Tensor, mode: int, ps: bool = False) -> Bf16x3Packed:
    """OIHW [O, I, 3, 3] fp32 -> [2 (hi, lo)][9, R/32, Nn, 32] bf16 (mode 0: forward, mode 1: dgrad with flipped taps; ps: sub-pixel-major O)."""
    _chk(w, "pack_conv3x3_bf16x3.w")
    O, I = w.shape[0], w.shape[1]
    assert tuple(w.shape[2:]) == (3, 3), f"pack_conv3x3_bf16x3: a 3x3 kernel is required, got {tuple(w.shape)}"
    out = torch.empty(2 * 9 * O * I, dtype=torch.bfloat16, device=w.device)
    rc = _lib.lib().pesr_pack_conv3x3_bf16x3(_p(w), _p(out), O, I, mode, int(ps), _stream())
    _lib.check(rc, f"pesr_pack_conv3x3_bf16x3[{O}x{I},mode{mode}]")
    return Bf16x3Packed(out)


def bf16_eligible(N: int, H: int, W: int, Cin: int, Cout: int, stride: int = 1, ps_out: bool = False, ps_in: bool = False) -> bool:
    """PRECISION is "bf16" and the bf16 kernel covers the shape with at least 78 % of its tile area inside the image.
    Cin / Cout are those of the problem the kernel runs."""
    bn = 256 if Cout % 256 == 0 else (128 if Cout % 128 == 0 else 64)        # output channels per workgroup (conv3x3_bf16.hip)
    if PRECISION != "bf16" or stride not in (1, 2) or Cin % 32 or Cout % 64 or (ps_out and Cout % (4 * bn)) or (ps_in and Cin % 128):
        return False
    if stride == 2 and (ps_out or ps_in or _BF16_NO_S2):
        return False
    key = (N, H, W, Cin, Cout, BF16_MIN_WGS, stride)
    sc = _B16_SCORE.get(key)
    if sc is None:       # stride 2: the FORWARD kernel only (H, W: the input's size); its gradients stay on the fp32 kernels
        score = _lib.lib().pesr_conv3x3_bf16_score if stride == 1 else _lib.lib().pesr_conv3x3_bf16_s2_score
        sc = _B16_SCORE[key] = score(N, H, W, Cin, Cout, BF16_MIN_WGS)
    return sc >= 780


def bf16_s2_dgrad_eligible(N: int, H: int, W: int, Cout_fwd: int, Cin_fwd: int) -> bool:
    """PRECISION is "bf16" and the stride-2 input-gradient form of the bf16 kernel covers dx [N, H, W, Cin_fwd]."""
    if PRECISION != "bf16" or _BF16_NO_S2 or Cout_fwd % 32 or Cin_fwd % 64:
        return False
    key = (N, H, W, Cout_fwd, Cin_fwd, BF16_MIN_WGS, "s2d")
    sc = _B16_SCORE.get(key)
    if sc is None:
        sc = _B16_SCORE[key] = _lib.lib().pesr_conv3x3_bf16_s2_dgrad_score(N, H, W, Cout_fwd, Cin_fwd, BF16_MIN_WGS)
    return sc >= 780


def pack_conv3x3_bf16(w: torch.Tensor, mode: int, ps: bool = False) -> Bf16Packed:
    """OIHW [O, I, 3, 3] fp32 -> [9, R/32, Nn, 32] bf16 (mode 0: forward, mode 1: dgrad with flipped taps; ps: sub-pixel-major O)."""
    _chk(w, "pack_conv3x3_bf16.w")
    O, I = w.shape[0], w.shape[1]
    assert tuple(w.shape[2:]) == (3, 3), f"pack_conv3x3_bf16: a 3x3 kernel is required, got {tuple(w.shape)}"
    out = torch.empty(9 * O * I, dtype=torch.bfloat16, device=w.device)
    rc = _lib.lib().pesr_pack_conv3x3_bf16(_p(w), _p(out), O, I, mode, int(ps), _stream())
    _lib.check(rc, f"pesr_pack_conv3x3_bf16[{O}x{I},mode{mode}]")
    return Bf16Packed(out)


def _conv3x3_wino(x, wp, bias, skip, mask, y, N, H, W, Cin, cout, alpha, act, slope, what, ps_out=False, ps_in=False):
    """Both Winograd kernels (WinoPacked -> F(2,3), Wino4Packed -> F(4,3)) and the bf16 kernel (Bf16Packed): same arguments,
    same fused epilogue."""
    L = _lib.lib()
    if isinstance(wp, Bf16x3Packed):
        rc = L.pesr_conv3x3_bf16x3(_p(x), _p(wp.t), _p(bias), _p(skip), _p(mask), _p(y), N, H, W, Cin, cout, alpha, act, slope,
                                   int(ps_out), int(ps_in), _stream())
        _lib.check(rc, f"pesr_conv3x3_bf16x3[{what} {N}x{H}x{W}x{Cin}->{cout}]")
        return
    if isinstance(wp, Bf16Packed):
        rc = L.pesr_conv3x3_bf16(_p(x), _p(wp.t), _p(bias), _p(skip), _p(mask), _p(y), N, H, W, Cin, cout, alpha, act, slope,
                                 int(ps_out), int(ps_in), _stream())
        _lib.check(rc, f"pesr_conv3x3_bf16[{what} {N}x{H}x{W}x{Cin}->{cout}]")
        return
    nws = L.pesr_conv3x3_workspace_bytes(N, H, W, cout) if not ps_out else 0     # split-K scratch for layers with few tiles
    ws = workspace(nws, x.device) if nws else None
    four = isinstance(wp, Wino4Packed)
    fn = L.pesr_conv3x3_wino4 if four else L.pesr_conv3x3_wino
    rc = fn(_p(x), _p(wp.t), _p(bias), _p(skip), _p(mask), _p(y), N, H, W, Cin, cout, alpha, act, slope,
            int(ps_out), int(ps_in), _p(ws), nws, _stream())
    _lib.check(rc, f"pesr_conv3x3_wino{'4' if four else ''}[{what} {N}x{H}x{W}x{Cin}->{cout}]")


USE_RGB_OUT = True   # tests switch it off to compare with the implicit-GEMM kernel


def rgb_out_eligible(cin: int, cout: int, stride: int) -> bool:
    """Shapes of pesr_conv3x3_rgb_out_fwd (C -> 3, stride 1)."""
    return USE_RGB_OUT and cout == 3 and stride == 1 and cin % 64 == 0 and cin <= 512


def conv3x3_fwd(x: torch.Tensor, wp: torch.Tensor, bias: Optional[torch.Tensor], cout: int, stride: int = 1,
                alpha: float = 1.0, act: int = ACT_NONE, slope: float = 0.0, skip: Optional[torch.Tensor] = None,
                mask: Optional[torch.Tensor] = None, ps_out: bool = False,
                w_oihw: Optional[torch.Tensor] = None) -> torch.Tensor:
    """wp: packed weights (callable returning them is accepted, so the pack is skipped when the direct RGB kernel runs)."""
    _chk(x, "conv3x3_fwd.x")
    N, H, W, Cin = x.shape
    OH, OW = (H - 1) // stride + 1, (W - 1) // stride + 1
    if ps_out:
        y = torch.empty((N, 2 * OH, 2 * OW, cout // 4), dtype=torch.float32, device=x.device)
    else:
        y = torch.empty((N, OH, OW, cout), dtype=torch.float32, device=x.device)
    for t, n in ((skip, "skip"), (mask, "mask")):
        if t is not None:
            _chk(t, f"conv3x3_fwd.{n}")
            assert t.shape == y.shape, (t.shape, y.shape)
    if Cin == 3 and stride == 1 and skip is None and mask is None and not ps_out and alpha == 1.0 and w_oihw is not None \
            and cout % 4 == 0 and 256 % (cout // 4) == 0:
        # RGB input layer: dedicated HBM-bound direct kernel on the un-packed OIHW weights
        FLOPS.add(18.0 * N * OH * OW * Cin * cout, 0.0, "rgb (HBM-bound, VALU)")
        br = OP_EVENTS.begin(f"conv_rgb_in 3->{cout}")
        rc = _lib.lib().pesr_conv3x3_rgb_fwd(_p(x), _p(w_oihw), _p(bias), _p(y), N, H, W, cout, act, slope, _stream())
        OP_EVENTS.end(br)
        _lib.check(rc, f"pesr_conv3x3_rgb_fwd[{N}x{H}x{W}x3->{cout}]")
        return y
    if rgb_out_eligible(Cin, cout, stride) and skip is None and mask is None and not ps_out and alpha == 1.0 and w_oihw is not None:
        # -> RGB output layer: dedicated HBM-bound kernel on the un-packed OIHW weights (no pack, no padded MFMAs)
        FLOPS.add(18.0 * N * OH * OW * Cin * cout, 1.0, "rgb (HBM-bound, MFMA)")
        br = OP_EVENTS.begin(f"conv_rgb_out {Cin}->3")
        rc = _lib.lib().pesr_conv3x3_rgb_out_fwd(_p(x), _p(w_oihw), _p(bias), _p(y), N, H, W, Cin, act, slope, _stream())
        OP_EVENTS.end(br)
        _lib.check(rc, f"pesr_conv3x3_rgb_out_fwd[{N}x{H}x{W}x{Cin}->3]")
        return y
    if callable(wp):
        wp = wp()
    if FLOPS.on:
        FLOPS.add(18.0 * N * OH * OW * Cin * cout, *_conv_family(wp))
    br = KERNEL_EVENTS.begin("fwd", N, H, W, Cin, cout, stride)
    L = _lib.lib()
    if isinstance(wp, Bf16Packed) and stride == 2:
        assert not ps_out
        rc = L.pesr_conv3x3_bf16_s2(_p(x), _p(wp.t), _p(bias), _p(skip), _p(mask), _p(y), N, H, W, Cin, cout, alpha, act, slope, _stream())
    elif isinstance(wp, (WinoPacked, Wino4Packed, Bf16Packed, Bf16x3Packed)):
        assert stride == 1
        _conv3x3_wino(x, wp, bias, skip, mask, y, N, H, W, Cin, cout, alpha, act, slope, "fwd", ps_out=ps_out)
        rc = 0
    else:
        nws = L.pesr_conv3x3_workspace_bytes(N, OH, OW, cout)
        ws = workspace(nws, x.device) if nws else None
        rc = L.pesr_conv3x3_fwd(_p(x), _p(wp), _p(bias), _p(skip), _p(mask), _p(y), N, H, W, Cin, cout, stride,
                                alpha, act, slope, int(ps_out), _p(ws), nws, _stream())
    KERNEL_EVENTS.end(br)
    _lib.check(rc, f"pesr_conv3x3_fwd[{N}x{H}x{W}x{Cin}->{cout},s{stride}]")
    return y


def conv3x3_dgrad(dy: torch.Tensor, wpd: torch.Tensor, in_shape, stride: int = 1, alpha: float = 1.0,
                  mask: Optional[torch.Tensor] = None, skip: Optional[torch.Tensor] = None,
                  ps_in: bool = False) -> torch.Tensor:
    """dx for a conv whose forward input had NHWC shape `in_shape`; dy is the (possibly shuffled) output grad."""
    _chk(dy, "conv3x3_dgrad.dy")
    N, H, W, Cin = in_shape
    cout = dy.shape[3] * (4 if ps_in else 1)
    dx = torch.empty((N, H, W, Cin), dtype=torch.float32, device=dy.device)
    for t, n in ((skip, "skip"), (mask, "mask")):
        if t is not None:
            _chk(t, f"conv3x3_dgrad.{n}")
            assert t.shape == dx.shape
    L = _lib.lib()
    if FLOPS.on:
        FLOPS.add(18.0 * N * ((H - 1) // stride + 1) * ((W - 1) // stride + 1) * Cin * cout, *_conv_family(wpd))
    br = KERNEL_EVENTS.begin("dgrad", N, H, W, Cin, cout, stride)
    if isinstance(wpd, Bf16Packed) and stride == 2:
        assert not ps_in
        rc = L.pesr_conv3x3_bf16_s2_dgrad(_p(dy), _p(wpd.t), _p(mask), _p(skip), _p(dx), N, H, W, cout, Cin, alpha, _stream())
        KERNEL_EVENTS.end(br)
        _lib.check(rc, f"pesr_conv3x3_bf16_s2_dgrad[{N}x{H}x{W}x{Cin}<-{cout}]")
        return dx
    if isinstance(wpd, (WinoPacked, Wino4Packed, Bf16Packed, Bf16x3Packed)):     # the input gradient is the conv of dy with the flipped, transposed kernel
        assert stride == 1
        _conv3x3_wino(dy, wpd, None, skip, mask, dx, N, H, W, cout, Cin, alpha, ACT_NONE, 0.0, "dgrad", ps_in=ps_in)
        KERNEL_EVENTS.end(br)
        return dx
    nws = L.pesr_conv3x3_workspace_bytes(N, H, W, Cin) if stride == 1 else 0
    ws = workspace(nws, dy.device) if nws else None
    rc = L.pesr_conv3x3_dgrad(_p(dy), _p(wpd), _p(mask), _p(skip), _p(dx), N, H, W, Cin, cout, stride, alpha,
                              int(ps_in), _p(ws), nws, _stream())
    KERNEL_EVENTS.end(br)
    _lib.check(rc, f"pesr_conv3x3_dgrad[{N}x{H}x{W}x{Cin}<-{cout},s{stride}]")
    return dx


USE_RGB_IN_DGRAD = True   # tests switch it off to compare with the implicit-GEMM path


def rgb_in_dgrad_eligible(cin: int, cout: int, stride: int) -> bool:
    """Shapes of pesr_conv3x3_rgb_in_dgrad (input gradient of a 3 -> C conv, stride 1)."""
    return USE_RGB_IN_DGRAD and cin == 3 and stride == 1 and cout % 64 == 0 and cout <= 512


def conv3x3_rgb_in_dgrad(dy: torch.Tensor, w_oihw: torch.Tensor, in_shape) -> torch.Tensor:
    """dx of a 3 -> C conv (stride 1, no fused mask): dy [N,H,W,C], w OIHW [C,3,3,3] -> dx [N,H,W,3] (HBM-bound: reads dy once)."""
    _chk(dy, "conv3x3_rgb_in_dgrad.dy"); _chk(w_oihw, "conv3x3_rgb_in_dgrad.w")
    N, H, W, three = in_shape
    C = dy.shape[3]
    assert three == 3 and dy.shape == (N, H, W, C) and w_oihw.shape == (C, 3, 3, 3)
    dx = torch.empty((N, H, W, 3), dtype=torch.float32, device=dy.device)
    FLOPS.add(18.0 * N * H * W * C * 3, 1.0, "rgb (HBM-bound, MFMA)")
    br = OP_EVENTS.begin(f"conv_rgb_in_dgrad {C}->3")
    rc = _lib.lib().pesr_conv3x3_rgb_in_dgrad(_p(dy), _p(w_oihw), _p(dx), N, H, W, C, _stream())
    OP_EVENTS.end(br)
    _lib.check(rc, f"pesr_conv3x3_rgb_in_dgrad[{N}x{H}x{W}x3<-{C}]")
    return dx


def conv3x3_rgb_dgrad(dy: torch.Tensor, w_oihw: torch.Tensor, in_shape) -> torch.Tensor:
    """dx of a C -> 3 conv (stride 1, no fused mask): dy [N,H,W,3], w OIHW [3,C,3,3] -> dx [N,H,W,C]."""
    _chk(dy, "conv3x3_rgb_dgrad.dy"); _chk(w_oihw, "conv3x3_rgb_dgrad.w")
    N, H, W, C = in_shape
    assert dy.shape == (N, H, W, 3) and w_oihw.shape == (3, C, 3, 3)
    dx = torch.empty((N, H, W, C), dtype=torch.float32, device=dy.device)
    FLOPS.add(18.0 * N * H * W * C * 3, 0.0, "rgb (HBM-bound, VALU)")
    rc = _lib.lib().pesr_conv3x3_rgb_dgrad(_p(dy), _p(w_oihw), _p(dx), N, H, W, C, _stream())
    _lib.check(rc, f"pesr_conv3x3_rgb_dgrad[{N}x{H}x{W}x{C}<-3]")
    return dx


def _out(t, shape, device):
    """Use the caller's output tensor (a view into a flat gradient buffer) when given, else allocate."""
    if t is not None:
        assert tuple(t.shape) == tuple(shape) and t.is_contiguous() and t.dtype == torch.float32
        return t
    return torch.empty(shape, dtype=torch.float32, device=device)


WGRAD_AUTO, WGRAD_DIRECT, WGRAD_WINO23, WGRAD_WINO4_16X16, WGRAD_WINO4_1D, WGRAD_WINO4_12W = 0, 1, 2, 3, 4, 5      # include/pesr_hip.h PESR_WGRAD_*
# PESR_WGRAD_WINO4_16X16=1: round 2's 16x16x4-MFMA form of the F(4,3) weight gradient instead of the 32x32x2 form (A/B switch)
USE_WGRAD_WINO4_16X16 = __import__("os").environ.get("PESR_WGRAD_WINO4_16X16", "0") == "1"
# PESR_WGRAD_WINO4_1D=1: round 3's 1-D F(4,3) transform on the 32x32x2 kernel instead of the y-nested one (A/B switch)
USE_WGRAD_WINO4_1D = __import__("os").environ.get("PESR_WGRAD_WINO4_1D", "0") == "1"


def wgrad_bf16_eligible(N: int, H: int, W: int, Cin: int, Cout: int, stride: int = 1, ps_in: bool = False) -> bool:
    """PRECISION is "bf16" and the bf16 weight-gradient kernel covers the shape (and has at least ~one round of workgroups)."""
    if PRECISION != "bf16" or stride != 1 or W % 48 or Cin % 64 or Cout % 128 or (ps_in and Cout % 512):
        return False
    if _lib.lib().pesr_conv3x3_wgrad_bf16_workspace_bytes(N, H, W, Cin, Cout) == 0:
        return False
    return N * ((H + 1) // 2) * (W // 48) >= (96 if BF16_MIN_WGS >= 64 else 1)


def conv3x3_wgrad_bf16(x: torch.Tensor, dy: torch.Tensor, alpha: float = 1.0, want_bias: bool = True, ps_in: bool = False,
                       dw_out=None, db_out=None, accumulate: bool = False):
    """(dw, db) on the bf16 MFMA: operands rounded to bf16, fp32 sums; db from the un-rounded dy."""
    assert not accumulate or (dw_out is not None and (db_out is not None or not want_bias))
    _chk(x, "conv3x3_wgrad_bf16.x"); _chk(dy, "conv3x3_wgrad_bf16.dy")
    N, H, W, Cin = x.shape
    cout = dy.shape[3] * (4 if ps_in else 1)
    L = _lib.lib()
    nbytes = L.pesr_conv3x3_wgrad_bf16_workspace_bytes(N, H, W, Cin, cout)
    if nbytes == 0:
        raise _lib.PesrHipError(f"pesr_conv3x3_wgrad_bf16: unsupported shape {N}x{H}x{W} Cin={Cin} Cout={cout}")
    ws = workspace(nbytes, x.device)
    dw = _out(dw_out, (cout, Cin, 3, 3), x.device)
    db = _out(db_out, (cout,), x.device) if want_bias else None
    if FLOPS.on:
        FLOPS.add(18.0 * N * H * W * Cin * cout, 1.0, "bf16")
    br = KERNEL_EVENTS.begin("wgrad", N, H, W, Cin, cout, 1)
    rc = L.pesr_conv3x3_wgrad_bf16(_p(x), _p(dy), _p(dw), _p(db), N, H, W, Cin, cout, alpha, int(ps_in), int(accumulate), _p(ws),
                                   ws.numel(), _stream())
    KERNEL_EVENTS.end(br)
    _lib.check(rc, f"pesr_conv3x3_wgrad_bf16[{N}x{H}x{W}x{Cin}->{cout}]")
    return dw, db


def conv3x3_wgrad(x: torch.Tensor, dy: torch.Tensor, stride: int = 1, alpha: float = 1.0, want_bias: bool = True,
                  ps_in: bool = False, dw_out=None, db_out=None, algo=None, accumulate: bool = False):
    """(dw [O, I, 3, 3], db [O] | None).  algo: None = by the PESR_* switches (default: auto = F(4,3) where it applies, else
    F(2,3), else direct), or one of WGRAD_AUTO / WGRAD_DIRECT / WGRAD_WINO23.  accumulate: add to dw_out / db_out (which then
    must be given) instead of overwriting them."""
    assert not accumulate or (dw_out is not None and (db_out is not None or not want_bias))
    _chk(x, "conv3x3_wgrad.x")
    _chk(dy, "conv3x3_wgrad.dy")
    N, H, W, Cin = x.shape
    cout = dy.shape[3] * (4 if ps_in else 1)
    if algo is None and wgrad_bf16_eligible(N, H, W, Cin, cout, stride, ps_in):
        return conv3x3_wgrad_bf16(x, dy, alpha, want_bias, ps_in, dw_out, db_out, accumulate)
    L = _lib.lib()
    if algo is None:
        algo = ((WGRAD_WINO4_16X16 if USE_WGRAD_WINO4_16X16 else (WGRAD_WINO4_1D if USE_WGRAD_WINO4_1D else WGRAD_AUTO))
                if (USE_WINO4 and USE_WGRAD_WINO4) else 2) if USE_WGRAD_WINO else 1
    nbytes = L.pesr_conv3x3_wgrad_workspace_bytes(N, H, W, Cin, cout, stride, algo)
    if nbytes == 0:
        raise _lib.PesrHipError(f"pesr_conv3x3_wgrad: unsupported shape Cin={Cin} Cout={cout} stride={stride}")
    ws = workspace(nbytes, x.device)
    dw = _out(dw_out, (cout, Cin, 3, 3), x.device)
    db = _out(db_out, (cout,), x.device) if want_bias else None
    if FLOPS.on:
        name, frac = wgrad_kernel_for(N, H, W, Cin, cout) if (stride == 1 and algo != WGRAD_DIRECT) else ("conv3x3_wgrad_kernel", 1.0)
        if algo == WGRAD_WINO23 and frac == 0.5:
            frac = 2.0 / 3.0
        FLOPS.add(18.0 * N * ((H - 1) // stride + 1) * ((W - 1) // stride + 1) * Cin * cout, frac,
                  {0.5: "F(4,3)", 1.0: "direct", 1.0 / 3.0: "F(2,3)y x F(4,3)x"}.get(frac, "F(2,3)"))
    br = KERNEL_EVENTS.begin("wgrad", N, H, W, Cin, cout, stride)
    rc = L.pesr_conv3x3_wgrad(_p(x), _p(dy), _p(dw), _p(db), N, H, W, Cin, cout, stride, alpha, int(ps_in), algo, int(accumulate),
                              _p(ws), ws.numel(), _stream())
    KERNEL_EVENTS.end(br)
    _lib.check(rc, f"pesr_conv3x3_wgrad[{N}x{H}x{W}x{Cin}->{cout},s{stride}]")
    return dw, db


def conv3x3_wgrad_rgb(a: torch.Tensor, b3: torch.Tensor, mode: int, alpha: float = 1.0, want_bias: bool = True,
                      dw_out=None, db_out=None, accumulate: bool = False):
    """Weight grad of a conv with a 3-channel side. mode 0: a=dy [N,H,W,C], b3=x -> dw [C,3,3,3]; mode 1: a=x, b3=dy -> dw [3,C,3,3]."""
    _chk(a, "conv3x3_wgrad_rgb.a")
    _chk(b3, "conv3x3_wgrad_rgb.b3")
    N, H, W, C = a.shape
    assert b3.shape == (N, H, W, 3)
    L = _lib.lib()
    nbytes = L.pesr_conv3x3_wgrad_rgb_workspace_bytes(N, H, W, C)
    if nbytes == 0:
        raise _lib.PesrHipError(f"pesr_conv3x3_wgrad_rgb: unsupported channel count {C}")
    ws = workspace(nbytes, a.device)
    dw = _out(dw_out, (C, 3, 3, 3) if mode == 0 else (3, C, 3, 3), a.device)
    db = _out(db_out, (C if mode == 0 else 3,), a.device) if want_bias else None
    FLOPS.add(18.0 * N * H * W * C * 3, 1.0 if C % 256 == 0 else 0.0, "rgb (HBM-bound, MFMA)" if C % 256 == 0 else "rgb (HBM-bound, VALU)")
    assert not accumulate or (dw_out is not None and not want_bias)
    rc = L.pesr_conv3x3_wgrad_rgb(_p(a), _p(b3), _p(dw), _p(db), N, H, W, C, mode, alpha, int(accumulate), _p(ws), ws.numel(), _stream())
    _lib.check(rc, "pesr_conv3x3_wgrad_rgb")
    return dw, db


# ------------------------------------------------------------------------------------------------
# MeanShift / PixelShuffle / masks / pooling
# ------------------------------------------------------------------------------------------------
def meanshift_fwd(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, x_nchw: bool = False, y_nchw: bool = False):
    """x: [N,H,W,3] (or [N,3,H,W] contiguous when x_nchw) -> y in NHWC (or NCHW when y_nchw)."""
    _chk(x, "meanshift_fwd.x")
    if x_nchw:
        N, _, H, W = x.shape
    else:
        N, H, W, _ = x.shape
    y = torch.empty((N, 3, H, W) if y_nchw else (N, H, W, 3), dtype=torch.float32, device=x.device)
    br = OP_EVENTS.begin(f"meanshift {H}x{W}")
    rc = _lib.lib().pesr_meanshift_fwd(_p(x), _p(w), _p(b), _p(y), N, H, W, int(x_nchw), int(y_nchw), _stream())
    OP_EVENTS.end(br)
    _lib.check(rc, "pesr_meanshift_fwd")
    return y


def meanshift_bwd(dy: torch.Tensor, x: torch.Tensor, w: torch.Tensor, x_nchw: bool = False, need_dx: bool = True,
                  dw_out=None, db_out=None):
    _chk(dy, "meanshift_bwd.dy")
    N, H, W, _ = dy.shape
    dx = torch.empty_like(dy) if need_dx else None
    dw = _out(dw_out, (3, 3, 1, 1), dy.device)
    db = _out(db_out, (3,), dy.device)
    ws = workspace(1024 * 12 * 4 + 128, dy.device)
    rc = _lib.lib().pesr_meanshift_bwd(_p(dy), _p(x), _p(w), _p(dx), _p(dw), _p(db), N, H, W, int(x_nchw), _p(ws),
                                       ws.numel(), _stream())
    _lib.check(rc, "pesr_meanshift_bwd")
    return dx, dw, db


def pixel_shuffle_fwd(x: torch.Tensor) -> torch.Tensor:
    _chk(x, "pixel_shuffle_fwd.x")
    N, H, W, C4 = x.shape
    y = torch.empty((N, 2 * H, 2 * W, C4 // 4), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().pesr_pixel_shuffle_fwd(_p(x), _p(y), N, H, W, C4 // 4, _stream()), "pesr_pixel_shuffle_fwd")
    return y


def pixel_shuffle_bwd(dy: torch.Tensor) -> torch.Tensor:
    _chk(dy, "pixel_shuffle_bwd.dy")
    N, H2, W2, C = dy.shape
    dx = torch.empty((N, H2 // 2, W2 // 2, 4 * C), dtype=torch.float32, device=dy.device)
    _lib.check(_lib.lib().pesr_pixel_shuffle_bwd(_p(dy), _p(dx), N, H2 // 2, W2 // 2, C, _stream()), "pesr_pixel_shuffle_bwd")
    return dx


def relu_mask(g: torch.Tensor, ref: Optional[torch.Tensor] = None, add: Optional[torch.Tensor] = None,
              alpha: float = 1.0, slope: float = 0.0) -> torch.Tensor:
    _chk(g, "relu_mask.g")
    out = torch.empty_like(g)
    _lib.check(_lib.lib().pesr_relu_mask(_p(g), _p(ref), _p(add), _p(out), g.numel(), alpha, slope, _stream()), "pesr_relu_mask")
    return out


def maxpool2x2_fwd(x: torch.Tensor) -> torch.Tensor:
    _chk(x, "maxpool2x2_fwd.x")
    N, H, W, C = x.shape
    y = torch.empty((N, H // 2, W // 2, C), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().pesr_maxpool2x2_fwd(_p(x), _p(y), N, H, W, C, _stream()), "pesr_maxpool2x2_fwd")
    return y


def maxpool2x2_bwd(x: torch.Tensor, dy: torch.Tensor, relu_in: bool) -> torch.Tensor:
    _chk(dy, "maxpool2x2_bwd.dy")
    N, H, W, C = x.shape
    dx = torch.empty_like(x)
    _lib.check(_lib.lib().pesr_maxpool2x2_bwd(_p(x), _p(dy), _p(dx), N, H, W, C, int(relu_in), _stream()), "pesr_maxpool2x2_bwd")
    return dx


# ------------------------------------------------------------------------------------------------
# BatchNorm(train) + LeakyReLU
# ------------------------------------------------------------------------------------------------
def bn_lrelu_fwd(x, gamma, beta, running_mean, running_var, num_batches, eps=1e-5, momentum=0.1, slope=0.2,
                 y_nchw=False):
    _chk(x, "bn_lrelu_fwd.x")
    N, H, W, C = x.shape
    L = _lib.lib()
    ws = workspace(L.pesr_bn_workspace_bytes(N * H * W, C), x.device)
    y = torch.empty((N, C, H, W) if y_nchw else (N, H, W, C), dtype=torch.float32, device=x.device)
    stats = torch.empty((2, C), dtype=torch.float32, device=x.device)
    rc = L.pesr_bn_lrelu_fwd(_p(x), _p(gamma), _p(beta), _p(y), _p(stats), _p(running_mean), _p(running_var),
                             _p(num_batches), N, H, W, C, eps, momentum, slope, int(y_nchw), _p(ws), ws.numel(), _stream())
    _lib.check(rc, "pesr_bn_lrelu_fwd")
    return y, stats


def conv_rgb_bn_eligible(cin: int, cout: int, stride: int) -> bool:
    """Shapes of pesr_conv3x3_rgb_bn_lrelu_fwd (3 -> C conv whose kernel also leaves the BatchNorm partial sums)."""
    return cin == 3 and stride == 1 and cout % 4 == 0 and 256 % (cout // 4) == 0


def conv_rgb_bn_lrelu_fwd(x, w_oihw, gamma, beta, running_mean, running_var, num_batches, eps=1e-5, momentum=0.1, slope=0.2,
                          y_nchw=False):
    """(z, y, stats): z = conv3x3(x [N,H,W,3], w) without bias, y = act(bn_train(z)); the statistics come out of the conv kernel's
    epilogue (no pass over z for them)."""
    _chk(x, "conv_rgb_bn_lrelu_fwd.x"); _chk(w_oihw, "conv_rgb_bn_lrelu_fwd.w")
    N, H, W, Cin = x.shape
    C = w_oihw.shape[0]
    assert tuple(w_oihw.shape) == (C, 3, 3, 3) and conv_rgb_bn_eligible(Cin, C, 1)
    L = _lib.lib()
    ws = workspace(L.pesr_conv3x3_rgb_bn_workspace_bytes(N, H, W, C), x.device)
    z = torch.empty((N, H, W, C), dtype=torch.float32, device=x.device)
    y = torch.empty((N, C, H, W) if y_nchw else (N, H, W, C), dtype=torch.float32, device=x.device)
    stats = torch.empty((2, C), dtype=torch.float32, device=x.device)
    FLOPS.add(18.0 * N * H * W * Cin * C, 0.0, "rgb (HBM-bound, VALU)")
    rc = L.pesr_conv3x3_rgb_bn_lrelu_fwd(_p(x), _p(w_oihw), _p(z), _p(gamma), _p(beta), _p(y), _p(stats), _p(running_mean), _p(running_var),
                                         _p(num_batches), N, H, W, C, eps, momentum, slope, int(y_nchw), _p(ws), ws.numel(), _stream())
    _lib.check(rc, f"pesr_conv3x3_rgb_bn_lrelu_fwd[{N}x{H}x{W}x3->{C}]")
    return z, y, stats


# ---- convs whose epilogue leaves the BatchNorm sums (round 6; include/pesr_hip.h PesrBnFuse) -------------------------------------------
class _BnFuse(__import__("ctypes").Structure):
    _fields_ = [("mode", __import__("ctypes").c_int), ("rows", __import__("ctypes").c_int), ("part", __import__("ctypes").c_void_p),
                ("z", __import__("ctypes").c_void_p), ("mean_invstd", __import__("ctypes").c_void_p), ("gamma", __import__("ctypes").c_void_p),
                ("beta", __import__("ctypes").c_void_p), ("slope", __import__("ctypes").c_float)]


BN_FWD_STATS, BN_BWD_MASK_SUMS = 1, 2
USE_BN_FUSE = __import__("os").environ.get("PESR_BN_FUSE", "1") != "0"     # PESR_BN_FUSE=0: the un-fused BatchNorm passes everywhere (A/B switch)
_BN_ROWS = {}


def conv_bn_rows(which: int, N: int, H: int, W: int, Cin: int, Cout: int, stride: int = 1) -> int:
    """Rows of BatchNorm sums the fused call `which` (0 = conv3x3_fwd_bn on direct packing, 1 = conv3x3_dgrad_bn on direct dgrad packing,
    2 = the F(4,3) kernel) leaves for this problem; 0: not covered (split-K layers, odd channel counts) - use the un-fused calls."""
    if not USE_BN_FUSE:
        return 0
    key = (which, N, H, W, Cin, Cout, stride)
    r = _BN_ROWS.get(key)
    if r is None:
        r = _BN_ROWS[key] = int(_lib.lib().pesr_conv3x3_bn_rows(which, N, H, W, Cin, Cout, stride))
    return r


def _fuse_struct(mode, part, z=None, stats=None, gamma=None, beta=None, slope=0.0):
    f = _BnFuse()
    f.mode, f.rows, f.part = mode, part.shape[0], part.data_ptr()
    f.z, f.mean_invstd, f.gamma, f.beta, f.slope = _p(z), _p(stats), _p(gamma), _p(beta), float(slope)
    return f


def conv3x3_fwd_bn_stats(x: torch.Tensor, wp, bias: Optional[torch.Tensor], cout: int, stride: int = 1):
    """z = conv(x) (+ bias) with the per-pixel-tile sums of z and z^2 left by the kernel's epilogue -> (z, part [rows, 2, cout]) or None when
    the fused form does not cover this problem / packing (the caller then runs conv3x3_fwd + bn_lrelu_fwd)."""
    import ctypes
    _chk(x, "conv3x3_fwd_bn_stats.x")
    N, H, W, Cin = x.shape
    four = isinstance(wp, Wino4Packed)
    if not four and not torch.is_tensor(wp):
        return None
    rows = conv_bn_rows(2, N, H, W, Cin, cout, 1) if four else conv_bn_rows(0, N, H, W, Cin, cout, stride)
    if rows == 0:
        return None
    OH, OW = (H - 1) // stride + 1, (W - 1) // stride + 1
    L = _lib.lib()
    z = torch.empty((N, OH, OW, cout), dtype=torch.float32, device=x.device)
    part = torch.empty((rows, 2, cout), dtype=torch.float32, device=x.device)
    f = _fuse_struct(BN_FWD_STATS, part)
    nws = L.pesr_conv3x3_workspace_bytes(N, OH, OW, cout)
    ws = workspace(nws, x.device) if nws else None
    if FLOPS.on:
        FLOPS.add(18.0 * N * OH * OW * Cin * cout, *_conv_family(wp))
    br = KERNEL_EVENTS.begin("fwd", N, H, W, Cin, cout, stride)
    if four:
        rc = L.pesr_conv3x3_wino4_bn(_p(x), _p(wp.t), _p(bias), _p(z), N, H, W, Cin, cout, _p(ws), nws, ctypes.byref(f), _stream())
    else:
        rc = L.pesr_conv3x3_fwd_bn(_p(x), _p(wp), _p(bias), _p(z), N, H, W, Cin, cout, stride, _p(ws), nws, ctypes.byref(f), _stream())
    KERNEL_EVENTS.end(br)
    _lib.check(rc, f"pesr_conv3x3_{'wino4' if four else 'fwd'}_bn[{N}x{H}x{W}x{Cin}->{cout},s{stride}]")
    return z, part


def bn_finalize_apply(z, part, gamma, beta, running_mean, running_var, num_batches, eps=1e-5, momentum=0.1, slope=0.2, y_nchw=False):
    """The BatchNorm of z from the sums a conv kernel left (conv3x3_fwd_bn_stats): finalize + apply -> (y, stats [2, C])."""
    N, H, W, C = z.shape
    L = _lib.lib()
    stats = torch.empty((2, C), dtype=torch.float32, device=z.device)
    _lib.check(L.pesr_bn_finalize(_p(part), part.shape[0], C, N * H * W, eps, momentum, _p(stats), _p(running_mean), _p(running_var),
                                  _p(num_batches), _stream()), "pesr_bn_finalize")
    y = torch.empty((N, C, H, W) if y_nchw else (N, H, W, C), dtype=torch.float32, device=z.device)
    _lib.check(L.pesr_bn_lrelu_eval_fwd(_p(z), _p(gamma), _p(beta), _p(stats), _p(y), N, H, W, C, slope, int(y_nchw), _stream()), "pesr_bn_lrelu_eval_fwd")
    return y, stats


def conv3x3_dgrad_bn_sums(dy: torch.Tensor, wpd, in_shape, stride, z, stats, gamma, beta, slope):
    """The input gradient dx of a conv whose INPUT was y = lrelu(bn(z)): the kernel stores g' = dx * lrelu'(bn(z)) and leaves the sums of g'
    and g' * xhat per pixel tile -> (g' [in_shape], part [rows, 2, C]) or None when the fused form does not cover the problem / packing."""
    import ctypes
    _chk(dy, "conv3x3_dgrad_bn_sums.dy")
    N, H, W, Cin = in_shape
    cout = dy.shape[3]
    four = isinstance(wpd, Wino4Packed)
    if not four and not torch.is_tensor(wpd):
        return None
    rows = conv_bn_rows(2, N, H, W, cout, Cin, 1) if four else conv_bn_rows(1, N, H, W, Cin, cout, stride)
    if rows == 0:
        return None
    assert tuple(z.shape) == tuple(in_shape) and z.is_contiguous()
    L = _lib.lib()
    g = torch.empty((N, H, W, Cin), dtype=torch.float32, device=dy.device)
    part = torch.empty((rows, 2, Cin), dtype=torch.float32, device=dy.device)
    f = _fuse_struct(BN_BWD_MASK_SUMS, part, z, stats, gamma, beta, slope)
    nws = L.pesr_conv3x3_workspace_bytes(N, H, W, Cin) if stride == 1 else 0
    ws = workspace(nws, dy.device) if nws else None
    if FLOPS.on:
        FLOPS.add(18.0 * N * ((H - 1) // stride + 1) * ((W - 1) // stride + 1) * Cin * cout, *_conv_family(wpd))
    br = KERNEL_EVENTS.begin("dgrad", N, H, W, Cin, cout, stride)
    if four:
        rc = L.pesr_conv3x3_wino4_bn(_p(dy), _p(wpd.t), None, _p(g), N, H, W, cout, Cin, _p(ws), nws, ctypes.byref(f), _stream())
    else:
        rc = L.pesr_conv3x3_dgrad_bn(_p(dy), _p(wpd), _p(g), N, H, W, Cin, cout, stride, _p(ws), nws, ctypes.byref(f), _stream())
    KERNEL_EVENTS.end(br)
    _lib.check(rc, f"pesr_conv3x3_{'wino4' if four else 'dgrad'}_bn[{N}x{H}x{W}x{Cin}<-{cout},s{stride}]")
    return g, part


def bn_lrelu_bwd_fused(z, g_masked, part, gamma, beta, stats, need_param_grads=True, dgamma_out=None, dbeta_out=None, accumulate=False):
    """BatchNorm backward from the sums the producing conv kernel left (conv3x3_dgrad_bn_sums): -> (dz, dgamma, dbeta)."""
    _chk(g_masked, "bn_lrelu_bwd_fused.g")
    assert not accumulate or (dgamma_out is not None and dbeta_out is not None)
    N, H, W, C = z.shape
    L = _lib.lib()
    ws = workspace(2 * C * 4 + 256, z.device)
    dz = torch.empty_like(z)
    dgamma = _out(dgamma_out, (C,), z.device) if need_param_grads else None
    dbeta = _out(dbeta_out, (C,), z.device) if need_param_grads else None
    rc = L.pesr_bn_lrelu_bwd_fused(_p(z), _p(g_masked), _p(part), part.shape[0], _p(gamma), _p(beta), _p(stats), _p(dz), _p(dgamma), _p(dbeta),
                                   N, H, W, C, int(accumulate), _p(ws), ws.numel(), _stream())
    _lib.check(rc, "pesr_bn_lrelu_bwd_fused")
    return dz, dgamma, dbeta


def bn_lrelu_bwd(x, dy, gamma, beta, stats, slope=0.2, dy_nchw=False, need_param_grads=True, dgamma_out=None, dbeta_out=None,
                 accumulate=False):
    """accumulate: add to dgamma_out / dbeta_out (which then must be given) instead of overwriting them."""
    _chk(dy, "bn_lrelu_bwd.dy")
    assert not accumulate or (dgamma_out is not None and dbeta_out is not None)
    N, H, W, C = x.shape
    L = _lib.lib()
    ws = workspace(L.pesr_bn_workspace_bytes(N * H * W, C), x.device)
    dx = torch.empty_like(x)
    dgamma = _out(dgamma_out, (C,), x.device) if need_param_grads else None
    dbeta = _out(dbeta_out, (C,), x.device) if need_param_grads else None
    rc = L.pesr_bn_lrelu_bwd(_p(x), _p(dy), _p(gamma), _p(beta), _p(stats), _p(dx), _p(dgamma), _p(dbeta), N, H, W, C, slope,
                             int(dy_nchw), int(accumulate), _p(ws), ws.numel(), _stream())
    _lib.check(rc, "pesr_bn_lrelu_bwd")
    return dx, dgamma, dbeta


def bn_bwd_bwd(z, du, g, gamma, stats, need_du=True, need_z=True, need_gamma=True):
    """Backward of the training-mode BatchNorm backward (pesr_bn_bwd_bwd): (dL/d(du), dL/dz, dL/dgamma) for g = dL/d(dz)."""
    for t, n in ((z, "z"), (du, "du"), (g, "g")):
        _chk(t, "bn_bwd_bwd." + n)
    N, H, W, C = z.shape
    L = _lib.lib()
    ws = workspace(L.pesr_bn_bwd_bwd_workspace_bytes(N * H * W, C), z.device)
    l_du = torch.empty_like(z) if need_du else None
    l_z = torch.empty_like(z) if need_z else None
    l_ga = torch.empty((C,), dtype=torch.float32, device=z.device) if need_gamma else None
    rc = L.pesr_bn_bwd_bwd(_p(z), _p(du), _p(g), _p(gamma), _p(stats), _p(l_du), _p(l_z), _p(l_ga), N, H, W, C, _p(ws), ws.numel(), _stream())
    _lib.check(rc, "pesr_bn_bwd_bwd")
    return l_du, l_z, l_ga


def bn_eval_stats(running_mean, running_var, eps=1e-5):
    """[2, C] {mean, invstd} of an eval-mode BatchNorm2d from its running statistics (C-sized torch glue)."""
    return torch.stack([running_mean.float(), torch.rsqrt(running_var.float() + eps)]).contiguous()


def bn_lrelu_eval_fwd(x, gamma, beta, stats, slope=0.2, y_nchw=False):
    _chk(x, "bn_lrelu_eval_fwd.x")
    N, H, W, C = x.shape
    y = torch.empty((N, C, H, W) if y_nchw else (N, H, W, C), dtype=torch.float32, device=x.device)
    rc = _lib.lib().pesr_bn_lrelu_eval_fwd(_p(x), _p(gamma), _p(beta), _p(stats), _p(y), N, H, W, C, slope, int(y_nchw), _stream())
    _lib.check(rc, "pesr_bn_lrelu_eval_fwd")
    return y


def bn_lrelu_eval_bwd(x, dy, gamma, beta, stats, slope=0.2, dy_nchw=False, need_param_grads=True, dgamma_out=None, dbeta_out=None):
    _chk(dy, "bn_lrelu_eval_bwd.dy")
    N, H, W, C = x.shape
    L = _lib.lib()
    ws = workspace(L.pesr_bn_workspace_bytes(N * H * W, C), x.device)
    dx = torch.empty_like(x)
    dgamma = _out(dgamma_out, (C,), x.device) if need_param_grads else None
    dbeta = _out(dbeta_out, (C,), x.device) if need_param_grads else None
    rc = L.pesr_bn_lrelu_eval_bwd(_p(x), _p(dy), _p(gamma), _p(beta), _p(stats), _p(dx), _p(dgamma), _p(dbeta), N, H, W, C, slope,
                                  int(dy_nchw), _p(ws), ws.numel(), _stream())
    _lib.check(rc, "pesr_bn_lrelu_eval_bwd")
    return dx, dgamma, dbeta


# ------------------------------------------------------------------------------------------------
# Linear
# ------------------------------------------------------------------------------------------------
LIN_MAXM = 32     # rows per kernel call (pesr_hip.h); larger batches are walked in chunks


def linear_fwd(x, w, b, act=ACT_NONE, slope=0.0):
    _chk(x, "linear_fwd.x")
    M, K = x.shape
    Nf = w.shape[0]
    L = _lib.lib()
    y = torch.empty((M, Nf), dtype=torch.float32, device=x.device)
    FLOPS.add(2.0 * M * Nf * K, 1.0, "linear (HBM-bound, MFMA)")
    for m0 in range(0, M, LIN_MAXM):
        m = min(LIN_MAXM, M - m0)
        ws = workspace(L.pesr_linear_workspace_bytes(m, Nf, K), x.device)
        _lib.check(L.pesr_linear_fwd(_p(x[m0:m0 + m]), _p(w), _p(b), _p(y[m0:m0 + m]), m, Nf, K, act, slope, _p(ws), ws.numel(), _stream()),
                   "pesr_linear_fwd")
    return y


def linear_dgrad(dy, w, out=None):
    """out: the [M, K] tensor (or leading rows of a larger one) the gradient is written into."""
    _chk(dy, "linear_dgrad.dy")
    M, Nf = dy.shape
    K = w.shape[1]
    L = _lib.lib()
    if out is not None:
        _chk(out, "linear_dgrad.out")
        assert tuple(out.shape) == (M, K)
    dx = out if out is not None else torch.empty((M, K), dtype=torch.float32, device=dy.device)
    FLOPS.add(2.0 * M * Nf * K, 0.0, "linear (HBM-bound, VALU)")
    for m0 in range(0, M, LIN_MAXM):
        m = min(LIN_MAXM, M - m0)
        ws = workspace(L.pesr_linear_workspace_bytes(m, Nf, K), dy.device)
        _lib.check(L.pesr_linear_dgrad(_p(dy[m0:m0 + m]), _p(w), _p(dx[m0:m0 + m]), m, Nf, K, _p(ws), ws.numel(), _stream()), "pesr_linear_dgrad")
    return dx


def linear_wgrad(dy, x, want_bias=True, dw_out=None, db_out=None, accumulate=False):
    """accumulate: add to dw_out / db_out (which then must be given) instead of overwriting them."""
    _chk(dy, "linear_wgrad.dy")
    M, Nf = dy.shape
    K = x.shape[1]
    assert not accumulate or (dw_out is not None and (db_out is not None or not want_bias))
    dw = _out(dw_out, (Nf, K), dy.device)
    db = _out(db_out, (Nf,), dy.device) if want_bias else None
    FLOPS.add(2.0 * M * Nf * K, 0.0, "linear (HBM-bound, VALU)")
    for m0 in range(0, M, LIN_MAXM):
        m = min(LIN_MAXM, M - m0)
        _lib.check(_lib.lib().pesr_linear_wgrad(_p(dy[m0:m0 + m]), _p(x[m0:m0 + m]), _p(dw), _p(db), m, Nf, K, int(accumulate or m0 > 0), _stream()),
                   "pesr_linear_wgrad")
    return dw, db


# ------------------------------------------------------------------------------------------------
# losses / optimizer
# ------------------------------------------------------------------------------------------------
def loss_l1_tv(sr, hr, g_l1: float, g_tv: float, need_grad=True):
    """-> (out2 device tensor [l1_mean, tv_sum], grad [N,H,W,3] | None)"""
    _chk(sr, "loss_l1_tv.sr")
    _chk(hr, "loss_l1_tv.hr")
    N, H, W, _ = sr.shape
    out = torch.empty((2,), dtype=torch.float32, device=sr.device)
    grad = torch.empty_like(sr) if need_grad else None
    ws = workspace(16384, sr.device)
    rc = _lib.lib().pesr_loss_l1_tv_fwd_bwd(_p(sr), _p(hr), _p(grad), _p(out), N, H, W, g_l1, g_tv, _p(ws), ws.numel(), _stream())
    _lib.check(rc, "pesr_loss_l1_tv_fwd_bwd")
    return out, grad


def loss_mse(a, b, gscale: float, need_grad=True):
    _chk(a, "loss_mse.a")
    _chk(b, "loss_mse.b")
    out = torch.empty((1,), dtype=torch.float32, device=a.device)
    grad = torch.empty_like(a) if need_grad else None
    ws = workspace(16384, a.device)
    rc = _lib.lib().pesr_mse_fwd_bwd(_p(a), _p(b), _p(grad), _p(out), a.numel(), gscale, _p(ws), ws.numel(), _stream())
    _lib.check(rc, "pesr_mse_fwd_bwd")
    return out, grad


def adam_step(p, g, m, v, lr, beta1, beta2, eps, step, grad_scale=1.0):
    for t, n in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
        _chk(t, f"adam_step.{n}")
    rc = _lib.lib().pesr_adam_step(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, step, grad_scale, _stream())
    _lib.check(rc, "pesr_adam_step")


def adam_step_dev(p, g, m, v, state, beta1, beta2, eps, grad_scale=1.0):
    """Adam step whose learning rate and step count live in the 6-float device tensor `state` (include/pesr_hip.h): safe to
    capture in a hipGraph - nothing that changes from step to step is a kernel argument."""
    for t, n in ((p, "p"), (g, "g"), (m, "m"), (v, "v"), (state, "state")):
        _chk(t, f"adam_step_dev.{n}")
    assert state.numel() == 6
    rc = _lib.lib().pesr_adam_step_dev(_p(p), _p(g), _p(m), _p(v), p.numel(), _p(state), beta1, beta2, eps, grad_scale, _stream())
    _lib.check(rc, "pesr_adam_step_dev")


def psnr_y(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """Y-channel PSNR of two [1, 3, H, W] image tensors (NCHW-contiguous or channels_last) -> device double [mse, psnr]."""
    outs = []
    for t in (a, b):
        if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 4 or t.shape[0] != 1 or t.shape[1] != 3:
            raise _lib.PesrHipError("psnr_y: expected [1, 3, H, W] float32 GPU tensors")
        if t.is_contiguous():
            outs.append((t, 0))
        elif t.is_contiguous(memory_format=torch.channels_last):
            outs.append((t, 1))
        else:
            outs.append((t.contiguous(), 0))
    (ta, la), (tb, lb) = outs
    assert ta.shape == tb.shape
    H, W = ta.shape[2], ta.shape[3]
    out = torch.empty(2, dtype=torch.float64, device=ta.device)
    ws = workspace(4096, ta.device)
    rc = _lib.lib().pesr_psnr_y(ta.data_ptr(), tb.data_ptr(), out.data_ptr(), H, W, la, lb, ws.data_ptr(), ws.numel(), _stream())
    _lib.check(rc, "pesr_psnr_y")
    return out


# ------------------------------------------------------------------------------------------------
# spectral normalisation (reference model/basic.py:25; torch.nn.utils.spectral_norm semantics)
# ------------------------------------------------------------------------------------------------
def spectral_norm_fwd(w: torch.Tensor, u: torch.Tensor, v: torch.Tensor, update: bool, eps: float = 1e-12):
    """w [O, ...] -> (w_hat = w / sigma, sigma [1]); update: one power iteration first, u and v rewritten IN PLACE."""
    for t, n in ((w, "w"), (u, "u"), (v, "v")):
        _chk(t, "spectral_norm_fwd." + n)
    O = w.shape[0]
    K = w.numel() // O
    assert u.numel() == O and v.numel() == K
    L = _lib.lib()
    ws = workspace(L.pesr_spectral_norm_workspace_bytes(O, K), w.device)
    w_hat = torch.empty_like(w)
    sigma = torch.empty(1, dtype=torch.float32, device=w.device)
    rc = L.pesr_spectral_norm_fwd(_p(w), _p(u), _p(v), _p(w_hat), _p(sigma), O, K, int(update), eps, _p(ws), ws.numel(), _stream())
    _lib.check(rc, f"pesr_spectral_norm_fwd[{O}x{K}]")
    return w_hat, sigma


def spectral_norm_bwd(g: torch.Tensor, w_hat: torch.Tensor, u: torch.Tensor, v: torch.Tensor, sigma: torch.Tensor, dw_out=None,
                      accumulate: bool = False):
    """dL/dw for g = dL/d(w_hat) with the u, v, sigma of that forward."""
    _chk(g, "spectral_norm_bwd.g")
    O = g.shape[0]
    K = g.numel() // O
    L = _lib.lib()
    ws = workspace(L.pesr_spectral_norm_workspace_bytes(O, K), g.device)
    assert not accumulate or dw_out is not None
    dw = _out(dw_out, tuple(g.shape), g.device)
    rc = L.pesr_spectral_norm_bwd(_p(g), _p(w_hat), _p(u), _p(v), _p(sigma), _p(dw), O, K, int(accumulate), _p(ws), ws.numel(), _stream())
    _lib.check(rc, f"pesr_spectral_norm_bwd[{O}x{K}]")
    return dw


# ------------------------------------------------------------------------------------------------
# generic k x k conv (odd k != 3): compatibility path of reference model/basic.py:4-7
# ------------------------------------------------------------------------------------------------
def conv_kxk_fwd(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], stride: int = 1) -> torch.Tensor:
    _chk(x, "conv_kxk_fwd.x"); _chk(w, "conv_kxk_fwd.w")
    N, H, W, Cin = x.shape
    Cout, k = w.shape[0], w.shape[2]
    assert w.shape == (Cout, Cin, k, k)
    y = torch.empty((N, (H - 1) // stride + 1, (W - 1) // stride + 1, Cout), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().pesr_conv_kxk_fwd(_p(x), _p(w), _p(bias), _p(y), N, H, W, Cin, Cout, k, stride, _stream()), f"pesr_conv_kxk_fwd[k={k}]")
    return y


def conv_kxk_dgrad(dy: torch.Tensor, w: torch.Tensor, in_shape, stride: int = 1) -> torch.Tensor:
    _chk(dy, "conv_kxk_dgrad.dy"); _chk(w, "conv_kxk_dgrad.w")
    N, H, W, Cin = in_shape
    Cout, k = w.shape[0], w.shape[2]
    dx = torch.empty((N, H, W, Cin), dtype=torch.float32, device=dy.device)
    _lib.check(_lib.lib().pesr_conv_kxk_dgrad(_p(dy), _p(w), _p(dx), N, H, W, Cin, Cout, k, stride, _stream()), f"pesr_conv_kxk_dgrad[k={k}]")
    return dx


def conv_kxk_wgrad(x: torch.Tensor, dy: torch.Tensor, k: int, stride: int = 1, want_bias: bool = True, dw_out=None, db_out=None):
    _chk(x, "conv_kxk_wgrad.x"); _chk(dy, "conv_kxk_wgrad.dy")
    N, H, W, Cin = x.shape
    Cout = dy.shape[3]
    dw = _out(dw_out, (Cout, Cin, k, k), x.device)
    db = _out(db_out, (Cout,), x.device) if want_bias else None
    _lib.check(_lib.lib().pesr_conv_kxk_wgrad(_p(x), _p(dy), _p(dw), _p(db), N, H, W, Cin, Cout, k, stride, _stream()), f"pesr_conv_kxk_wgrad[k={k}]")
    return dw, db


# ------------------------------------------------------------------------------------------------
# GAN losses on the logits (reference train.py:210-213,244-253; model/focal_loss.py)
# ------------------------------------------------------------------------------------------------
GAN_TYPES = {"SGAN": 0, "RSGAN": 1, "RaSGAN": 2}


def gan_loss(pred_real: torch.Tensor, pred_fake: torch.Tensor, gan_type: str, side: int, focal: bool, gamma: float, scale: float = 1.0,
             need_real: bool = True, need_fake: bool = True):
    """-> (out [1] = scale * loss, d_real [B,1] | None, d_fake [B,1] | None): value and gradients in one launch."""
    _chk(pred_real, "gan_loss.pred_real"); _chk(pred_fake, "gan_loss.pred_fake")
    B = pred_real.numel()
    assert pred_fake.numel() == B
    out = torch.empty(1, dtype=torch.float32, device=pred_real.device)
    d_r = torch.empty_like(pred_real) if need_real else None
    d_f = torch.empty_like(pred_fake) if need_fake else None
    rc = _lib.lib().pesr_gan_loss_fwd_bwd(_p(pred_real), _p(pred_fake), B, GAN_TYPES[gan_type], side, int(focal), float(gamma), float(scale),
                                          _p(out), _p(d_r), _p(d_f), _stream())
    _lib.check(rc, f"pesr_gan_loss_fwd_bwd[{gan_type}, side {side}]")
    return out, d_r, d_f
