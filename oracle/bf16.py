"""CPU restatement of the build's OPTIONAL bf16-operand mode - test infrastructure only, like the rest of oracle/.

The reference (thangvubk/PESR) is fp32 only, so this mode has no reference file to follow; what it restates is the build's own
definition (pesr_amd/csrc/conv3x3_bf16.hip, conv3x3_wgrad_bf16.hip, pesr_amd/ops.py bf16_eligible / wgrad_bf16_eligible):
    * a stride-1 3x3 conv whose shape the bf16 kernels cover rounds BOTH operands of every product to bfloat16 (round to nearest
      even) and sums the exact products in fp32:   y  = conv(r(x), r(w)) + b
      (a stride-2 conv likewise in its forward pass and its input gradient where the stride-2 forms of the kernel cover them; its
      weight gradient stays fp32)
    * its input gradient, where the (transposed) shape is covered:   dx = conv_T(r(dy), r(w))
    * its weight gradient, where THAT kernel covers the shape:       dw = corr(r(x), r(dy));   db = sum(dy) always un-rounded
    * every other op, and every conv the rules below reject, is the fp32 arithmetic of oracle/model.py.
`enabled(True)` switches oracle.model's 3x3 convs to conv3x3(); the rules are a line-by-line Python mirror of the library's
planners (tests/test_bf16_cpu.py checks them against pesr_conv3x3_bf16_score / pesr_conv3x3_wgrad_bf16_workspace_bytes).
"""
import contextlib

import torch
import torch.nn.functional as F

from .ops import round_bf16

ON = False            # oracle.model consults this
MIN_WGS = 64          # pesr_amd.ops.BF16_MIN_WGS: fewer workgroups than this stay on the fp32 kernels


@contextlib.contextmanager
def enabled(on=True, min_wgs=None):
    global ON, MIN_WGS
    old = (ON, MIN_WGS)
    ON = on
    if min_wgs is not None:
        MIN_WGS = min_wgs
    try:
        yield
    finally:
        ON, MIN_WGS = old


def _cdiv(a, b):
    return (a + b - 1) // b


def conv_score(N, H, W, Cin, Cout, min_wgs):
    """conv3x3_bf16.hip b16_plan: per-mille of the 144-pixel tiles' area inside the image, 0 if unsupported / too few workgroups."""
    if N < 1 or H < 1 or W < 1 or Cin % 32 or Cin < 32 or Cout % 64 or H * W * Cin * 4 >= 1 << 31:
        return 0
    best = None
    for TW in range(1, 145):
        if 144 % TW:
            continue
        TR = 144 // TW
        HT, WT = TR + 2, TW + 2
        if HT * WT * 8 > 4 * 512:
            continue
        cover = _cdiv(H, TR) * TR * _cdiv(W, TW) * TW
        score = cover * 8192 + (4096 if TW % 16 else 0) + HT * WT
        if best is None or score < best[0]:
            best = (score, TR, TW)
    if best is None:
        return 0
    _, TR, TW = best
    bn = 256 if Cout % 256 == 0 else (128 if Cout % 128 == 0 else 64)
    if bn == 256 and N * _cdiv(H, TR) * _cdiv(W, TW) * (Cout // 256) < 128:
        bn = 128                                          # few tiles: 128 output channels per workgroup instead of 256
    tiles = N * _cdiv(H, TR) * _cdiv(W, TW) * (Cout // bn)
    eff = H * W / (_cdiv(H, TR) * TR * _cdiv(W, TW) * TW)
    return int(1000.0 * eff) if tiles >= min_wgs else 0


def conv_s2_score(N, H, W, Cin, Cout, min_wgs):
    """conv3x3_bf16.hip b16_plan_s2 (H, W: the INPUT's size): 144 output pixels per tile, a (2 TR + 1) x (2 TW + 1) halo of at most
    640 pixels, 128 (64) output channels per workgroup; half of min_wgs workgroups qualify."""
    if N < 1 or H < 2 or W < 2 or Cin % 32 or Cin < 32 or Cout % 64 or H * W * Cin * 4 >= 1 << 31:
        return 0
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    best = None
    for TW in range(1, 145):
        if 144 % TW:
            continue
        TR = 144 // TW
        HT, WT = 2 * TR + 1, 2 * TW + 1
        if HT * WT * 8 > 5120:
            continue
        cover = _cdiv(OH, TR) * TR * _cdiv(OW, TW) * TW
        score = cover * 8192 + (4096 if TW % 16 else 0) + HT * WT
        if best is None or score < best[0]:
            best = (score, TR, TW)
    if best is None:
        return 0
    _, TR, TW = best
    bn = 128 if Cout % 128 == 0 else 64
    tiles = N * _cdiv(OH, TR) * _cdiv(OW, TW) * (Cout // bn)
    eff = OH * OW / (_cdiv(OH, TR) * TR * _cdiv(OW, TW) * TW)
    return int(1000.0 * eff) if tiles >= (min_wgs + 1) // 2 else 0


def conv_s2_dgrad_score(N, H, W, Cout_fwd, Cin_fwd, min_wgs):
    """conv3x3_bf16.hip b16_plan_s2d (H, W: dx's size): four parity classes over dy, 144 dy positions per tile, a (TR + 1) x (TW + 1)
    halo; the four classes' workgroups together count against min_wgs."""
    if H < 1 or W < 1:
        return 0
    DH, DW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    Cin, Cout = Cout_fwd, Cin_fwd
    if N < 1 or Cin % 32 or Cin < 32 or Cout % 64 or DH * DW * Cin * 4 >= 1 << 31:
        return 0
    best = None
    for TW in range(1, 145):
        if 144 % TW:
            continue
        TR = 144 // TW
        HT, WT = TR + 1, TW + 1
        if HT * WT * 8 > 2048:
            continue
        cover = _cdiv(DH, TR) * TR * _cdiv(DW, TW) * TW
        score = cover * 8192 + (4096 if TW % 16 else 0) + HT * WT
        if best is None or score < best[0]:
            best = (score, TR, TW)
    if best is None:
        return 0
    _, TR, TW = best
    bn = 256 if Cout % 256 == 0 else (128 if Cout % 128 == 0 else 64)
    if bn == 256 and 4 * N * _cdiv(DH, TR) * _cdiv(DW, TW) * (Cout // 256) < 128:
        bn = 128
    tiles = N * _cdiv(DH, TR) * _cdiv(DW, TW) * (Cout // bn)
    eff = DH * DW / (_cdiv(DH, TR) * TR * _cdiv(DW, TW) * TW)
    return int(1000.0 * eff) if 4 * tiles >= min_wgs else 0


def conv_s2_dgrad_eligible(N, H, W, Cout_fwd, Cin_fwd):
    """pesr_amd.ops.bf16_s2_dgrad_eligible (H, W: the forward input's = dx's size)."""
    if Cout_fwd % 32 or Cin_fwd % 64:
        return False
    return conv_s2_dgrad_score(N, H, W, Cout_fwd, Cin_fwd, MIN_WGS) >= 780


def conv_eligible(N, H, W, Cin, Cout, stride=1, ps_out=False, ps_in=False):
    """pesr_amd.ops.bf16_eligible for the problem the kernel runs (a stride-1 input gradient: Cin / Cout swapped; stride 2: the forward)."""
    bn = 256 if Cout % 256 == 0 else (128 if Cout % 128 == 0 else 64)
    if stride not in (1, 2) or Cin % 32 or Cout % 64 or (ps_out and Cout % (4 * bn)) or (ps_in and Cin % 128):
        return False
    if stride == 2:
        return not (ps_out or ps_in) and conv_s2_score(N, H, W, Cin, Cout, MIN_WGS) >= 780
    return conv_score(N, H, W, Cin, Cout, MIN_WGS) >= 780


def wgrad_eligible(N, H, W, Cin, Cout, stride=1, ps_in=False):
    """pesr_amd.ops.wgrad_bf16_eligible."""
    if stride != 1 or W % 48 or Cin % 64 or Cout % 128 or (ps_in and Cout % 512):
        return False
    if H * W * Cin * 4 >= 1 << 30 or H * W * Cout * 4 >= 1 << 30:
        return False
    return N * ((H + 1) // 2) * (W // 48) >= (96 if MIN_WGS >= 64 else 1)


class _Bf16Conv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, stride, f_fwd, f_dgrad, f_wgrad):
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, f_dgrad, f_wgrad, b is not None)
        if f_fwd:
            return F.conv2d(round_bf16(x), round_bf16(w), b, stride=stride, padding=1)
        return F.conv2d(x, w, b, stride=stride, padding=1)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        stride, f_dgrad, f_wgrad, has_b = ctx.cfg
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.nn.grad.conv2d_input(x.shape, round_bf16(w) if f_dgrad else w, round_bf16(gy) if f_dgrad else gy,
                                            stride=stride, padding=1)
        if ctx.needs_input_grad[1]:
            dw = torch.nn.grad.conv2d_weight(round_bf16(x) if f_wgrad else x, w.shape, round_bf16(gy) if f_wgrad else gy,
                                             stride=stride, padding=1)
        if has_b and ctx.needs_input_grad[2]:
            db = gy.sum(dim=(0, 2, 3))
        return dx, dw, db, None, None, None, None


def conv3x3(x, w, b=None, stride=1, ps=False):
    """A 3x3 conv (padding 1) of oracle.model in the bf16 mode.  ps: the conv feeds nn.PixelShuffle(2) (model/basic.py:56-59), which
    the build fuses into the kernels - the fused forms have stricter channel conditions."""
    N, Cin, H, W = x.shape
    Cout = w.shape[0]
    f_fwd = conv_eligible(N, H, W, Cin, Cout, stride, ps_out=ps)
    f_dgrad = conv_eligible(N, H, W, Cout, Cin, 1, ps_in=ps) if stride == 1 else (not ps and conv_s2_dgrad_eligible(N, H, W, Cout, Cin))
    f_wgrad = wgrad_eligible(N, H, W, Cin, Cout, stride, ps_in=ps)
    if not (f_fwd or f_dgrad or f_wgrad):
        return F.conv2d(x, w, b, stride=stride, padding=1)
    return _Bf16Conv.apply(x, w, b, stride, f_fwd, f_dgrad, f_wgrad)
