"""GPU parity of the OPTIONAL bf16-operand mode: the bf16 MFMA conv kernels against THEIR oracle - the same
bf16 rounding of both operands, sums in float64 (oracle/ops.py conv3x3_bf16*) - to fp32-accumulation accuracy (1e-5 of the
maximum), and against the fp32 oracle at the accuracy bf16 operands give (a few 1e-3)."""
import pytest
import torch

from oracle import ops as O

pytestmark = pytest.mark.gpu


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous().cuda()


def _nchw(y):
    return y.permute(0, 3, 1, 2).cpu()


def _close(a, b, rel):
    scale = b.abs().max().item() + 1e-30
    err = (a - b).abs().max().item()
    assert err <= rel * scale, f"max err {err:.3e} vs scale {scale:.3e} (rel {err / scale:.3e} > {rel})"


BF16_CASES = [
    # N, H, W, Cin, Cout
    (2, 48, 48, 256, 256),     # G body
    (1, 7, 48, 64, 128),       # ragged rows, 128-channel n-tiles
    (2, 12, 24, 32, 128),      # 6 x 24 tiles, one chunk
    (1, 20, 100, 64, 256),     # ragged in x
    (3, 5, 16, 96, 128),       # 9 x 16 tiles, three chunks
    (1, 30, 36, 128, 384),     # three 128-channel n-tiles
]


@pytest.mark.parametrize("N,H,W,Cin,Cout", BF16_CASES)
def test_conv3x3_bf16_kernel(N, H, W, Cin, Cout):
    """Forward with every fused epilogue and the input gradient on the bf16 kernel."""
    from pesr_amd import ops
    x = _rand(N, Cin, H, W, seed=1); w = _rand(Cout, Cin, 3, 3, seed=2, scale=0.1); b = _rand(Cout, seed=3)
    skip = _rand(N, Cout, H, W, seed=4); mk = _rand(N, Cout, H, W, seed=5)
    ref = O.conv3x3_bf16(x, w, b)
    wf = ops.pack_conv3x3_bf16(w.cuda(), 0)
    assert torch.equal(wf.t.cpu().view(9, Cin // 32, Cout, 32)[4, 0, :, :].float(), O.round_bf16(w[:, :32, 1, 1]))   # pack = RNE
    y = ops.conv3x3_fwd(_nhwc(x), wf, b.cuda(), Cout, act=ops.ACT_RELU)
    _close(_nchw(y), torch.relu(ref), 1e-5)
    _close(_nchw(y), torch.relu(O.conv3x3(x, w, b)), 1e-2)              # and it IS the conv, to bf16-operand accuracy
    y = ops.conv3x3_fwd(_nhwc(x), wf, b.cuda(), Cout, alpha=0.1, skip=_nhwc(skip), mask=_nhwc(mk))
    _close(_nchw(y), torch.where(mk > 0, ref * 0.1, torch.zeros_like(ref)) + skip, 1e-5)
    if Cin % 128 == 0:
        dy = _rand(N, Cout, H, W, seed=6)
        dx_ref, _, _ = O.conv3x3_bf16_grads(x, w, dy)
        dx = ops.conv3x3_dgrad(_nhwc(dy), ops.pack_conv3x3_bf16(w.cuda(), 1), (N, H, W, Cin), mask=_nhwc(x), skip=_nhwc(x))
        _close(_nchw(dx), torch.where(x > 0, dx_ref, torch.zeros_like(dx_ref)) + x, 1e-5)


def test_conv3x3_bf16_pixel_shuffle_fused():
    """The upsampler convs on the bf16 kernel: fused PixelShuffle store (forward), pixel-unshuffle load (input gradient)."""
    import torch.nn.functional as F
    from pesr_amd import ops
    N, H, W, C = 1, 12, 24, 256
    x = _rand(N, C, H, W, seed=1); w = _rand(4 * C, C, 3, 3, seed=2, scale=0.1); b = _rand(4 * C, seed=3)
    ref = F.pixel_shuffle(O.conv3x3_bf16(x, w, b), 2)
    y = ops.conv3x3_fwd(_nhwc(x), ops.pack_conv3x3_bf16(w.cuda(), 0, ps=True), ops.pack_bias_ps(b.cuda()), 4 * C, ps_out=True)
    _close(_nchw(y), ref, 1e-5)
    dys = _rand(N, C, 2 * H, 2 * W, seed=4)
    dx_ref, _, _ = O.conv3x3_bf16_grads(x, w, F.pixel_unshuffle(dys, 2))
    dx = ops.conv3x3_dgrad(_nhwc(dys), ops.pack_conv3x3_bf16(w.cuda(), 1, ps=True), (N, H, W, C), ps_in=True)
    _close(_nchw(dx), dx_ref, 1e-5)
