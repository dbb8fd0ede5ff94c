"""GPU parity: MFMA 3x3 conv fwd / dgrad / wgrad (through the C ABI) vs the CPU oracle."""
import pytest
import torch

from oracle import ops as O

pytestmark = pytest.mark.gpu


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def _nhwc(x):  # logical NCHW cpu -> NHWC contiguous cuda
    return x.permute(0, 2, 3, 1).contiguous().cuda()


def _nchw(y):  # NHWC cuda -> logical NCHW cpu
    return y.permute(0, 3, 1, 2).cpu()


def _close(a, b, rel):
    scale = b.abs().max().item() + 1e-30
    err = (a - b).abs().max().item()
    assert err <= rel * scale, f"max err {err:.3e} vs scale {scale:.3e} (rel {err / scale:.3e} > {rel})"


CASES = [
    # N, H, W, Cin, Cout, stride
    (2, 12, 12, 16, 64, 1),
    (1, 7, 9, 32, 64, 1),      # ragged: partial tiles in both directions
    (2, 48, 48, 64, 128, 1),
    (1, 48, 48, 256, 256, 1),  # K1 shape, one image
    (2, 24, 24, 64, 64, 2),
    (1, 13, 11, 32, 128, 2),   # odd sizes with stride 2
    (1, 24, 24, 128, 256, 2),
]


@pytest.mark.parametrize("N,H,W,Cin,Cout,stride", CASES)
def test_conv3x3_fwd(N, H, W, Cin, Cout, stride):
    from pesr_amd import ops
    x = _rand(N, Cin, H, W, seed=1)
    w = _rand(Cout, Cin, 3, 3, seed=2, scale=0.1)
    b = _rand(Cout, seed=3)
    ref = O.conv3x3(x, w, b, stride)
    wp = ops.pack_conv3x3(w.cuda(), 0)
    y = ops.conv3x3_fwd(_nhwc(x), wp, b.cuda(), Cout, stride)
    _close(_nchw(y), ref, 2e-6 * (Cin * 9) ** 0.5)


def test_conv3x3_fwd_epilogue():
    from pesr_amd import ops
    N, H, W, C = 2, 12, 12, 64
    x = _rand(N, C, H, W, seed=1)
    w = _rand(C, C, 3, 3, seed=2, scale=0.1)
    b = _rand(C, seed=3)
    skip = _rand(N, C, H, W, seed=4)
    wp = ops.pack_conv3x3(w.cuda(), 0)
    # relu epilogue
    y = ops.conv3x3_fwd(_nhwc(x), wp, b.cuda(), C, act=ops.ACT_RELU)
    _close(_nchw(y), torch.relu(O.conv3x3(x, w, b)), 1e-5)
    # scale + skip (ResBlock tail, reference model/basic.py:49-50)
    y = ops.conv3x3_fwd(_nhwc(x), wp, b.cuda(), C, alpha=0.1, skip=_nhwc(skip))
    _close(_nchw(y), O.conv3x3(x, w, b) * 0.1 + skip, 1e-5)


@pytest.mark.parametrize("N,H,W,Cin,Cout,stride", CASES)
def test_conv3x3_dgrad_wgrad(N, H, W, Cin, Cout, stride):
    from pesr_amd import ops
    x = _rand(N, Cin, H, W, seed=1)
    w = _rand(Cout, Cin, 3, 3, seed=2, scale=0.1)
    OH, OW = (H - 1) // stride + 1, (W - 1) // stride + 1
    dy = _rand(N, Cout, OH, OW, seed=5)
    dx_ref, dw_ref, db_ref = O.conv3x3_grads(x, w, dy, stride)
    if Cout % 16 == 0 and Cin % 64 == 0:
        wpd = ops.pack_conv3x3(w.cuda(), 1)
        dx = ops.conv3x3_dgrad(_nhwc(dy), wpd, (N, H, W, Cin), stride)
        _close(_nchw(dx), dx_ref, 2e-6 * (Cout * 9) ** 0.5)
    if Cin % 64 == 0 and Cout % 64 == 0:
        dw, db = ops.conv3x3_wgrad(_nhwc(x), _nhwc(dy), stride)
        _close(dw.cpu(), dw_ref, 2e-6 * (N * OH * OW) ** 0.5)
        _close(db.cpu(), db_ref, 2e-6 * (N * OH * OW) ** 0.5)


def test_conv3x3_pixel_shuffle_fused():
    """conv -> PixelShuffle(2) fused in the epilogue; dgrad/wgrad read the shuffled gradient directly."""
    from pesr_amd import ops
    N, H, W, Cin, C = 1, 12, 12, 64, 128
    Cout = 4 * C
    x = _rand(N, Cin, H, W, seed=1)
    w = _rand(Cout, Cin, 3, 3, seed=2, scale=0.1)
    b = _rand(Cout, seed=3)
    ref = O.pixel_shuffle(O.conv3x3(x, w, b))
    wp = ops.pack_conv3x3(w.cuda(), 0, ps=True)
    bp = ops.pack_bias_ps(b.cuda())
    y = ops.conv3x3_fwd(_nhwc(x), wp, bp, Cout, ps_out=True)
    assert y.shape == (N, 2 * H, 2 * W, C)
    _close(_nchw(y), ref, 1e-5)
    # backward
    dys = _rand(N, C, 2 * H, 2 * W, seed=7)
    dy = O.pixel_unshuffle(dys)
    dx_ref, dw_ref, db_ref = O.conv3x3_grads(x, w, dy)
    wpd = ops.pack_conv3x3(w.cuda(), 1, ps=True)
    dx = ops.conv3x3_dgrad(_nhwc(dys), wpd, (N, H, W, Cin), ps_in=True)
    _close(_nchw(dx), dx_ref, 1e-5)
    dw, db = ops.conv3x3_wgrad(_nhwc(x), _nhwc(dys), ps_in=True)
    _close(dw.cpu(), dw_ref, 1e-5)
    _close(db.cpu(), db_ref, 1e-5)
