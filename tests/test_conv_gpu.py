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
    (1, 20, 100, 64, 128, 1),  # wider than one 48-pixel wgrad segment: 3 segments per row, the last one ragged
    (1, 30, 70, 64, 64, 2),    # stride 2 with 2 segments per output row (OW = 35)
    (4, 192, 192, 64, 64, 2),  # D features.1 at full resolution: the 64-pixel-tile forward config
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
    _close(_nchw(y), ref, 1e-5)          # flat kernel-level bound, the same the Winograd tests use; measured <= 6e-7


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
        _close(_nchw(dx), dx_ref, 1e-5)
    if Cin % 64 == 0 and Cout % 64 == 0:
        dw, db = ops.conv3x3_wgrad(_nhwc(x), _nhwc(dy), stride)
        _close(dw.cpu(), dw_ref, 1e-5)
        _close(db.cpu(), db_ref, 1e-5)


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
    # the Winograd kernel with the same fusions (Upsampler convs 256 -> 1024 and their input gradients)
    yw = ops.conv3x3_fwd(_nhwc(x), ops.pack_conv3x3_wino(w.cuda(), 0, ps=True), bp, Cout, ps_out=True)
    _close(_nchw(yw), ref, 1e-5)
    w2 = _rand(Cout, 128, 3, 3, seed=8, scale=0.1); x2 = _rand(N, 128, H, W, seed=9)
    dx2_ref, _, _ = O.conv3x3_grads(x2, w2, dy)
    dxw = ops.conv3x3_dgrad(_nhwc(dys), ops.pack_conv3x3_wino(w2.cuda(), 1, ps=True), (N, H, W, 128), ps_in=True)
    _close(_nchw(dxw), dx2_ref, 1e-5)


def test_conv3x3_determinism_race_screen():
    """The LDS-DMA pipeline hands data between waves by counted waits + barriers only; a misplaced wait shows up as rare
    run-to-run differences.  Fixed-order accumulation makes every kernel bitwise reproducible, so: run the G-body shape
    (and a fused-PixelShuffle, a stride-2 and a split-K shape) repeatedly and demand identical bits every time."""
    from pesr_amd import ops
    shapes = [(16, 48, 48, 256, 256, 1, False), (2, 24, 24, 256, 1024, 1, True), (4, 48, 48, 128, 128, 2, False),
              (16, 12, 12, 512, 512, 1, False)]
    for (N, H, W, Cin, Cout, s, ps) in shapes:
        x = _rand(N, Cin, H, W, seed=11).permute(0, 2, 3, 1).contiguous().cuda()
        w = _rand(Cout, Cin, 3, 3, seed=12, scale=0.05).cuda()
        OH, OW = (H - 1) // s + 1, (W - 1) // s + 1
        dy = (_rand(N, Cout // 4 if ps else Cout, 2 * OH if ps else OH, 2 * OW if ps else OW, seed=13)
              .permute(0, 2, 3, 1).contiguous().cuda())
        wp, wpd = ops.pack_conv3x3(w, 0, ps), ops.pack_conv3x3(w, 1, ps)
        y0 = ops.conv3x3_fwd(x, wp, None, Cout, s, act=ops.ACT_RELU, ps_out=ps)
        d0 = ops.conv3x3_dgrad(dy, wpd, (N, H, W, Cin), s, mask=x, ps_in=ps) if s == 1 else ops.conv3x3_dgrad(dy, wpd, (N, H, W, Cin), s)
        g0, b0 = ops.conv3x3_wgrad(x, dy, s, ps_in=ps)
        for _ in range(12):
            assert torch.equal(ops.conv3x3_fwd(x, wp, None, Cout, s, act=ops.ACT_RELU, ps_out=ps), y0)
            d = ops.conv3x3_dgrad(dy, wpd, (N, H, W, Cin), s, mask=x, ps_in=ps) if s == 1 else ops.conv3x3_dgrad(dy, wpd, (N, H, W, Cin), s)
            assert torch.equal(d, d0)
            g, b = ops.conv3x3_wgrad(x, dy, s, ps_in=ps)
            assert torch.equal(g, g0) and torch.equal(b, b0)


def test_conv3x3_random_shape_sweep():
    """Seeded sweep over odd sizes / channel counts / strides (partial tiles, padded channels, every tile config)."""
    from pesr_amd import ops
    import random
    rng = random.Random(7)
    for it in range(14):
        N = rng.choice([1, 2, 3]); H = rng.randint(3, 40); W = rng.randint(3, 40)
        Cin = rng.choice([16, 32, 48, 64, 128]); Cout = rng.choice([3, 16, 64, 96, 128, 256]); s = rng.choice([1, 1, 2])
        if Cout <= 16:
            s = 1          # the <= 16-channel tile exists for stride 1 only (the reference's C -> 3 convs are stride 1)
        x = _rand(N, Cin, H, W, seed=100 + it); w = _rand(Cout, Cin, 3, 3, seed=200 + it, scale=0.1); b = _rand(Cout, seed=300 + it)
        ref = O.conv3x3(x, w, b, s)
        y = ops.conv3x3_fwd(_nhwc(x), ops.pack_conv3x3(w.cuda(), 0), b.cuda(), Cout, s)
        _close(_nchw(y), ref, 1e-5)
        if Cout % 16 == 0:
            dy = _rand(*ref.shape, seed=400 + it)
            dx_ref, dw_ref, db_ref = O.conv3x3_grads(x, w, dy, s)
            dx = ops.conv3x3_dgrad(_nhwc(dy), ops.pack_conv3x3(w.cuda(), 1), (N, H, W, Cin), s)
            _close(_nchw(dx), dx_ref, 1e-5)
            dw, db = ops.conv3x3_wgrad(_nhwc(x), _nhwc(dy), s)
            _close(dw.cpu(), dw_ref, 2e-5); _close(db.cpu(), db_ref, 2e-5)


WINO_CASES = [
    # N, H, W, Cin, Cout     (W even, Cout % 128 == 0)
    (1, 6, 48, 16, 128),
    (2, 7, 10, 32, 128),      # ragged: partial tile rows and a partial x-tile row
    (1, 48, 48, 256, 256),    # K1 shape, one image
    (1, 13, 96, 64, 256),
    (2, 5, 2, 16, 128),       # one pixel pair per row
    (1, 9, 194, 16, 128),     # wider than one tile
    (2, 24, 24, 256, 256),    # few tiles: split-K over the Cin chunks + finish kernel
]


@pytest.mark.parametrize("N,H,W,Cin,Cout", WINO_CASES)
def test_conv3x3_winograd_kernel(N, H, W, Cin, Cout):
    """Winograd F(2,3)-along-x kernel (conv3x3_wino.hip) against the oracle: forward with every fused epilogue, input gradient."""
    from pesr_amd import ops
    x = _rand(N, Cin, H, W, seed=1); w = _rand(Cout, Cin, 3, 3, seed=2, scale=0.1); b = _rand(Cout, seed=3)
    skip = _rand(N, Cout, H, W, seed=4); mk = _rand(N, Cout, H, W, seed=5)
    ref = O.conv3x3(x, w, b)
    wf = ops.pack_conv3x3_wino(w.cuda(), 0)
    y = ops.conv3x3_fwd(_nhwc(x), wf, b.cuda(), Cout, act=ops.ACT_RELU)
    _close(_nchw(y), torch.relu(ref), 1e-5)
    y = ops.conv3x3_fwd(_nhwc(x), wf, b.cuda(), Cout, alpha=0.1, skip=_nhwc(skip), mask=_nhwc(mk))
    _close(_nchw(y), torch.where(mk > 0, ref * 0.1, torch.zeros_like(ref)) + skip, 1e-5)
    if Cin % 128 == 0:
        dy = _rand(N, Cout, H, W, seed=6)
        dx_ref, _, _ = O.conv3x3_grads(x, w, dy)
        dx = ops.conv3x3_dgrad(_nhwc(dy), ops.pack_conv3x3_wino(w.cuda(), 1), (N, H, W, Cin), mask=_nhwc(x), skip=_nhwc(x))
        _close(_nchw(dx), torch.where(x > 0, dx_ref, torch.zeros_like(dx_ref)) + x, 2e-6 * (Cout * 9) ** 0.5)


WINO4_CASES = [
    # N, H, W, Cin, Cout     (W % 4 == 0, Cout % 64 == 0)
    (1, 12, 48, 16, 64),
    (2, 7, 12, 32, 64),       # ragged: partial tile rows, 3 x-tiles per row
    (1, 48, 48, 256, 256),    # K1 shape, one image
    (1, 13, 96, 64, 128),
    (2, 5, 4, 16, 64),        # one x-tile per row
    (1, 9, 196, 16, 64),      # wider than one tile
    (16, 24, 24, 512, 512),   # few tiles: split-K over the Cin chunks + finish kernel
    (1, 30, 20, 48, 192),     # 5 x-tiles per row, Cin not a multiple of 64
    (16, 12, 12, 512, 512),   # VGG conv5_x: 16 images stacked into one 207-row image (zero separator rows), 3-x-tile rows in
                              # the dense LDS layout, split-K
    (3, 12, 12, 128, 64),     # stacked, no split-K, odd image count
    (5, 6, 8, 64, 64),        # stacked, 2 x-tiles per row
    (4, 9, 20, 64, 128),      # stacked candidate with 5-x-tile rows (tile rows of 4 / 6 x-tiles: ragged in x as well)
    (2, 6, 96, 64, 64),       # 24-x-tile rows, 6 rows per tile: the kernel's compile-time-row-length form for 96-wide layers
    (1, 12, 192, 64, 64),     # 48-x-tile rows (dense layout), 3 rows per tile
]


@pytest.mark.parametrize("N,H,W,Cin,Cout", WINO4_CASES)
def test_conv3x3_winograd4_kernel(N, H, W, Cin, Cout):
    """Winograd F(4,3)-along-x kernel (conv3x3_wino4.hip) against the oracle: forward with every fused epilogue, input
    gradient.  Its measured error vs fp64 is <= 2.3e-6 of the output maximum (scripts/wino4_test.py); the bound here is 1e-5."""
    from pesr_amd import ops
    x = _rand(N, Cin, H, W, seed=1); w = _rand(Cout, Cin, 3, 3, seed=2, scale=0.1); b = _rand(Cout, seed=3)
    skip = _rand(N, Cout, H, W, seed=4); mk = _rand(N, Cout, H, W, seed=5)
    ref = O.conv3x3(x, w, b)
    wf = ops.pack_conv3x3_wino4(w.cuda(), 0)
    y = ops.conv3x3_fwd(_nhwc(x), wf, b.cuda(), Cout, act=ops.ACT_RELU)
    _close(_nchw(y), torch.relu(ref), 1e-5)
    y = ops.conv3x3_fwd(_nhwc(x), wf, b.cuda(), Cout, alpha=0.1, skip=_nhwc(skip), mask=_nhwc(mk))
    _close(_nchw(y), torch.where(mk > 0, ref * 0.1, torch.zeros_like(ref)) + skip, 1e-5)
    if Cin % 64 == 0:
        dy = _rand(N, Cout, H, W, seed=6)
        dx_ref, _, _ = O.conv3x3_grads(x, w, dy)
        dx = ops.conv3x3_dgrad(_nhwc(dy), ops.pack_conv3x3_wino4(w.cuda(), 1), (N, H, W, Cin), mask=_nhwc(x), skip=_nhwc(x))
        _close(_nchw(dx), torch.where(x > 0, dx_ref, torch.zeros_like(dx_ref)) + x, 1e-5)


def test_conv3x3_winograd4_pixel_shuffle_fused():
    """The upsampler convs on the F(4,3) kernel: fused PixelShuffle store (forward) and pixel-unshuffle load (input gradient)."""
    import torch.nn.functional as F
    from pesr_amd import ops
    N, H, W, C = 2, 12, 24, 64
    x = _rand(N, C, H, W, seed=1); w = _rand(4 * C, C, 3, 3, seed=2, scale=0.1); b = _rand(4 * C, seed=3)
    ref = F.pixel_shuffle(O.conv3x3(x, w, b), 2)
    y = ops.conv3x3_fwd(_nhwc(x), ops.pack_conv3x3_wino4(w.cuda(), 0, ps=True), ops.pack_bias_ps(b.cuda()), 4 * C, ps_out=True)
    _close(_nchw(y), ref, 1e-5)
    dys = _rand(N, C, 2 * H, 2 * W, seed=4)
    dx_ref, _, _ = O.conv3x3_grads(x, w, F.pixel_unshuffle(dys, 2))
    dx = ops.conv3x3_dgrad(_nhwc(dys), ops.pack_conv3x3_wino4(w.cuda(), 1, ps=True), (N, H, W, C), ps_in=True)
    _close(_nchw(dx), dx_ref, 1e-5)


def test_winograd4_random_shape_sweep():
    from pesr_amd import ops
    import random
    rng = random.Random(13)
    for it in range(10):
        N = rng.choice([1, 2, 3]); H = rng.randint(1, 30); W = 4 * rng.randint(1, 26)
        Cin = rng.choice([16, 64, 128]); Cout = rng.choice([64, 128, 192])
        x = _rand(N, Cin, H, W, seed=500 + it); w = _rand(Cout, Cin, 3, 3, seed=600 + it, scale=0.1); b = _rand(Cout, seed=700 + it)
        ref = O.conv3x3(x, w, b)
        y = ops.conv3x3_fwd(_nhwc(x), ops.pack_conv3x3_wino4(w.cuda(), 0), b.cuda(), Cout)
        _close(_nchw(y), ref, 1e-5)
        if Cin % 64 == 0:
            dy = _rand(N, Cout, H, W, seed=800 + it)
            dx_ref, _, _ = O.conv3x3_grads(x, w, dy)
            dx = ops.conv3x3_dgrad(_nhwc(dy), ops.pack_conv3x3_wino4(w.cuda(), 1), (N, H, W, Cin))
            _close(_nchw(dx), dx_ref, 1e-5)


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 9, 48, 64, 64), (1, 5, 96, 64, 128), (3, 4, 96, 128, 64), (1, 50, 48, 64, 192),
                                             (2, 3, 144, 64, 64), (1, 7, 92, 64, 64), (2, 48, 48, 256, 256), (1, 5, 48, 64, 64)])
def test_conv3x3_wgrad_all_algorithms(N, H, W, Cin, Cout):
    """Weight + bias gradient through the three algorithms of pesr_conv3x3_wgrad (auto = transposed Winograd F(4,3) where it
    applies, F(2,3), direct) against the oracle; the same bound for all (measured vs fp64: <= 2e-6 of the gradient maximum)."""
    from pesr_amd import ops
    x = _rand(N, Cin, H, W, seed=900); w = _rand(Cout, Cin, 3, 3, seed=910, scale=0.1)
    dy = _rand(N, Cout, H, W, seed=920)
    _, dw_ref, db_ref = O.conv3x3_grads(x.double(), w.double(), dy.double())
    got = {}
    for algo in (ops.WGRAD_AUTO, ops.WGRAD_WINO23, ops.WGRAD_DIRECT, ops.WGRAD_WINO4_16X16, ops.WGRAD_WINO4_1D, ops.WGRAD_WINO4_12W):
        dw, db = ops.conv3x3_wgrad(_nhwc(x), _nhwc(dy), 1, alpha=0.5, algo=algo)
        _close(dw.cpu().double(), 0.5 * dw_ref, 1e-5); _close(db.cpu().double(), 0.5 * db_ref, 1e-5)
        got[algo] = (dw, db)
    # the 16-wave kernel (producer waves) and round 4's 12-wave kernel: same transform, same order of additions
    assert torch.equal(got[ops.WGRAD_AUTO][0], got[ops.WGRAD_WINO4_12W][0]) and torch.equal(got[ops.WGRAD_AUTO][1], got[ops.WGRAD_WINO4_12W][1])


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(16, 24, 24, 256, 512), (4, 9, 24, 64, 128), (9, 6, 24, 64, 64), (6, 7, 16, 64, 64),
                                             (4, 12, 12, 128, 64), (12, 5, 8, 64, 64)])
def test_conv3x3_wgrad_narrow_images_side_by_side(N, H, W, Cin, Cout):
    """Rows shorter than the transposed-Winograd kernel's 48-pixel strip: 12 / (W / 4) images are laid side by side in one strip
    (the Discriminator's 24-pixel-wide features.6; an odd image count leaves the last strip half empty).  A pixel's halo must stop
    at its own image's border - the neighbour in the strip is another image -, odd heights end in a half-empty row pair.  Both
    transforms (y-nested, 1-D) against the oracle, with and without accumulation; the result must differ in its bits from the
    direct kernel's (i.e. the shape really ran on the Winograd kernel), and repeat bit-identically."""
    from pesr_amd import ops
    assert ops.wgrad_kernel_for(N, H, W, Cin, Cout)[0] == "conv3x3_wgrad_wino4p_kernel"
    x = _rand(N, Cin, H, W, seed=930); w = _rand(Cout, Cin, 3, 3, seed=931, scale=0.1)
    dy = _rand(N, Cout, H, W, seed=932)
    _, dw_ref, db_ref = O.conv3x3_grads(x.double(), w.double(), dy.double())
    dw_d, _ = ops.conv3x3_wgrad(_nhwc(x), _nhwc(dy), 1, algo=ops.WGRAD_DIRECT)
    for algo in (ops.WGRAD_AUTO, ops.WGRAD_WINO4_1D, ops.WGRAD_WINO4_12W):
        dw, db = ops.conv3x3_wgrad(_nhwc(x), _nhwc(dy), 1, algo=algo)
        _close(dw.cpu().double(), dw_ref, 1e-5); _close(db.cpu().double(), db_ref, 1e-5)
        assert not torch.equal(dw, dw_d)
        dw2, db2 = ops.conv3x3_wgrad(_nhwc(x), _nhwc(dy), 1, algo=algo)
        assert torch.equal(dw, dw2) and torch.equal(db, db2)
        acc_w, acc_b = dw.clone(), db.clone()
        ops.conv3x3_wgrad(_nhwc(x), _nhwc(dy), 1, algo=algo, dw_out=acc_w, db_out=acc_b, accumulate=True)
        _close(acc_w.cpu().double(), 2 * dw_ref, 1e-5); _close(acc_b.cpu().double(), 2 * db_ref, 1e-5)


def test_conv3x3_wgrad_producer_kernel_random_shape_sweep():
    """The 16-wave weight-gradient kernel (four producer waves) over random shapes the planner accepts - odd heights, strips that
    end half empty, several images side by side in a strip with an odd image count, one segment per split-K slice, with and without
    the bias gradient, accumulating -: bit-identical to round 4's 12-wave kernel (same transform, same order of additions) and
    within 1e-5 of the oracle."""
    from pesr_amd import ops
    import random
    rng = random.Random(31)
    done = 0
    for it in range(200):
        if done == 24:
            break
        N = rng.randint(1, 13); H = rng.randint(1, 27)
        W = rng.choice([8, 12, 16, 24, 48, 48, 96, 92, 144]); Cin = rng.choice([64, 128, 256]); Cout = rng.choice([64, 128, 192])
        if ops.wgrad_kernel_for(N, H, W, Cin, Cout)[0] != "conv3x3_wgrad_wino4p_kernel":
            continue
        done += 1
        x = _rand(N, Cin, H, W, seed=3000 + it); w = _rand(Cout, Cin, 3, 3, seed=3100 + it, scale=0.1)
        dy = _rand(N, Cout, H, W, seed=3200 + it)
        _, dw_ref, db_ref = O.conv3x3_grads(x.double(), w.double(), dy.double())
        want_bias = bool(it & 1)
        dw, db = ops.conv3x3_wgrad(_nhwc(x), _nhwc(dy), 1, algo=ops.WGRAD_AUTO, want_bias=want_bias)
        dw2, db2 = ops.conv3x3_wgrad(_nhwc(x), _nhwc(dy), 1, algo=ops.WGRAD_WINO4_12W, want_bias=want_bias)
        assert torch.equal(dw, dw2), (N, H, W, Cin, Cout)
        _close(dw.cpu().double(), dw_ref, 1e-5)
        if want_bias:
            assert torch.equal(db, db2), (N, H, W, Cin, Cout)
            _close(db.cpu().double(), db_ref, 1e-5)
            ops.conv3x3_wgrad(_nhwc(x), _nhwc(dy), 1, algo=ops.WGRAD_AUTO, dw_out=dw, db_out=db, accumulate=True)
            _close(dw.cpu().double(), 2 * dw_ref, 1e-5); _close(db.cpu().double(), 2 * db_ref, 1e-5)
    assert done == 24


def test_conv3x3_wgrad_winograd4_pixel_shuffle_fused():
    """Weight gradient of an upsampler conv (its output gradient arrives pixel-shuffled) on the F(4,3) kernel."""
    import torch.nn.functional as F
    from pesr_amd import ops
    N, H, W, C = 2, 6, 48, 64
    x = _rand(N, C, H, W, seed=1); w = _rand(4 * C, C, 3, 3, seed=2, scale=0.1)
    dys = _rand(N, C, 2 * H, 2 * W, seed=4)
    _, dw_ref, db_ref = O.conv3x3_grads(x.double(), w.double(), F.pixel_unshuffle(dys, 2).double())
    for algo in (ops.WGRAD_AUTO, ops.WGRAD_WINO23, ops.WGRAD_WINO4_16X16, ops.WGRAD_WINO4_1D, ops.WGRAD_WINO4_12W):
        dw, db = ops.conv3x3_wgrad(_nhwc(x), _nhwc(dys), 1, ps_in=True, algo=algo)
        _close(dw.cpu().double(), dw_ref, 1e-5); _close(db.cpu().double(), db_ref, 1e-5)


def test_winograd_dispatch_rule():
    """functional picks the Winograd packing only where the kernel applies and fills the chip (the G body shape), and the
    direct one elsewhere (odd widths, fused PixelShuffle, stride 2, small layers)."""
    from pesr_amd import ops
    assert ops.wino_eligible(16, 48, 48, 256, 256)
    assert not ops.wino_eligible(16, 48, 47, 256, 256)          # odd width
    assert not ops.wino_eligible(16, 48, 48, 256, 256, stride=2)
    assert ops.wino_eligible(16, 24, 24, 512, 512)              # 128 tiles: split-K over the Cin chunks fills the chip
    assert not ops.wino_eligible(16, 12, 12, 512, 512)          # half-empty tiles: the direct kernel (smaller tiles) instead
    assert not ops.wino_eligible(16, 48, 48, 256, 64)
    # F(4,3) (preferred where it applies): W % 4 == 0, Cout % 64 == 0, >= 192 workgroups, tiles mostly inside the image
    assert ops.wino4_eligible(16, 48, 48, 256, 256) and ops.wino4_eligible(16, 48, 48, 256, 64)
    assert ops.wino4_eligible(16, 24, 24, 512, 512)             # split-K fills the chip
    assert ops.wino4_eligible(16, 96, 96, 256, 1024, ps_out=True)
    assert not ops.wino4_eligible(16, 48, 50, 256, 256)         # width not a multiple of 4
    assert ops.wino4_eligible(16, 12, 12, 512, 512)             # 36 x-tiles per image: the 16 images are stacked into one (80 % cover)
    assert not ops.wino4_eligible(2, 12, 12, 512, 512)          # too few workgroups even stacked
    assert not ops.wino4_eligible(1, 48, 48, 256, 256)          # 16 workgroups: the direct kernel's small tiles instead
    assert not ops.wino4_eligible(16, 48, 48, 256, 256, stride=2)


def test_winograd_random_shape_sweep():
    """Seeded sweep of the Winograd kernels over odd heights, ragged widths (partial tiles, partial last strip), batch sizes
    and channel counts: forward, input gradient and - through the ordinary wgrad entry point, which picks the Winograd form
    where it applies - weight + bias gradient."""
    from pesr_amd import ops
    import random
    rng = random.Random(11)
    for it in range(10):
        N = rng.choice([1, 2, 3]); H = rng.randint(1, 30); W = 2 * rng.randint(1, 50)
        Cin = rng.choice([16, 64, 128]); Cout = rng.choice([128, 256])
        x = _rand(N, Cin, H, W, seed=500 + it); w = _rand(Cout, Cin, 3, 3, seed=600 + it, scale=0.1); b = _rand(Cout, seed=700 + it)
        ref = O.conv3x3(x, w, b)
        y = ops.conv3x3_fwd(_nhwc(x), ops.pack_conv3x3_wino(w.cuda(), 0), b.cuda(), Cout)
        _close(_nchw(y), ref, 1e-5)
        if Cin % 128 == 0:
            dy = _rand(N, Cout, H, W, seed=800 + it)
            dx_ref, _, _ = O.conv3x3_grads(x, w, dy)
            dx = ops.conv3x3_dgrad(_nhwc(dy), ops.pack_conv3x3_wino(w.cuda(), 1), (N, H, W, Cin))
            _close(_nchw(dx), dx_ref, 1e-5)
    for it, (N, H, W, Cin, Cout) in enumerate([(2, 9, 48, 64, 64), (1, 5, 94, 64, 128), (3, 4, 96, 128, 64), (1, 50, 48, 64, 192),
                                               (2, 3, 144, 64, 64)]):
        x = _rand(N, Cin, H, W, seed=900 + it); w = _rand(Cout, Cin, 3, 3, seed=910 + it, scale=0.1)
        dy = _rand(N, Cout, H, W, seed=920 + it)
        _, dw_ref, db_ref = O.conv3x3_grads(x, w, dy)
        dw, db = ops.conv3x3_wgrad(_nhwc(x), _nhwc(dy), 1)
        _close(dw.cpu(), dw_ref, 2e-5); _close(db.cpu(), db_ref, 2e-5)


@pytest.mark.parametrize("N,H,W,Cin,bias,act", [
    (2, 7, 192, 64, True, 0),       # one full-width strip, a band shorter than 12 rows
    (1, 13, 50, 64, False, 0),      # narrower than a strip: columns >= W read zeros; ragged band (13 = 12 + 1)
    (1, 9, 400, 64, True, 0),       # three overlapping 190-column strips, the last one ragged
    (2, 192, 192, 256, True, 0),    # the Generator's last conv at two images of the training shape
    (1, 30, 16, 128, True, 2),      # LeakyReLU epilogue, three bands
])
def test_conv3x3_rgb_out_kernel(N, H, W, Cin, bias, act):
    """C -> 3 forward (dedicated kernel, x stencil folded into the MFMA N dimension) vs the oracle conv and vs the
    implicit-GEMM kernel it replaces (reference model/pesr.py:38 via model/basic.py:4-7)."""
    from pesr_amd import ops
    x = _rand(N, Cin, H, W, seed=1)
    w = _rand(3, Cin, 3, 3, seed=2, scale=0.1)
    b = _rand(3, seed=3) if bias else None
    ref = O.conv3x3(x, w, b, 1)
    if act == 2:
        ref = torch.nn.functional.leaky_relu(ref, 0.2)
    assert ops.rgb_out_eligible(Cin, 3, 1)
    xg, wg, bg = _nhwc(x), w.cuda(), (b.cuda() if bias else None)
    kw = dict(act=ops.ACT_LRELU, slope=0.2) if act == 2 else {}
    y = ops.conv3x3_fwd(xg, lambda: (_ for _ in ()).throw(AssertionError("the packed weights must not be needed")), bg, 3, w_oihw=wg, **kw)
    _close(_nchw(y), ref, 1e-5)          # flat kernel-level bound, the same the Winograd tests use; measured <= 6e-7
    ops.USE_RGB_OUT = False
    try:
        y2 = ops.conv3x3_fwd(xg, ops.pack_conv3x3(wg, 0), bg, 3, w_oihw=wg, **kw)
    finally:
        ops.USE_RGB_OUT = True
    _close(_nchw(y), _nchw(y2), 2e-6 * (Cin * 9) ** 0.5)


@pytest.mark.parametrize("k,stride,bias", [(1, 1, True), (5, 1, True), (5, 2, False), (7, 1, False), (1, 2, True)])
def test_conv_other_kernel_sizes_vs_torch(k, stride, bias):
    """reference `Conv(in, out, kernel_size, stride, bias)` (model/basic.py:4-7) takes any kernel size; its networks use 3 only.
    The generic kernels (conv_kxk.hip) behind pesr_amd's Conv for odd k != 3: forward, input gradient, weight and bias gradient
    against torch's conv2d on the CPU, same seeded construction."""
    from pesr_amd.model.basic import Conv
    torch.manual_seed(k * 10 + stride)
    ours = Conv(6, 10, k, stride=stride, bias=bias)
    torch.manual_seed(k * 10 + stride)
    ref = torch.nn.Conv2d(6, 10, k, padding=k // 2, stride=stride, bias=bias)
    assert torch.equal(ours.weight.detach(), ref.weight.detach())
    ours.cuda()
    x = (torch.rand(2, 6, 9, 11, generator=torch.Generator().manual_seed(k)) * 2 - 1)
    xr = x.clone().requires_grad_(True)
    xg = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    yr, yg = ref(xr), ours(xg)
    assert yg.shape == yr.shape
    g = torch.rand(yr.shape, generator=torch.Generator().manual_seed(k + 1)) * 2 - 1
    yr.backward(g); yg.backward(g.cuda())
    scale = lambda t: float(t.abs().max()) + 1e-30
    assert float((yg.detach().cpu() - yr.detach()).abs().max()) <= 2e-6 * scale(yr)
    assert float((xg.grad.cpu() - xr.grad).abs().max()) <= 2e-6 * scale(xr.grad)
    assert float((ours.weight.grad.cpu() - ref.weight.grad).abs().max()) <= 5e-6 * scale(ref.weight.grad)
    if bias:
        assert float((ours.bias.grad.cpu() - ref.bias.grad).abs().max()) <= 5e-6 * scale(ref.bias.grad)


def test_basic_block_with_5x5_conv_vs_torch():
    """BasicBlock(kernel_size=5, bn=True, LeakyReLU) - a constructor combination of reference model/basic.py:19-31 - on the generic
    conv followed by the BatchNorm / activation kernels, against the same block built from torch modules."""
    import torch.nn as nn
    from pesr_amd.model.basic import BasicBlock
    torch.manual_seed(4)
    ours = BasicBlock(4, 8, 5, stride=1, bias=True, bn=True, act=nn.LeakyReLU(0.2, True), sn=False)
    torch.manual_seed(4)
    ref = nn.Sequential(nn.Conv2d(4, 8, 5, padding=2, bias=True), nn.BatchNorm2d(8), nn.LeakyReLU(0.2, True))
    ours.cuda()
    x = torch.rand(3, 4, 8, 8, generator=torch.Generator().manual_seed(1)) * 2 - 1
    xr = x.clone().requires_grad_(True)
    xg = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    yr, yg = ref(xr), ours(xg)
    g = torch.rand(yr.shape, generator=torch.Generator().manual_seed(2)) * 2 - 1
    yr.backward(g); yg.backward(g.cuda())
    sc = lambda t: float(t.abs().max()) + 1e-30
    assert float((yg.detach().cpu() - yr.detach()).abs().max()) <= 5e-6 * sc(yr)
    assert float((xg.grad.cpu() - xr.grad).abs().max()) <= 2e-5 * sc(xr.grad)
    assert float((ours[0].weight.grad.cpu() - ref[0].weight.grad).abs().max()) <= 2e-5 * sc(ref[0].weight.grad)
    assert float((ours[1].weight.grad.cpu() - ref[1].weight.grad).abs().max()) <= 2e-5 * sc(ref[1].weight.grad)
    assert torch.allclose(ours[1].running_mean.cpu(), ref[1].running_mean, rtol=1e-5, atol=1e-7)


def test_conv3x3_s2_dgrad_256_channel_tiles_one_tap_class():
    """Stride-2 input gradient on the 144-pixel x 256-channel tiles (>= 192 of them: dx [3,192,192,256]).  Its one-tap parity class
    reads a weight slab in the same iteration that stages the slab two ahead; with two weight buffers the early waves overwrote
    rows the late waves had not read (rounds 1 - 3: channels 128 - 191 of the (even, even) pixels wrong by O(1); found in round 4,
    when the four classes became one launch) - now three buffers.  Every parity class against the oracle."""
    from pesr_amd import ops
    N, H, W, Cin, Cout = 3, 192, 192, 256, 256
    x = _rand(N, Cin, H, W, seed=1)
    w = _rand(Cout, Cin, 3, 3, seed=2, scale=0.1)
    dy = _rand(N, Cout, 96, 96, seed=5)
    torch.set_num_threads(max(1, min(16, len(__import__("os").sched_getaffinity(0)))))
    dx_ref, _, _ = O.conv3x3_grads(x, w, dy, 2)
    dx = _nchw(ops.conv3x3_dgrad(_nhwc(dy), ops.pack_conv3x3(w.cuda(), 1), (N, H, W, Cin), 2))
    for py in (0, 1):
        for px in (0, 1):
            _close(dx[:, :, py::2, px::2], dx_ref[:, :, py::2, px::2], 1e-5)
    for rep in range(3):        # and bit-reproducible (a race would not be)
        assert torch.equal(dx, _nchw(ops.conv3x3_dgrad(_nhwc(dy), ops.pack_conv3x3(w.cuda(), 1), (N, H, W, Cin), 2)))


@pytest.mark.parametrize("O,I,ps", [(64, 64, False), (256, 256, False), (128, 64, False), (64, 192, False), (1024, 256, True), (256, 64, True)])
def test_batched_wino4_repack_equals_the_single_pack(O, I, ps):
    """The one-launch re-pack behind every optimizer step (pesr_pack_conv3x3_batched, F(4,3) modes 4 / 5: one thread per
    (row, channel) of a 16 x 16 tile) against the per-element packing the first forward uses (pesr_pack_conv3x3_wino4): bit-identical, forward and input-gradient
    layouts, with and without the pixel-shuffle channel order; also from a weight tensor that sits 4 bytes off a 16-byte boundary."""
    import numpy as np
    from pesr_amd import ops, _lib
    for shift in (0, 1):
        base = torch.empty(O * I * 9 + 4, device="cuda")
        w = base[shift:shift + O * I * 9].view(O, I, 3, 3)
        w.copy_(_rand(O, I, 3, 3, seed=O + I + shift, scale=0.1))
        jobs, outs = [], []
        for mode in (0, 1):
            out = torch.full((18 * O * I,), float("nan"), device="cuda")
            outs.append(out)
            jobs.append((w.data_ptr(), out.data_ptr(), O, I, 4 + mode, int(ps), I if mode == 0 else O, O if mode == 0 else I))
        table = torch.from_numpy(np.array(jobs, dtype=np.int64)).cuda()
        _lib.check(_lib.lib().pesr_pack_conv3x3_batched(table.data_ptr(), len(jobs), torch.cuda.current_stream().cuda_stream), "batched pack")
        for mode in (0, 1):
            ref = ops.pack_conv3x3_wino4(w, mode, ps=ps).t
            assert torch.equal(outs[mode], ref), (mode, shift)
