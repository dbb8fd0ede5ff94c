"""GPU parity of the non-MFMA kernels (through the C ABI) vs torch-CPU restatements (oracle/ops.py)."""
import pytest
import torch
import torch.nn.functional as F

from oracle import detrand
from oracle import ops as O

pytestmark = pytest.mark.gpu


def _rand(*shape, seed=0, lo=-1.0, hi=1.0):
    return detrand.uniform(shape, seed, lo, hi)


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous().cuda()


def _nchw(y):
    return y.permute(0, 3, 1, 2).cpu()


def _close(a, b, rel, what=""):
    scale = b.abs().max().item() + 1e-30
    err = (a - b).abs().max().item()
    assert err <= rel * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e} (rel {err / scale:.3e} > {rel})"


@pytest.mark.parametrize("N,H,W,C", [(2, 12, 12, 64), (1, 9, 7, 256), (2, 37, 70, 64), (2, 37, 70, 256), (1, 5, 192, 512)])
def test_rgb_convs(N, H, W, C):
    """3 -> C and C -> 3 convs (zero-padded onto the MFMA kernel) and their grads."""
    from pesr_amd import ops
    x3 = detrand.image_batch((N, 3, H, W), 3)
    w = _rand(C, 3, 3, 3, seed=2, lo=-0.2, hi=0.2)
    b = _rand(C, seed=4)
    # forward 3 -> C
    wp = ops.pack_conv3x3(w.cuda(), 0)
    y = ops.conv3x3_fwd(_nhwc(x3), wp, b.cuda(), C, act=ops.ACT_RELU)
    _close(_nchw(y), torch.relu(O.conv3x3(x3, w, b)), 1e-5, "fwd 3->C")
    y_direct = ops.conv3x3_fwd(_nhwc(x3), None, b.cuda(), C, act=ops.ACT_RELU, w_oihw=w.cuda())   # direct RGB-input kernel
    _close(_nchw(y_direct), torch.relu(O.conv3x3(x3, w, b)), 1e-5, "fwd 3->C direct")
    y_nb = ops.conv3x3_fwd(_nhwc(x3), None, None, C, w_oihw=w.cuda())
    _close(_nchw(y_nb), O.conv3x3(x3, w, None), 1e-5, "fwd 3->C direct, no bias")
    dy = _rand(N, C, H, W, seed=5)
    dx_ref, dw_ref, db_ref = O.conv3x3_grads(x3, w, dy)
    wpd = ops.pack_conv3x3(w.cuda(), 1)
    dx = ops.conv3x3_dgrad(_nhwc(dy), wpd, (N, H, W, 3))
    _close(_nchw(dx), dx_ref, 1e-5, "dgrad C->3")
    assert ops.rgb_in_dgrad_eligible(3, C, 1)
    dxd = ops.conv3x3_rgb_in_dgrad(_nhwc(dy), w.cuda(), (N, H, W, 3))     # HBM-bound kernel (the C -> 3 forward kernel, transposing weight index)
    _close(_nchw(dxd), dx_ref, 1e-5, "dgrad C->3, streaming kernel")
    dw, db = ops.conv3x3_wgrad_rgb(_nhwc(dy), _nhwc(x3), 0)
    _close(dw.cpu(), dw_ref, 1e-5, "wgrad 3->C")
    _close(db.cpu(), db_ref, 1e-5, "bgrad 3->C")
    # forward C -> 3
    xc = _rand(N, C, H, W, seed=6)
    w2 = _rand(3, C, 3, 3, seed=7, lo=-0.1, hi=0.1)
    b2 = _rand(3, seed=8)
    wp2 = ops.pack_conv3x3(w2.cuda(), 0)
    y2 = ops.conv3x3_fwd(_nhwc(xc), wp2, b2.cuda(), 3)
    assert y2.shape == (N, H, W, 3)
    _close(_nchw(y2), O.conv3x3(xc, w2, b2), 1e-5, "fwd C->3")
    dy3 = _rand(N, 3, H, W, seed=9)
    dx_ref, dw_ref, db_ref = O.conv3x3_grads(xc, w2, dy3)
    dx2 = ops.conv3x3_dgrad(_nhwc(dy3), ops.pack_conv3x3(w2.cuda(), 1), (N, H, W, C))
    _close(_nchw(dx2), dx_ref, 1e-5, "dgrad 3->C")
    dx2d = ops.conv3x3_rgb_dgrad(_nhwc(dy3), w2.cuda(), (N, H, W, C))                             # direct kernel (3 -> C conv of dy)
    _close(_nchw(dx2d), dx_ref, 1e-5, "dgrad 3->C direct")
    dw2, db2 = ops.conv3x3_wgrad_rgb(_nhwc(xc), _nhwc(dy3), 1)
    _close(dw2.cpu(), dw_ref, 1e-5, "wgrad C->3")
    _close(db2.cpu(), db_ref, 1e-5, "bgrad C->3")


def test_rgb_in_dgrad_full_size_and_through_autograd():
    """Input gradient of Discriminator features.0 / vgg19 features.0 at the benchmarked size (16 x 192 x 192, 3 -> 64) on the
    streaming kernel vs the implicit-GEMM path it replaces (reference model/pesr.py:53, model/vgg.py:8-10), and through the
    autograd Functions that dispatch to it."""
    from pesr_amd import functional as PF
    from pesr_amd import ops
    N, H, W, C = 16, 192, 192, 64
    w = _rand(C, 3, 3, 3, seed=2, lo=-0.2, hi=0.2).cuda()
    dy = _rand(N, H, W, C, seed=5).cuda()
    a = ops.conv3x3_rgb_in_dgrad(dy, w, (N, H, W, 3))
    b = ops.conv3x3_dgrad(dy, ops.pack_conv3x3(w, 1), (N, H, W, 3))
    _close(a.cpu(), b.cpu(), 1e-5, "streaming vs implicit GEMM")
    x = detrand.image_batch((2, 3, 20, 28), 3)
    ws = _rand(C, 3, 3, 3, seed=7, lo=-0.2, hi=0.2)
    g = _rand(2, C, 20, 28, seed=8)
    dx_ref, _, _ = O.conv3x3_grads(x, ws, g)
    xg = _nhwc(x).requires_grad_(True)
    wg = ws.cuda().requires_grad_(True)
    y = PF.conv3x3(xg, wg, None, PF.PackedConvWeights())
    calls = []
    real = ops.conv3x3_rgb_in_dgrad
    ops.conv3x3_rgb_in_dgrad = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        y.backward(_nhwc(g))
    finally:
        ops.conv3x3_rgb_in_dgrad = real
    assert calls, "Conv3x3Fn.backward did not take the streaming kernel"
    _close(xg.grad.permute(0, 3, 1, 2).cpu(), dx_ref, 1e-5, "Conv3x3Fn dx")


def test_meanshift():
    from pesr_amd import ops
    x = detrand.image_batch((2, 3, 10, 14), 1)
    w = (torch.eye(3) + _rand(3, 3, seed=2, lo=-0.1, hi=0.1)).view(3, 3, 1, 1)
    b = _rand(3, seed=3, lo=-100, hi=100)
    ref = F.conv2d(x, w, b)
    y = ops.meanshift_fwd(x.cuda().contiguous(), w.cuda(), b.cuda(), x_nchw=True)      # NCHW in -> NHWC out
    _close(_nchw(y), ref, 1e-6, "meanshift fwd nchw-in")
    y2 = ops.meanshift_fwd(_nhwc(x), w.cuda(), b.cuda(), y_nchw=True)                   # NHWC in -> NCHW out
    _close(y2.cpu(), ref, 1e-6, "meanshift fwd nchw-out")
    dy = _rand(2, 3, 10, 14, seed=4)
    xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
    F.conv2d(xr, wr, br).backward(dy)
    dx, dw, db = ops.meanshift_bwd(_nhwc(dy), x.cuda().contiguous(), w.cuda(), x_nchw=True)
    _close(_nchw(dx), xr.grad, 1e-6); _close(dw.cpu(), wr.grad, 1e-5); _close(db.cpu(), br.grad, 1e-5)


def test_pixel_shuffle_bit_exact():
    from pesr_amd import ops
    x = torch.arange(2 * 16 * 3 * 5, dtype=torch.float32).reshape(2, 16, 3, 5)
    y = ops.pixel_shuffle_fwd(_nhwc(x))
    assert torch.equal(_nchw(y), O.pixel_shuffle(x))
    gy = torch.arange(y.numel(), dtype=torch.float32).reshape(2, 4, 6, 10) * 0.5
    gx = ops.pixel_shuffle_bwd(_nhwc(gy))
    assert torch.equal(_nchw(gx), O.pixel_unshuffle(gy))


def test_maxpool_and_relu_mask():
    from pesr_amd import ops
    x = torch.relu(_rand(2, 8, 6, 10, seed=1))
    y = ops.maxpool2x2_fwd(_nhwc(x))
    assert torch.equal(_nchw(y), F.max_pool2d(x, 2, 2))
    xr = _rand(2, 8, 6, 10, seed=1).requires_grad_(True)
    dy = _rand(2, 8, 3, 5, seed=2)
    F.max_pool2d(torch.relu(xr), 2, 2).backward(dy)
    dx = ops.maxpool2x2_bwd(_nhwc(x), _nhwc(dy), relu_in=True)       # pool backward + ReLU mask in one pass
    _close(_nchw(dx), xr.grad, 1e-7)
    g = _rand(2, 8, 6, 10, seed=3)
    out = ops.relu_mask(_nhwc(g), _nhwc(x), _nhwc(g), alpha=0.5)
    _close(_nchw(out), 0.5 * g * (x > 0) + g, 1e-7)


@pytest.mark.parametrize("N,C,H,W,nchw", [(4, 64, 16, 16, False), (4, 512, 2, 2, True), (3, 128, 5, 7, False)])
def test_bn_lrelu(N, C, H, W, nchw):
    from pesr_amd import ops
    x = _rand(N, C, H, W, seed=1, lo=-2, hi=3)
    gamma = _rand(C, seed=2, lo=0.5, hi=1.5); beta = _rand(C, seed=3, lo=-0.3, hi=0.3)
    rm = torch.zeros(C); rv = torch.ones(C)
    xr = x.clone().requires_grad_(True); gr = gamma.clone().requires_grad_(True); br = beta.clone().requires_grad_(True)
    ref = F.leaky_relu(F.batch_norm(xr, rm, rv, gr, br, True, 0.1, 1e-5), 0.2)
    rmg = torch.zeros(C).cuda(); rvg = torch.ones(C).cuda(); nb = torch.zeros((), dtype=torch.long).cuda()
    y, stats = ops.bn_lrelu_fwd(_nhwc(x), gamma.cuda(), beta.cuda(), rmg, rvg, nb, y_nchw=nchw)
    yy = y.cpu() if nchw else _nchw(y)
    _close(yy, ref.detach(), 2e-5, "bn fwd")
    _close(rmg.cpu(), rm, 1e-5, "running_mean"); _close(rvg.cpu(), rv, 1e-4, "running_var"); assert int(nb) == 1
    dy = _rand(N, C, H, W, seed=4)
    ref.backward(dy)
    dyg = dy.cuda().contiguous() if nchw else _nhwc(dy)
    dx, dg, db = ops.bn_lrelu_bwd(_nhwc(x), dyg, gamma.cuda(), beta.cuda(), stats, dy_nchw=nchw)
    _close(_nchw(dx), xr.grad, 5e-5, "bn dx"); _close(dg.cpu(), gr.grad, 5e-5, "dgamma"); _close(db.cpu(), br.grad, 5e-5, "dbeta")


@pytest.mark.parametrize("N,H,W,C,nchw", [(16, 192, 192, 64, False), (2, 37, 50, 64, False), (3, 9, 33, 32, True), (1, 8, 32, 256, False)])
def test_conv_rgb_bn_lrelu_fused_statistics(N, H, W, C, nchw):
    """The 3 -> C conv whose kernel leaves the BatchNorm partial sums (the Discriminator's features.0, reference model/pesr.py:53 +
    model/basic.py:26-30; SURVEY K10's statistics half): conv output, statistics, normalised output and running statistics against
    torch on the CPU in double, and against the two-call path (conv, then bn_lrelu_fwd with its own statistics pass) - the partial
    sums are fp32 per thread and double from there on, so the two agree far inside the tolerance; run to run the bits repeat."""
    from pesr_amd import ops
    x = _rand(N, 3, H, W, seed=11, lo=0, hi=255); w = _rand(C, 3, 3, 3, seed=12, lo=-0.2, hi=0.2)
    gamma = _rand(C, seed=13, lo=0.5, hi=1.5); beta = _rand(C, seed=14, lo=-0.3, hi=0.3)
    zr = F.conv2d(x.double(), w.double(), None, 1, 1)
    rm = torch.zeros(C, dtype=torch.double); rv = torch.ones(C, dtype=torch.double)
    ref = F.leaky_relu(F.batch_norm(zr, rm, rv, gamma.double(), beta.double(), True, 0.1, 1e-5), 0.2)
    rmg = torch.zeros(C).cuda(); rvg = torch.ones(C).cuda(); nb = torch.zeros((), dtype=torch.long).cuda()
    z, y, stats = ops.conv_rgb_bn_lrelu_fwd(_nhwc(x), w.cuda(), gamma.cuda(), beta.cuda(), rmg, rvg, nb, y_nchw=nchw)
    _close(_nchw(z).double(), zr, 1e-5, "conv")
    _close((y.cpu() if nchw else _nchw(y)).double(), ref, 2e-5, "bn(conv)")
    _close(rmg.cpu().double(), rm, 1e-5, "running_mean"); _close(rvg.cpu().double(), rv, 1e-4, "running_var"); assert int(nb) == 1
    mean = zr.mean((0, 2, 3)); var = zr.var((0, 2, 3), unbiased=False)
    _close(stats[0].cpu().double(), mean, 1e-6, "mean"); _close(stats[1].cpu().double(), 1.0 / torch.sqrt(var + 1e-5), 1e-5, "invstd")
    # the two-call path on the same z
    rm2 = torch.zeros(C).cuda(); rv2 = torch.ones(C).cuda(); nb2 = torch.zeros((), dtype=torch.long).cuda()
    y2, stats2 = ops.bn_lrelu_fwd(z, gamma.cuda(), beta.cuda(), rm2, rv2, nb2, y_nchw=nchw)
    _close(stats.cpu(), stats2.cpu(), 1e-6, "statistics vs the stand-alone pass"); _close(y.cpu(), y2.cpu(), 1e-6, "y vs the two-call path")
    z3, y3, stats3 = ops.conv_rgb_bn_lrelu_fwd(_nhwc(x), w.cuda(), gamma.cuda(), beta.cuda(), torch.zeros(C).cuda(), torch.ones(C).cuda(),
                                               torch.zeros((), dtype=torch.long).cuda(), y_nchw=nchw)
    assert torch.equal(y, y3) and torch.equal(stats, stats3) and torch.equal(z, z3)


@pytest.mark.parametrize("M,N,K", [(4, 1024, 2048), (16, 1, 1024), (3, 70, 1000), (16, 1024, 73728), (70, 96, 2048), (5, 100, 1000), (16, 96, 4100),
                                   (32, 1024, 73728), (20, 1, 1024), (32, 130, 1000), (24, 64, 4100)])
def test_linear(M, N, K):
    """(70 rows: more than one 32-row kernel call - a per-GPU batch of 64 must not abort in D's classifier.  17 .. 32 rows: the two-tile forms
    of the MFMA kernels, which the GAN step uses for the classifier on [hr; sr] - one pass over the 302 MB of weights instead of two.)"""
    from pesr_amd import ops
    x = _rand(M, K, seed=1); w = _rand(N, K, seed=2, lo=-0.01, hi=0.01); b = _rand(N, seed=3)
    ref = F.leaky_relu(F.linear(x, w, b), 0.2)
    y = ops.linear_fwd(x.cuda(), w.cuda(), b.cuda(), ops.ACT_LRELU, 0.2)
    _close(y.cpu(), ref, 2e-5, "linear fwd")
    dy = _rand(M, N, seed=4)
    dx = ops.linear_dgrad(dy.cuda(), w.cuda())
    _close(dx.cpu(), dy @ w, 2e-5, "linear dgrad")
    dw, db = ops.linear_wgrad(dy.cuda(), x.cuda())
    _close(dw.cpu(), dy.t() @ x, 2e-5, "linear wgrad"); _close(db.cpu(), dy.sum(0), 1e-5, "linear bgrad")
    # accumulate mode: a second contribution lands in the same buffers (the layer used twice in one backward pass)
    dy2 = _rand(M, N, seed=5); x2 = _rand(M, K, seed=6)
    ops.linear_wgrad(dy2.cuda(), x2.cuda(), dw_out=dw, db_out=db, accumulate=True)
    _close(dw.cpu(), dy.t() @ x + dy2.t() @ x2, 2e-5, "linear wgrad accumulate"); _close(db.cpu(), dy.sum(0) + dy2.sum(0), 1e-5, "bgrad accumulate")


def test_losses_and_adam():
    from pesr_amd import ops
    from oracle import step as OS
    sr = (detrand.image_batch((2, 3, 10, 12), 51) + _rand(2, 3, 10, 12, seed=52, lo=-0.5, hi=0.5)).requires_grad_(True)
    hr = detrand.image_batch((2, 3, 10, 12), 53)
    l1 = F.l1_loss(sr, hr); tv = OS.tv_loss(sr)
    (0.3 * l1 + 1e-3 * tv).backward()
    out, grad = ops.loss_l1_tv(_nhwc(sr.detach()), _nhwc(hr), 0.3 / sr.numel(), 1e-3)
    _close(out.cpu(), torch.stack([l1.detach(), tv.detach()]), 1e-6, "l1/tv")
    _close(_nchw(grad), sr.grad, 1e-6, "l1/tv grad")
    a = _rand(2, 512, 3, 3, seed=1).requires_grad_(True); b = _rand(2, 512, 3, 3, seed=2)
    m = F.mse_loss(a, b); (50 * m).backward()
    out, g = ops.loss_mse(_nhwc(a.detach()), _nhwc(b), 50 * 2.0 / a.numel())
    _close(out.cpu(), m.detach().reshape(1), 1e-6, "mse"); _close(_nchw(g), a.grad, 1e-6, "mse grad")
    # Adam: 3 steps vs torch.optim.Adam
    p = _rand(4096, seed=3).requires_grad_(True)
    opt = torch.optim.Adam([p], lr=5e-5, betas=(0.9, 0.999))
    pg = p.detach().clone().cuda(); mg = torch.zeros(4096).cuda(); vg = torch.zeros(4096).cuda()
    for it in range(1, 4):
        gr = _rand(4096, seed=10 + it, lo=-3, hi=3)
        p.grad = gr.clone(); opt.step()
        ops.adam_step(pg, gr.cuda(), mg, vg, 5e-5, 0.9, 0.999, 1e-8, it)
    _close(pg.cpu(), p.detach(), 1e-6, "adam")


def test_gpu_input_pipeline_bit_exact():
    """Device-side crop + 8-way augmentation + uint8->float vs the oracle's restatement of reference data.py:79-126."""
    import random
    import numpy as np
    from oracle import image as OI
    from pesr_amd.input_pipeline import GpuPatchSampler
    rs = np.random.RandomState(0)
    lrs = [rs.randint(0, 256, (h, w, 3), dtype=np.uint8) for h, w in ((20, 31), (17, 16), (40, 23))]
    hrs = [rs.randint(0, 256, (4 * im.shape[0], 4 * im.shape[1], 3), dtype=np.uint8) for im in lrs]
    samp = GpuPatchSampler(lrs, hrs, torch.device("cuda"))
    picks = [(i % 3, y, x, aug) for i, (y, x, aug) in enumerate([(0, 0, a) for a in range(8)] + [(4, 3, 5), (1, 0, 6), (0, 2, 3)])]
    picks += samp.draw(5, 12, random.Random(3))
    for nhwc in (False, True):
        lr, hr = samp.assemble(picks, 12, nhwc=nhwc)
        assert lr.shape == (len(picks), 3, 12, 12) and hr.shape == (len(picks), 3, 48, 48)
        for b, (i, y, x, aug) in enumerate(picks):
            rl, rh = OI.crop_augment(lrs[i], hrs[i], 12, y, x, aug)
            assert torch.equal(lr[b].cpu(), rl) and torch.equal(hr[b].cpu(), rh), (b, i, y, x, aug)


def test_gpu_input_pipeline_vs_the_reference_fixture_gv12():
    """The same kernel against golden GV12: outputs of the reference's OWN SRDataset._crop / _aug_data / _to_tensor
    (reference data.py:79-126; tests/golden/make_golden_plumbing.py) - every aug_idx, crops on every image edge, both crop types."""
    from helpers import load_golden
    from oracle import detrand
    from pesr_amd.input_pipeline import GpuPatchSampler
    g = load_golden("gv12_crop_aug")
    cases = g["cases"].tolist()
    mk = lambda h, w, s: detrand.image_batch((1, 3, h, w), s)[0].permute(1, 2, 0).contiguous().numpy().astype("uint8")  # noqa: E731
    keys = sorted({(c[0], c[1], c[2], c[3], c[4]) for c in cases})
    samp = GpuPatchSampler([mk(ih, iw, s) for ih, iw, _, s, _ in keys], [mk(4 * ih, 4 * iw, s) for ih, iw, _, _, s in keys], torch.device("cuda"))
    augs = set()
    for img, key in enumerate(keys):
        mine = [(n, c) for n, c in enumerate(cases) if tuple(c[:5]) == key]
        picks = [(img, c[7], c[8], c[9]) for _, c in mine]
        for nhwc in (False, True):
            lr, hr = samp.assemble(picks, key[2], nhwc=nhwc)
            for b, (n, c) in enumerate(mine):
                assert torch.equal(lr[b].cpu(), torch.from_numpy(g[f"inp{n}"]).float()), (n, c)
                assert torch.equal(hr[b].cpu(), torch.from_numpy(g[f"lbl{n}"]).float()), (n, c)
                augs.add(c[9])
    assert augs == set(range(8))


def test_psnr_on_device_bit_identical():
    """utils.compute_PSNR on GPU tensors (device kernel) == the oracle's numpy restatement of reference utils.py:32-41."""
    import importlib.util, os
    from oracle import image as OI
    spec = importlib.util.spec_from_file_location("entry_utils", os.path.join(os.path.dirname(os.path.dirname(__file__)), "utils.py"))
    U = importlib.util.module_from_spec(spec); spec.loader.exec_module(U)
    a = detrand.image_batch((1, 3, 37, 52), 41)
    b = a + _rand(1, 3, 37, 52, seed=42, lo=-30, hi=30)            # includes values outside 0..255 (clipped)
    ref = OI.psnr_y(a, b)
    assert U.compute_PSNR(a.cuda(), b.cuda()) == ref                                                       # NCHW-contiguous
    assert U.compute_PSNR(a.cuda().contiguous(memory_format=torch.channels_last), b.cuda()) == ref         # mixed layouts
    assert abs(U.compute_PSNR(a, b) - ref) < 1e-12                                                         # host path


@pytest.mark.parametrize("N,C,H,W", [(2, 8, 5, 7), (4, 64, 12, 12), (3, 512, 2, 2)])
def test_bn_backward_of_backward(N, C, H, W):
    """pesr_bn_bwd_bwd (the gradient penalty's BatchNorm second-order rule) against autograd through the training-mode
    BatchNorm backward formula in float64 on the CPU."""
    from pesr_amd import ops
    z = _rand(N, C, H, W, seed=1, lo=-2, hi=2); du = _rand(N, C, H, W, seed=2); g = _rand(N, C, H, W, seed=3)
    gamma = _rand(C, seed=4, lo=0.5, hi=1.5)
    zd, dud, gd, gad = (t.double() for t in (z, du, g, gamma))
    zd.requires_grad_(True); dud.requires_grad_(True); gad.requires_grad_(True)
    mu = zd.mean((0, 2, 3), keepdim=True); var = zd.var((0, 2, 3), unbiased=False, keepdim=True)
    s = (var + 1e-5).rsqrt(); xh = (zd - mu) * s
    dz = gad.view(1, C, 1, 1) * s * (dud - dud.mean((0, 2, 3), keepdim=True) - xh * (dud * xh).mean((0, 2, 3), keepdim=True))
    l_du, l_z, l_ga = torch.autograd.grad((dz * gd).sum(), [dud, zd, gad])
    # the GPU side gets the statistics the forward kernel saved
    y, stats = ops.bn_lrelu_fwd(_nhwc(z), gamma.cuda(), torch.zeros(C).cuda(), None, None, None, 1e-5, 0.1, 1.0)
    dz_gpu, _, _ = ops.bn_lrelu_bwd(_nhwc(z), _nhwc(du), gamma.cuda(), gamma.cuda(), stats, 1.0, False, False)
    _close(_nchw(dz_gpu), dz.detach().float(), 2e-5, "dz (first-order backward, slope 1)")
    a, b, c = ops.bn_bwd_bwd(_nhwc(z), _nhwc(du), _nhwc(g), gamma.cuda(), stats)
    _close(_nchw(a), l_du.float(), 2e-5, "dL/d(du)"); _close(_nchw(b), l_z.float(), 5e-5, "dL/dz"); _close(c.cpu(), l_ga.float(), 5e-5, "dL/dgamma")


@pytest.mark.parametrize("Co,Ci", [(64, 3), (128, 64), (512, 512)])
def test_spectral_norm_kernels_vs_oracle(Co, Ci):
    """pesr_spectral_norm_fwd / _bwd (torch.nn.utils.spectral_norm semantics; reference model/basic.py:25) vs the oracle's
    restatement: power iteration (u, v rewritten in place), sigma, the normalised weight, the gradient; eval mode too."""
    from oracle import model as OM
    from pesr_amd import ops
    O_, I = Co, Ci
    gen = torch.Generator().manual_seed(O_ + I)
    w = (torch.rand(O_, I, 3, 3, generator=gen) * 2 - 1) / (I * 9) ** 0.5
    u = torch.nn.functional.normalize(torch.randn(O_, generator=gen), dim=0)
    v = torch.nn.functional.normalize(torch.randn(I * 9, generator=gen), dim=0)
    g = torch.randn(O_, I, 3, 3, generator=gen)
    for update in (True, True, False):            # two training calls in a row (the buffers carry over), then eval
        wr = w.clone().requires_grad_(True)
        ur, vr = u.clone(), v.clone()
        ref = OM.spectral_normalize(wr, ur, vr, update)
        ref.backward(g)
        ud, vd = u.cuda(), v.cuda()
        w_hat, sigma = ops.spectral_norm_fwd(w.cuda(), ud, vd, update)
        _close(w_hat.cpu(), ref.detach(), 2e-6, what=f"w_hat update={update}")
        _close(ud.cpu(), ur, 5e-6, what="u"); _close(vd.cpu(), vr, 5e-6, what="v")
        dw = ops.spectral_norm_bwd(g.cuda(), w_hat, ud, vd, sigma)
        _close(dw.cpu(), wr.grad, 5e-6, what="dw")
        acc = torch.ones(O_, I, 3, 3).cuda()
        ops.spectral_norm_bwd(g.cuda(), w_hat, ud, vd, sigma, dw_out=acc, accumulate=True)
        _close(acc.cpu(), wr.grad + 1.0, 5e-6, what="dw accumulate")
        u, v = ur, vr                              # carry the oracle's buffers into the next round


@pytest.mark.parametrize("gan_type", ["SGAN", "RSGAN", "RaSGAN"])
@pytest.mark.parametrize("side,focal,gamma", [(0, False, 1.0), (1, False, 1.0), (1, True, 1.0), (1, True, 2.0), (1, True, 0.0)])
def test_gan_loss_kernel_vs_oracle(gan_type, side, focal, gamma):
    """pesr_gan_loss_fwd_bwd - the discriminator / generator losses of reference train.py:210-213,244-253 with the reference's
    FocalLoss (model/focal_loss.py:9-13, torch-0.4 gradient) - against the oracle's expressions (oracle/step.py) under
    torch autograd on the CPU: value and the gradients w.r.t. both logit vectors, incl. saturated logits."""
    from oracle import step as OS
    from pesr_amd import ops
    gen = torch.Generator().manual_seed(7)
    B = 16
    r = (torch.randn(B, 1, generator=gen) * 4).requires_grad_(True)
    f = (torch.randn(B, 1, generator=gen) * 4).requires_grad_(True)
    with torch.no_grad():
        r[0] = 30.0; f[1] = -40.0; r[2] = -25.0
    ones, zeros = torch.ones(B, 1), torch.zeros(B, 1)
    bce = F.binary_cross_entropy_with_logits
    lf = (lambda z, t: OS.focal_loss(z, t, gamma)) if focal else bce
    if side == 0:
        ref = {"SGAN": lambda: bce(r, ones) + bce(f, zeros), "RSGAN": lambda: bce(r - f, ones),
               "RaSGAN": lambda: 0.5 * (bce(r - f.mean(), ones) + bce(f - r.mean(), zeros))}[gan_type]()
    else:
        ref = {"SGAN": lambda: lf(f, ones), "RSGAN": lambda: lf(f - r, ones),
               "RaSGAN": lambda: 0.5 * (lf(r - f.mean(), zeros) + lf(f - r.mean(), ones))}[gan_type]()
    (ref * 0.7).backward()
    out, d_r, d_f = ops.gan_loss(r.detach().cuda(), f.detach().cuda(), gan_type, side, focal, gamma, 0.7)
    assert float(out[0]) == pytest.approx(0.7 * float(ref), rel=2e-6, abs=1e-9)
    gr = r.grad if r.grad is not None else torch.zeros_like(r)
    _close(d_f.cpu(), f.grad, 5e-6, "d_fake")
    if float(gr.abs().max()) > 0:
        _close(d_r.cpu(), gr, 5e-6, "d_real")
    else:
        assert float(d_r.abs().max()) == 0.0
