"""GPU: BatchNorm sums out of the conv kernels' epilogues (round 6; reference model/basic.py:26-30 conv -> BatchNorm2d(train)
-> LeakyReLU(0.2), model/pesr.py:53-65 the Discriminator's eight BasicBlocks).

* forward: the conv kernel in front of a BatchNorm (direct stride-2 / stride-1 kernel, F(4,3) kernel) leaves sum(z), sum(z^2) per pixel
  tile -> statistics / y / running statistics against torch in double, against the stand-alone statistics pass on the same z, and
  bit-identical run to run (no atomics anywhere);
* backward: the input-gradient kernel of the NEXT conv (the one-launch stride-2 form, the F(4,3) kernel, the direct stride-1 kernel) stores
  the gradient already multiplied by lrelu'(bn(z)) and leaves sum(g'), sum(g' xhat) -> dz, dgamma, dbeta against torch autograd in double and
  against the un-fused path;
* the Discriminator through its chained forward (fused) against the same network with PESR_BN_FUSE off, and the shapes the planner
  refuses (split-K layers) fall back.
"""
import pytest
import torch
import torch.nn.functional as F

from oracle import detrand

pytestmark = pytest.mark.gpu


def _rand(*shape, seed=0, lo=-1.0, hi=1.0):
    return detrand.uniform(shape, seed, lo, hi)


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous().cuda()


def _nchw(y):
    return y.permute(0, 3, 1, 2).cpu()


def _close(a, b, rel, what=""):
    scale = b.abs().max().item() + 1e-30
    err = (a.double() - b.double()).abs().max().item()
    assert err <= rel * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e} (rel {err / scale:.3e} > {rel})"


def _packed(w, x_shape, stride, kind, mode):
    """kind: 'direct' | 'wino4'; mode 0 forward / 1 input gradient"""
    from pesr_amd import ops
    wg = w.cuda()
    if kind == "wino4":
        return ops.pack_conv3x3_wino4(wg, mode)
    return ops.pack_conv3x3(wg, mode)


# (N, H, W, Cin, Cout, stride, kernel): the Discriminator's features.1 / .3 / .5 (direct stride-2 kernel: 64-, 144- and 48-pixel tiles),
# features.2 / .4 (F(4,3)), ragged and small shapes, the direct stride-1 kernel
FWD = [(16, 192, 192, 64, 64, 2, "direct"), (16, 96, 96, 128, 128, 2, "direct"), (16, 48, 48, 256, 256, 2, "direct"),
       (2, 37, 50, 64, 64, 2, "direct"), (16, 96, 96, 64, 128, 1, "wino4"), (16, 48, 48, 128, 256, 1, "wino4"), (3, 20, 28, 64, 128, 1, "direct"),
       (16, 40, 44, 64, 64, 1, "wino4"),
       # split-K layers (features.7 / .6): the sums come from the finish kernel that adds the slabs
       (16, 24, 24, 512, 512, 2, "direct"), (16, 24, 24, 256, 512, 1, "wino4")]


@pytest.mark.parametrize("N,H,W,Cin,Cout,stride,kind", FWD)
def test_conv_bn_forward_statistics_from_the_epilogue(N, H, W, Cin, Cout, stride, kind):
    from pesr_amd import ops
    x = _rand(N, Cin, H, W, seed=21, lo=-1, hi=2); w = _rand(Cout, Cin, 3, 3, seed=22, lo=-0.05, hi=0.05)
    gamma = _rand(Cout, seed=23, lo=0.5, hi=1.5); beta = _rand(Cout, seed=24, lo=-0.3, hi=0.3)
    zr = F.conv2d(x.double(), w.double(), None, stride, 1)
    rm = torch.zeros(Cout, dtype=torch.double); rv = torch.ones(Cout, dtype=torch.double)
    ref = F.leaky_relu(F.batch_norm(zr, rm, rv, gamma.double(), beta.double(), True, 0.1, 1e-5), 0.2)
    xg = _nhwc(x)
    wp = _packed(w, xg.shape, stride, kind, 0)
    res = ops.conv3x3_fwd_bn_stats(xg, wp, None, Cout, stride)
    assert res is not None, "the planner refused a shape this test expects the fused kernel to cover"
    z, part = res
    assert part.shape[1:] == (2, Cout) and part.shape[0] >= 1
    _close(_nchw(z), zr, 1e-5, "conv")
    rmg = torch.zeros(Cout).cuda(); rvg = torch.ones(Cout).cuda(); nb = torch.zeros((), dtype=torch.long).cuda()
    y, stats = ops.bn_finalize_apply(z, part, gamma.cuda(), beta.cuda(), rmg, rvg, nb)
    _close(_nchw(y), ref, 3e-5, "bn(conv)")
    mean = zr.mean((0, 2, 3)); var = zr.var((0, 2, 3), unbiased=False)
    _close(stats[0].cpu(), mean, 2e-6 * (zr.abs().max().item() / (mean.abs().max().item() + 1e-30)), "mean")
    _close(stats[1].cpu(), 1.0 / torch.sqrt(var + 1e-5), 1e-5, "invstd")
    _close(rmg.cpu(), rm, 1e-5, "running_mean"); _close(rvg.cpu(), rv, 1e-4, "running_var"); assert int(nb) == 1
    # the sums themselves: sum over rows == sum over pixels (double) of z and z^2
    zs = _nchw(z).double()
    _close(part[:, 0].double().sum(0).cpu(), zs.sum((0, 2, 3)), 1e-6 * (zs.abs().sum((0, 2, 3)).max().item() / (zs.sum((0, 2, 3)).abs().max().item() + 1e-30)), "sum z")
    _close(part[:, 1].double().sum(0).cpu(), (zs * zs).sum((0, 2, 3)), 1e-6, "sum z^2")
    # the stand-alone statistics pass on the same z
    y2, stats2 = ops.bn_lrelu_fwd(z, gamma.cuda(), beta.cuda(), torch.zeros(Cout).cuda(), torch.ones(Cout).cuda(), torch.zeros((), dtype=torch.long).cuda())
    _close(stats.cpu(), stats2.cpu(), 2e-6, "statistics vs the stand-alone pass"); _close(y.cpu(), y2.cpu(), 2e-6, "y vs the two-call path")
    # bit-identical run to run, and the conv output identical to the plain kernel's
    z3, part3 = ops.conv3x3_fwd_bn_stats(xg, wp, None, Cout, stride)
    assert torch.equal(z, z3) and torch.equal(part, part3)
    zp = ops.conv3x3_fwd(xg, wp, None, Cout, stride)
    assert torch.equal(z, zp), "the epilogue sums must not change the conv's own output"


# (N, H, W of the BatchNorm'd tensor = the next conv's input, C of it, Cout of the next conv, its stride, kernel)
BWD = [(16, 192, 192, 64, 64, 2, "direct"), (16, 96, 96, 128, 128, 2, "direct"), (16, 48, 48, 256, 256, 2, "direct"), (16, 24, 24, 512, 512, 2, "direct"),
       (2, 37, 50, 64, 64, 2, "direct"), (16, 96, 96, 64, 128, 1, "wino4"), (3, 20, 28, 128, 64, 1, "direct"), (16, 40, 44, 64, 64, 1, "wino4"),
       # split-K input gradients (of features.4 / .6): masked gradient and sums from the finish kernel
       (16, 48, 48, 128, 256, 1, "wino4"), (16, 24, 24, 256, 512, 1, "wino4")]


@pytest.mark.parametrize("N,H,W,C,Cnext,stride,kind", BWD)
def test_bn_backward_sums_from_the_next_convs_input_gradient_kernel(N, H, W, C, Cnext, stride, kind):
    from pesr_amd import ops
    z = _rand(N, C, H, W, seed=31, lo=-2, hi=3)
    gamma = _rand(C, seed=32, lo=0.5, hi=1.5); beta = _rand(C, seed=33, lo=-0.3, hi=0.3)
    w = _rand(Cnext, C, 3, 3, seed=34, lo=-0.05, hi=0.05)
    OH, OW = (H - 1) // stride + 1, (W - 1) // stride + 1
    dy = _rand(N, Cnext, OH, OW, seed=35)
    # reference in double: z -> bn -> lrelu -> conv(w, stride); gradient dy at the conv's output
    zr = z.double().requires_grad_(True); gr = gamma.double().requires_grad_(True); br = beta.double().requires_grad_(True)
    yr = F.leaky_relu(F.batch_norm(zr, None, None, gr, br, True, 0.1, 1e-5), 0.2)
    F.conv2d(yr, w.double(), None, stride, 1).backward(dy.double())
    zg, dyg = _nhwc(z), _nhwc(dy)
    y, stats = ops.bn_lrelu_fwd(zg, gamma.cuda(), beta.cuda(), torch.zeros(C).cuda(), torch.ones(C).cuda(), torch.zeros((), dtype=torch.long).cuda())
    wpd = _packed(w, zg.shape, stride, kind, 1)
    res = ops.conv3x3_dgrad_bn_sums(dyg, wpd, tuple(zg.shape), stride, zg, stats, gamma.cuda(), beta.cuda(), 0.2)
    assert res is not None, "the planner refused a shape this test expects the fused kernel to cover"
    gm, part = res
    dz, dg, db = ops.bn_lrelu_bwd_fused(zg, gm, part, gamma.cuda(), beta.cuda(), stats)
    _close(_nchw(dz), zr.grad, 5e-5, "dz"); _close(dg.cpu(), gr.grad, 5e-5, "dgamma"); _close(db.cpu(), br.grad, 5e-5, "dbeta")
    # the un-fused path: plain input gradient, then the two-pass BatchNorm backward
    gy = ops.conv3x3_dgrad(dyg, wpd, tuple(zg.shape), stride)
    dz2, dg2, db2 = ops.bn_lrelu_bwd(zg, gy, gamma.cuda(), beta.cuda(), stats)
    _close(dz.cpu(), dz2.cpu(), 1e-5, "dz vs the un-fused path"); _close(dg.cpu(), dg2.cpu(), 1e-5, "dgamma vs un-fused"); _close(db.cpu(), db2.cpu(), 1e-5, "dbeta vs un-fused")
    # g' is exactly the plain kernel's gradient times lrelu'(bn(z)) - the same bits
    # (away from the kink: at |bn(z)| ~ 1e-7 the kernel's fused multiply-add and torch's two roundings may disagree on the sign)
    zz = gamma.cuda() * ((zg - stats[0]) * stats[1]) + beta.cuda()
    far = zz.abs() > 1e-5
    assert torch.equal(gm[far], torch.where(zz > 0, gy, gy * 0.2)[far]), "masked gradient"
    assert float(far.float().mean()) > 0.999
    # accumulate form (second use of a layer inside one backward pass) and run-to-run bits
    dg3, db3 = dg.clone(), db.clone()
    ops.bn_lrelu_bwd_fused(zg, gm, part, gamma.cuda(), beta.cuda(), stats, dgamma_out=dg3, dbeta_out=db3, accumulate=True)
    _close(dg3.cpu(), 2 * dg.cpu(), 1e-6, "accumulated dgamma")
    gm4, part4 = ops.conv3x3_dgrad_bn_sums(dyg, wpd, tuple(zg.shape), stride, zg, stats, gamma.cuda(), beta.cuda(), 0.2)
    assert torch.equal(gm, gm4) and torch.equal(part, part4)


def test_planner_rows_for_the_discriminators_layers():
    """Rows of sums per call: one per pixel tile where the conv kernel's epilogue leaves them, 256 (the finish kernel's workgroups) where
    split-K slabs are summed first; 0 only for what neither covers (a packing without the epilogue: the caller runs the stand-alone passes)."""
    from pesr_amd import ops
    assert ops.conv_bn_rows(0, 16, 192, 192, 64, 64, 2) == 16 * 144   # features.1: 64-pixel tiles
    assert ops.conv_bn_rows(0, 16, 24, 24, 512, 512, 2) == 256        # features.7 forward (split-K 2): the finish kernel
    assert ops.conv_bn_rows(2, 16, 24, 24, 256, 512, 1) == 256        # features.6 forward (F(4,3), split-K)
    assert ops.conv_bn_rows(2, 16, 48, 48, 256, 128, 1) == 256        # input gradient of features.4
    assert ops.conv_bn_rows(1, 16, 192, 192, 64, 64, 2) > 0 and ops.conv_bn_rows(2, 16, 96, 96, 64, 128, 1) > 0
    x = _nhwc(_rand(2, 64, 20, 20, seed=41)); w = _rand(128, 64, 3, 3, seed=42, lo=-0.02, hi=0.02)
    assert ops.conv3x3_fwd_bn_stats(x, ops.pack_conv3x3_wino(w.cuda(), 0), None, 128, 1) is None      # F(2,3) packing: no epilogue


@pytest.mark.parametrize("ps", [8, 24])
def test_discriminator_fused_chain_equals_unfused(monkeypatch, ps):
    """reference model/pesr.py:40-81 at two sizes: logits, input gradient, every parameter gradient and the running statistics of the
    chained (fused) forward / backward against the same network with the fusion switched off."""
    from helpers import dis_sd
    from model import Discriminator
    from pesr_amd import ops

    def run(fuse):
        monkeypatch.setattr(ops, "USE_BN_FUSE", fuse)
        ops._BN_ROWS.clear()
        D = Discriminator({"patch_size": ps, "spectral_norm": False}); D.load_state_dict(dis_sd(ps)); D = D.cuda()
        a = detrand.image_batch((4, 3, 4 * ps, 4 * ps), 51).cuda()
        b = detrand.image_batch((4, 3, 4 * ps, 4 * ps), 52).cuda().requires_grad_(True)
        o1, o2 = D(a), D(b)                                   # two uses of every layer in one backward pass
        F.binary_cross_entropy_with_logits(o1 - o2, torch.ones(4, 1, device="cuda")).backward()
        return o1.detach().cpu(), o2.detach().cpu(), b.grad.cpu(), {k: p.grad.cpu() for k, p in D.named_parameters()}, \
            {k: v.cpu() for k, v in D.state_dict().items() if "running" in k}
    try:
        f = run(True)
        u = run(False)
    finally:
        ops._BN_ROWS.clear()
    _close(f[0], u[0], 2e-5, "o1"); _close(f[1], u[1], 2e-5, "o2"); _close(f[2], u[2], 2e-3, "input gradient")
    for k in f[3]:
        if u[3][k].abs().max() > 1e-10:
            _close(f[3][k], u[3][k], 2e-3, "grad " + k)        # (BatchNorm over 4 samples + LeakyReLU kinks: fp32 noise of a few 1e-4)
    for k in f[4]:
        _close(f[4][k], u[4][k], 1e-5, k)
