"""GPU parity of the OPTIONAL split-bf16 mode (`--precision split-bf16`; pesr_amd/csrc/conv3x3_bf16x3.hip): every fp32
operand of a covered stride-1 3x3 conv is split into hi + lo bf16 terms and every product into three bf16 MFMA products.  Its
measured error (3.6 .. 4.7e-6 of the output maximum, profiles/r04_split_bf16_numerics.txt) lies inside the tolerances the fp32
kernels are held to, so - unlike the plain bf16 mode - it has NO oracle restatement of its own: every test here compares with the
reference's fp32 arithmetic (oracle/ops.py, oracle/model.py, the goldens made by importing the reference): kernel level at the fp32
kernels' own 1e-5 of the maximum, outputs and losses at the fp32 tolerances (2e-3 on the 0..255 scale / 1e-5, 5e-5).  THE MODE'S OWN
TOLERANCE is on whole-network gradients: through the benchmarked Generator's 67 convs the 3e-6 per layer add up to 2e-5 .. 3e-4 of
a tensor's maximum against float64 (measured, GV2b: median 5e-5, worst upsample.4.weight 3.0e-4 - a sum of activations times the
L1 loss's +-1/N, i.e. heavy cancellation; the F(4,3) fp32 dispatch: median <= 1e-5, every such tensor <= 1e-4), so the gradient bound
here is 5e-4 with the median at 1e-4, not SURVEY 8c's 1e-4 - which is why this is a separate, separately-toleranced row and never
the default."""
import warnings

import numpy as np
import pytest
import torch

from helpers import close, gen_sd, grads_vs_fp64, load_golden
from oracle import detrand
from oracle import model as OM
from oracle import ops as O

pytestmark = pytest.mark.gpu
warnings.filterwarnings("ignore", message=".*pretrained vgg19.*")


@pytest.fixture(autouse=True)
def _split_mode():
    from pesr_amd import ops
    old = ops.PRECISION
    ops.set_precision("split-bf16")
    yield
    ops.set_precision(old)


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous().cuda()


def _nchw(y):
    return y.permute(0, 3, 1, 2).cpu()


def _close(a, b, rel, what=""):
    scale = b.abs().max().item() + 1e-30
    err = (a.double() - b.double()).abs().max().item()
    assert err <= rel * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e} (rel {err / scale:.3e} > {rel})"


CASES = [
    # N, H, W, Cin, Cout
    (16, 48, 48, 256, 256),    # the G body shape (256-channel workgroups, 12 x 12 tiles)
    (2, 48, 48, 64, 128),      # two 32-channel chunks, 128-channel workgroups
    (3, 37, 50, 32, 128),      # ragged tiles in both directions, one chunk
    (2, 96, 96, 128, 256),     # 96-wide rows
    (4, 24, 24, 512, 512),     # sixteen chunks, small images
]


@pytest.mark.parametrize("N,H,W,Cin,Cout", CASES)
def test_split_conv_fwd_dgrad_vs_fp32_oracle(N, H, W, Cin, Cout):
    """Forward with every fused epilogue and the input gradient (mode-1 packing) against the oracle's fp32 conv at the fp32 kernels'
    own bound (1e-5 of the maximum), and against a float64 conv at the measured 6e-6."""
    from pesr_amd import ops
    x = _rand(N, Cin, H, W, seed=1)
    w = _rand(Cout, Cin, 3, 3, seed=2, scale=0.1)
    b = _rand(Cout, seed=3)
    ref = O.conv3x3(x, w, b, 1)
    ref64 = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=1)
    wf = ops.pack_conv3x3_bf16x3(w.cuda(), 0)
    assert isinstance(wf, ops.Bf16x3Packed) and wf.t.numel() == 2 * 9 * Cin * Cout
    xg = _nhwc(x)
    y = ops.conv3x3_fwd(xg, wf, b.cuda(), Cout)
    _close(_nchw(y), ref, 1e-5, "forward vs fp32 oracle")
    _close(_nchw(y), ref64, 6e-6, "forward vs fp64")
    y = ops.conv3x3_fwd(xg, wf, b.cuda(), Cout, act=ops.ACT_RELU)
    _close(_nchw(y), torch.relu(ref), 1e-5, "bias + ReLU")
    skip, mk = _rand(N, Cout, H, W, seed=4), _rand(N, Cout, H, W, seed=5)
    y = ops.conv3x3_fwd(xg, wf, b.cuda(), Cout, alpha=0.1, skip=_nhwc(skip), mask=_nhwc(mk))
    _close(_nchw(y), torch.where(mk > 0, ref * 0.1, torch.zeros_like(ref)) + skip, 1e-5, "scale, mask, skip")
    y = ops.conv3x3_fwd(xg, wf, b.cuda(), Cout, act=ops.ACT_LRELU, slope=0.2)
    _close(_nchw(y), torch.nn.functional.leaky_relu(ref, 0.2), 1e-5, "LeakyReLU")
    if Cin % 128 == 0:          # the input gradient runs the kernel with Cin / Cout swapped: its "Cout" is the forward Cin
        dy = _rand(N, Cout, H, W, seed=6)
        dx_ref, _, _ = O.conv3x3_grads(x, w, dy)
        wd = ops.pack_conv3x3_bf16x3(w.cuda(), 1)
        dx = ops.conv3x3_dgrad(_nhwc(dy), wd, (N, H, W, Cin), mask=xg, skip=xg, alpha=0.5)
        _close(_nchw(dx), torch.where(x > 0, 0.5 * dx_ref, torch.zeros_like(dx_ref)) + x, 1e-5, "input gradient, mask + skip")
    # bit-reproducible
    y1 = ops.conv3x3_fwd(xg, wf, b.cuda(), Cout)
    y2 = ops.conv3x3_fwd(xg, wf, b.cuda(), Cout)
    assert torch.equal(y1, y2)


def test_split_conv_pixel_shuffle_fused():
    """The upsampler convs (reference model/basic.py:56-59): PixelShuffle fused into the store, pixel-unshuffle into the input
    gradient's loads."""
    from pesr_amd import ops
    N, H, W, Cin, C = 2, 24, 24, 256, 256
    Cout = 4 * C
    x = _rand(N, Cin, H, W, seed=1)
    w = _rand(Cout, Cin, 3, 3, seed=2, scale=0.05)
    b = _rand(Cout, seed=3)
    ref = O.pixel_shuffle(O.conv3x3(x, w, b))
    wf = ops.pack_conv3x3_bf16x3(w.cuda(), 0, ps=True)
    y = ops.conv3x3_fwd(_nhwc(x), wf, ops.pack_bias_ps(b.cuda()), Cout, ps_out=True)
    assert y.shape == (N, 2 * H, 2 * W, C)
    _close(_nchw(y), ref, 1e-5, "fused PixelShuffle store")
    dys = _rand(N, C, 2 * H, 2 * W, seed=7)
    dx_ref, _, _ = O.conv3x3_grads(x, w, O.pixel_unshuffle(dys))
    wd = ops.pack_conv3x3_bf16x3(w.cuda(), 1, ps=True)
    dx = ops.conv3x3_dgrad(_nhwc(dys), wd, (N, H, W, Cin), ps_in=True)
    _close(_nchw(dx), dx_ref, 1e-5, "fused pixel-unshuffle load")


def test_split_mode_dispatch_and_repack():
    """Which layers the mode takes (stride 1, Cin % 32 == 0, Cout % 128 == 0, enough workgroups) and that the batched re-pack after an
    optimizer step refreshes the split packings in place."""
    from pesr_amd import functional as PF
    from pesr_amd import ops
    assert ops.bf16x3_eligible(16, 48, 48, 256, 256) and ops.bf16x3_eligible(16, 96, 96, 256, 1024, ps_out=True)
    assert ops.bf16x3_eligible(16, 96, 96, 1024, 256, ps_in=True)
    assert not ops.bf16x3_eligible(16, 48, 48, 256, 256, stride=2) and not ops.bf16x3_eligible(16, 192, 192, 64, 64)
    assert not ops.bf16x3_eligible(16, 12, 12, 512, 512)            # 64 workgroups: stays on the fp32 F(4,3) kernel
    assert not ops.bf16_eligible(16, 48, 48, 256, 256) and not ops.wgrad_bf16_eligible(16, 48, 48, 256, 256)
    w = torch.nn.Parameter((_rand(128, 64, 3, 3, seed=1) * 0.1).cuda())
    cache = PF.PackedConvWeights()
    p1 = cache.for_fwd(w, (8, 48, 48, 64))                          # 8 x 16 tiles x one 128-channel n-tile = 128 workgroups
    d1 = cache.for_dgrad(w, (8, 48, 48, 64))                        # dgrad problem: Cout = 64: not covered -> fp32 packing
    assert isinstance(p1, ops.Bf16x3Packed) and not isinstance(d1, ops.Bf16x3Packed)
    before = p1.t.clone()
    with torch.no_grad():
        w.mul_(1.5)                                                 # (bumps _version: the cached packing is stale)
    PF.repack_all([w])
    assert not torch.equal(p1.t, before)
    assert torch.equal(p1.t, ops.pack_conv3x3_bf16x3(w.detach(), 0).t)
    ops.set_precision("fp32")
    assert not ops.bf16x3_eligible(16, 48, 48, 256, 256)


def test_split_mode_generator_vs_fp32_oracle():
    """A 128-channel Generator (every body / upsampler conv on the split kernel) forward + L1 backward against the oracle's fp32
    autograd: outputs at the G-forward tolerance, every well-conditioned gradient tensor at the mode's 5e-4 of its maximum."""
    from model import Generator
    from pesr_amd import functional as PF
    from pesr_amd import ops
    from pesr_amd.model.basic import nhwc
    C, depth = 128, 3
    sd = gen_sd(C, depth)
    G = Generator({"num_channels": C, "depth": depth, "res_scale": 0.1}); G.load_state_dict(sd); G = G.cuda()
    lr = detrand.image_batch((8, 3, 48, 48), 1234)
    hr = detrand.image_batch((8, 3, 192, 192), 1235)
    ops.FLOPS.start()
    sr = G(lr.cuda())
    PF.l1_loss(nhwc(sr), nhwc(hr.cuda().contiguous(memory_format=torch.channels_last))).backward()
    fl = ops.FLOPS.stop()
    fam = fl["by_kernel_family"]
    split = [k for k in fam if k.startswith("split-bf16")]
    # forward + input gradient of the body convs (the 128 -> 512 upsampler convs feed a PixelShuffle, whose fused form needs
    # Cout % 1024 == 0; weight gradients stay fp32): > 1/3 of the network's flops
    assert split and fam[split[0]][0] >= 0.35 * fl["algorithmic"], fam
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = OM.generator_forward(leaves, lr, depth, 0.1)
    (ref - hr).abs().mean().backward()
    close(sr, ref.detach(), 1e-5, 2e-3, "sr")
    # float64 truth of the same computation: where fp32 itself cannot deliver 1e-4 (the convs in front of a ReLU, the sums of
    # sign(sr - hr): tests/test_fullsize_gpu.py) the bound is 3 x the fp32 oracle's own distance to it (helpers.grads_vs_fp64's rule)
    l64 = {k: v.double().clone().requires_grad_(True) for k, v in sd.items()}
    (OM.generator_forward(l64, lr.double(), depth, 0.1) - hr.double()).abs().mean().backward()
    worst, floor = 0.0, 0.0
    errs = {}
    import re
    for k, p in G.named_parameters():
        g64 = l64[k].grad
        mx = float(g64.abs().max())
        d = (p.grad.cpu().double() - g64).abs() / mx
        e, e_ref = float(d.max()), float((leaves[k].grad.double() - g64).abs().max()) / mx
        errs[k] = e
        floor = max(floor, e_ref)
        if re.fullmatch(r"body\.\d+\.body\.0\.(weight|bias)", k):
            # The conv in front of a ResBlock's ReLU: a unit whose pre-activation lies within the arithmetic's noise of zero (3e-6 of
            # the maximum here, 1e-6 on the F(4,3) kernels, 1e-7 on the CPU) flips its mask - a discrete event that moves one output
            # channel's row of dW by ~1 / sqrt(pixels) of its scale (one flip: ~2e-3 of the maximum at 8 x 48 x 48 pixels; the fp32
            # dispatches show the same class, tests/test_fullsize_gpu.py).  What a wrong scale, mask or tap would break is the BULK of
            # the tensor: the median element must sit at 1e-5, the worst within a few flips.
            assert float(d.median()) <= 1e-5 and e <= 1e-2, f"grad {k}: median {float(d.median()):.2e}, max {e:.2e} of the maximum vs fp64"
        else:
            tol = max(5e-4, 3.0 * e_ref)
            assert e <= tol, f"grad {k}: {e:.2e} of the maximum vs fp64 > {tol:.2e} (the fp32 oracle's own error: {e_ref:.2e})"
            worst = max(worst, e / tol)
    med = sorted(errs.values())[len(errs) // 2]
    print(f"split-bf16 generator: median tensor error vs fp64 {med:.2e}, worst error / allowance (well-conditioned tensors) {worst:.2f}, "
          f"fp32 oracle's worst {floor:.2e}")


def test_split_mode_gv2b_generator_full_batch16():
    """The benchmarked Generator (256 ch x 32 blocks, [16,3,48,48]) in the split mode against the goldens made by importing the
    reference: the same assertions as the fp32 dispatches' GV2b test (tests/test_fullsize_gpu.py)."""
    from model import Generator
    from pesr_amd import functional as PF
    from pesr_amd import ops
    from pesr_amd.model.basic import nhwc
    assert ops.bf16x3_eligible(16, 48, 48, 256, 256)
    g = load_golden("gv2b_generator_full_b16")
    G = Generator({"num_channels": 256, "depth": 32, "res_scale": 0.1}); G.load_state_dict(gen_sd(256, 32)); G = G.cuda()
    lr = detrand.image_batch((16, 3, 48, 48), 1234).cuda()
    hr = detrand.image_batch((16, 3, 192, 192), 1235).cuda()
    sr = G(lr)
    flat = sr.contiguous().reshape(-1)
    close(flat[torch.from_numpy(g["sr_idx"]).cuda()], g["sr_val"], 1e-5, 2e-3, "sr samples")
    close(sr.sum(), g["sr_sum"], 1e-5, what="sr sum")
    loss = PF.l1_loss(nhwc(sr), nhwc(hr.contiguous(memory_format=torch.channels_last)))
    close(loss, g["loss"], 1e-5, what="l1")
    loss.backward()
    g64 = load_golden("gv2b_fp64")
    params = dict(G.named_parameters())
    detail = {}
    n, w = grads_vs_fp64(lambda k: params[k].grad, g, g64, detail=detail, record="gv2b[split-bf16]")
    SIGN_SUMS = {"add_mean.bias", "upsample.4.bias", "body.0.body.0.bias", "body.0.body.0.weight"}
    for k, (e_ours, e_ref) in sorted(detail.items(), key=lambda kv: kv[1][0]):
        print(f"  {k:28s} ours {e_ours:.2e}   reference fp32 {e_ref:.2e}")
        if k not in SIGN_SUMS:
            assert e_ours <= 5e-4, f"grad {k}: {e_ours:.2e} of the maximum > 5e-4 (the mode's stated gradient tolerance)"
    errs = sorted(e for e, _ in detail.values())
    assert errs[len(errs) // 2] <= 1e-4, f"median gradient error {errs[len(errs) // 2]:.2e}"
    print(f"[split-bf16] worst gradient error / allowance: {w[0]:.2f} ({w[1]})")
