"""CPU: the oracle (oracle/) against golden vectors made by importing the reference (tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import detrand
from oracle import model as OM
from oracle import step as OS

from helpers import dis_sd, gen_sd, vgg_sd  # noqa: E402
from helpers import load_golden as load  # noqa: E402


def close(a, b, rtol=1e-6, atol=0.0):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = np.abs(b).max() + 1e-30
    assert a.shape == b.shape
    assert np.abs(a - b).max() <= rtol * scale + atol, (np.abs(a - b).max(), scale)


def test_gv1_generator_small():
    g = load("gv1_generator_small")
    sd = gen_sd(16, 2)
    assert list(sd.keys()) == [str(k) for k in g["keys"]]          # state_dict key order = reference's
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    lr = detrand.image_batch((2, 3, 12, 12), 1234)
    hr = detrand.image_batch((2, 3, 48, 48), 1235)
    sr = OM.generator_forward(leaves, lr, 2, 0.1)
    close(sr.detach(), g["sr"], 1e-6)
    loss = F.l1_loss(sr, hr)
    close(loss.item(), g["loss"], 1e-6)
    loss.backward()
    for k, v in leaves.items():
        close(v.grad, g["grad." + k], 2e-5)


def test_gv2c_generator_full_batch1():
    """The oracle at the reference's inference shape (reference test.py:100-106, BASELINE config 1): 256 ch x 32 blocks,
    [1,3,48,48] -> [1,3,192,192], sampled output and gradients under L1 against the reference's own (fp32) values."""
    g = load("gv2c_generator_full_b1")
    leaves = {k: v.clone().requires_grad_(True) for k, v in gen_sd(256, 32).items()}
    lr = detrand.image_batch((1, 3, 48, 48), 1234)
    hr = detrand.image_batch((1, 3, 192, 192), 1235)
    sr = OM.generator_forward(leaves, lr, 32, 0.1)
    assert sr.shape == (1, 3, 192, 192)
    close(sr.detach().reshape(-1)[g["sr_idx"]], g["sr_val"], 1e-6)
    close(sr.sum().item(), g["sr_sum"], 1e-6)
    loss = F.l1_loss(sr, hr)
    close(loss.item(), g["loss"], 1e-6)
    loss.backward()
    for key in [k[5:] for k in g.files if k.startswith("gidx.")]:
        close(leaves[key].grad.reshape(-1)[g["gidx." + key]], g["gval." + key], 0.0, 2e-5 * float(g["gmax." + key]))


def test_gv3_pixel_shuffle_bit_exact():
    g = load("gv3_pixel_shuffle")
    x = torch.arange(2 * 16 * 3 * 5, dtype=torch.float32).reshape(2, 16, 3, 5)
    y = F.pixel_shuffle(x, 2)
    assert np.array_equal(y.numpy(), g["y"])
    # explicit index formula (reference semantics: out[n,c,2h+i,2w+j] = in[n,4c+2i+j,h,w])
    n, c, h, w = 1, 3, 2, 4
    for i in range(2):
        for j in range(2):
            assert y[n, c, 2 * h + i, 2 * w + j] == x[n, 4 * c + 2 * i + j, h, w]
    gy = torch.arange(y.numel(), dtype=torch.float32).reshape(y.shape) * 0.5
    assert np.array_equal(F.pixel_unshuffle(gy, 2).numpy(), g["gx"])


def test_gv4_discriminator_small():
    g = load("gv4_discriminator_small")
    sd = dis_sd(8)
    leaves = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone())
              for k, v in sd.items()}
    a = detrand.image_batch((4, 3, 32, 32), 21)
    b = detrand.image_batch((4, 3, 32, 32), 22).requires_grad_(True)
    o1 = OM.discriminator_forward(leaves, a)
    o2 = OM.discriminator_forward(leaves, b)
    close(o1.detach(), g["o1"], 1e-5)
    close(o2.detach(), g["o2"], 1e-5)
    l = F.binary_cross_entropy_with_logits(o1 - o2, torch.ones(4, 1))
    l.backward()
    close(b.grad, g["gin"], 1e-4)
    for k, v in leaves.items():
        if v.requires_grad:
            close(v.grad.reshape(-1)[g["gidx." + k]], g["gval." + k], 1e-4, atol=1e-4 * float(g["gmax." + k]))
        elif "running" in k:
            close(v, g["buf." + k], 1e-5)
        else:
            assert int(v) == int(g["buf." + k]) == 2


def test_gv5_focal_forward_and_grad():
    g = load("gv5_focal")
    x = torch.from_numpy(g["x"])
    for gamma in (0, 1, 2):
        for t in (0, 1):
            tt = torch.full_like(x, float(t))
            per = torch.stack([OS.focal_loss(x[i:i + 1], tt[i:i + 1], gamma) for i in range(x.size(0))])
            close(per, g[f"f_g{gamma}_t{t}"], 1e-6, atol=1e-9)
            close(OS.focal_loss(x, tt, gamma).item(), g[f"mean_g{gamma}_t{t}"], 1e-6)
            # backward: autograd of the composite == closed form
            xr = x.clone().requires_grad_(True)
            OS.focal_loss(xr, tt, gamma).backward()
            close(xr.grad, OS.focal_loss_grad_closed_form(x, tt, gamma), 1e-5, atol=1e-9)


def test_gv6_losses():
    g = load("gv6_losses")
    s = (detrand.image_batch((2, 3, 10, 12), 51) + detrand.uniform((2, 3, 10, 12), 52, -0.5, 0.5)).requires_grad_(True)
    h = detrand.image_batch((2, 3, 10, 12), 53)
    l1 = F.l1_loss(s, h)
    tv = OS.tv_loss(s)
    (l1 + tv).backward()
    close(l1.item(), g["l1"]); close(tv.item(), g["tv"]); close(s.grad, g["g_l1_tv"])
    z = detrand.uniform((6, 1), 54, -3, 3).requires_grad_(True)
    bce = F.binary_cross_entropy_with_logits(z, torch.ones(6, 1))
    bce.backward()
    close(bce.item(), g["bce"]); close(z.grad, g["g_bce"])


def test_gv7_vgg_small():
    g = load("gv7_vgg_small")
    sd = vgg_sd()
    assert sorted(sd.keys()) == sorted(str(k) for k in g["keys"])
    close(sd["sub_mean.weight"], g["sub_w"], 0); close(sd["sub_mean.bias"], g["sub_b"], 0)
    a = detrand.image_batch((2, 3, 32, 32), 31).requires_grad_(True)
    b = detrand.image_batch((2, 3, 32, 32), 32)
    fa, fb = OM.vgg_forward(sd, a, b)
    close(fa.detach(), g["f_sr"], 1e-5); close(fb, g["f_hr"], 1e-5)
    m = F.mse_loss(fa, fb)
    close(m.item(), g["mse"], 1e-5)
    m.backward()
    close(a.grad, g["gin"], 1e-4)


def test_gv8_two_gan_steps():
    g = load("gv8_gan_steps_small")
    cfg = {"depth": 2, "res_scale": 0.1, "learning_rate": 5e-5}
    st = OS.TrainState(gen_sd(16, 2), dis_sd(8), vgg_sd(), cfg)
    for it in range(2):
        lr = detrand.image_batch((4, 3, 8, 8), 100 + it)
        hr = detrand.image_batch((4, 3, 32, 32), 200 + it)
        log = OS.gan_step(st, lr, hr)
        close([log["l1"], log["vgg"], log["g"], log["tv"], log["d"]], g["losses"][it], 2e-5)
    for k, v in st.g.items():
        close(v.detach().reshape(-1)[g["G.idx." + k]], g["G.val." + k], 1e-5)
    for k, v in st.d.items():
        close(v.detach().reshape(-1).float()[g["D.idx." + k]], g["D.val." + k], 1e-5)


def _sampled(leaves, g, prefix, rel):
    n = 0
    for key in [k[len(prefix) + 5:] for k in g.files if k.startswith(prefix + "gidx.")]:
        got = leaves[key].grad.reshape(-1)[g[f"{prefix}gidx.{key}"]]
        close(got, g[f"{prefix}gval.{key}"], 0.0, atol=rel * float(g[f"{prefix}gmax.{key}"]))
        n += 1
    return n


def test_gv4b_discriminator_ps48():
    """Full-size Discriminator (patch_size 48, batch 16, Linear(73728, 1024)) vs the imported reference."""
    g = load("gv4b_discriminator_ps48")
    sd = dis_sd(48)
    leaves = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone())
              for k, v in sd.items()}
    a = detrand.image_batch((16, 3, 192, 192), 21)
    b = detrand.image_batch((16, 3, 192, 192), 22).requires_grad_(True)
    o1 = OM.discriminator_forward(leaves, a)
    o2 = OM.discriminator_forward(leaves, b)
    close(o1.detach(), g["o1"], 1e-5); close(o2.detach(), g["o2"], 1e-5)
    l = F.binary_cross_entropy_with_logits(o1 - o2, torch.ones(16, 1))
    close(l.item(), g["loss"], 1e-6)
    l.backward()
    close(b.grad.reshape(-1)[g["gin_idx"]], g["gin_val"], 0.0, atol=1e-4 * float(g["gin_max"]))
    assert _sampled(leaves, g, "", 1e-4) == 28
    for k, v in leaves.items():
        if "running" in k:
            close(v, g["buf." + k], 1e-5)


def test_gv7b_vgg_192():
    g = load("gv7b_vgg_192")
    sd = vgg_sd()
    a = detrand.image_batch((2, 3, 192, 192), 31).requires_grad_(True)
    b = detrand.image_batch((2, 3, 192, 192), 32)
    fa, fb = OM.vgg_forward(sd, a, b)
    close(fa.detach().reshape(-1)[g["f_idx"]], g["f_sr"], 1e-5); close(fb.reshape(-1)[g["f_idx"]], g["f_hr"], 1e-5)
    m = F.mse_loss(fa, fb)
    close(m.item(), g["mse"], 1e-5)
    m.backward()
    close(a.grad.reshape(-1)[g["gin_idx"]], g["gin_val"], 0.0, atol=1e-4 * float(g["gin_max"]))


def test_gv8b_full_gan_step():
    """The oracle's whole GAN step at the BENCHMARKED size (batch 16, 256 ch x 32 blocks, ps 48) against the step made with
    the reference's own modules: losses, sampled gradients of all 170 parameter tensors, post-Adam parameters.  (Also covers
    GV2b - the full generator forward/backward at batch 16 - which the GPU tests use directly.)  ~1 minute on 8 cores."""
    g = load("gv8b_gan_step_full")
    cfg = {"depth": 32, "res_scale": 0.1, "learning_rate": 5e-5}
    st = OS.TrainState(gen_sd(256, 32), dis_sd(48), vgg_sd(), cfg)
    lr = detrand.image_batch((16, 3, 48, 48), 100)
    hr = detrand.image_batch((16, 3, 192, 192), 200)
    log = OS.gan_step(st, lr, hr)
    close([log["l1"], log["vgg"], log["g"], log["tv"], log["d"]], g["losses"], 2e-5)
    assert _sampled(st.g, g, "G.", 2e-5) == 142
    assert _sampled({k: v for k, v in st.d.items() if v.requires_grad or v.grad is not None}, g, "D.", 2e-5) == 28
    for k, v in st.g.items():
        close(v.detach().reshape(-1)[g["G.idx." + k]], g["G.val." + k], 1e-5)
    for k, v in st.d.items():
        close(v.detach().reshape(-1).float()[g["D.idx." + k]], g["D.val." + k], 1e-5)


@pytest.mark.parametrize("epoch,expect", [(1, 5e-5), (120, 5e-5), (121, 2.5e-5), (240, 2.5e-5), (241, 1.25e-5)])
def test_step_lr_schedule(epoch, expect):
    """Under the reference's pinned torch 0.4 (README.md:22) `_LRScheduler.__init__` leaves last_epoch = -1, so the
    `scheduler.step()` at the start of epoch e (train.py:156,185-186) sets last_epoch = e-1: the first halving is epoch
    lr_step + 1 = 121."""
    assert OS.step_lr(5e-5, epoch, 120) == pytest.approx(expect)


def test_gv11_gradient_penalty_step():
    """The oracle's gradient-penalty branch (reference train.py:216-226, double backward through D) against the reference's own
    modules: D loss, penalty, dD/dx at the interpolate, D's gradients and post-Adam parameters."""
    g = load("gv11_gradient_penalty_small")
    cfg = {"depth": 2, "res_scale": 0.1, "learning_rate": 5e-5, "GP": True}
    st = OS.TrainState(gen_sd(16, 2), dis_sd(8), vgg_sd(), cfg)
    lr = detrand.image_batch((4, 3, 8, 8), 100)
    hr = detrand.image_batch((4, 3, 32, 32), 200)
    u = detrand.uniform((4, 1, 1, 1), 77, 0.0, 1.0)
    log = OS.gan_step(st, lr, hr, gp_u=u)
    close(log["gp"], g["gp"], 2e-5); close(log["d"], g["total"], 2e-5)
    n = 0
    for k, v in st.d.items():
        if ("gidx." + k) in g.files:
            close(v.grad.reshape(-1)[g["gidx." + k]], g["gval." + k], 0.0, atol=2e-5 * float(g["gmax." + k]) + 1e-12)
            n += 1
    assert n == 28
    for k, v in st.d.items():
        if "num_batches" in k:
            assert int(g["pval." + k][0]) == 3 and int(v) == 5    # hr, sr, x_both in the D phase (golden) + sr, hr in the G phase
        elif "running" in k:
            continue                                              # (the golden was taken before the G phase's two D forwards)
        else:
            close(v.detach().reshape(-1)[g["pidx." + k]], g["pval." + k], 1e-5)


def test_rasgan_extension_closed_form_gradient():
    """RaSGAN is NOT in the reference (train.py:210-213 implements SGAN / RSGAN); the oracle restates Jolicoeur-Martineau's
    published form.  Its discriminator-side gradient against the closed form:
    dL/dr_i = [ (s(r_i - mf) - 1) - mean_j s(f_j - mr) ] / (2B),  dL/df_i = [ s(f_i - mr) - mean_j (s(r_j - mf) - 1) ] / (2B),
    s = sigmoid, mr / mf the batch means; and the product's loss functions give the same numbers on the CPU."""
    import torch.nn.functional as F
    from pesr_amd.model.focal_loss import FocalLoss
    from pesr_amd.step import rasgan_d_loss, rasgan_g_loss
    gen = torch.Generator().manual_seed(9)
    B = 7
    r = (torch.randn(B, 1, generator=gen, dtype=torch.float64) * 3).requires_grad_(True)
    f = (torch.randn(B, 1, generator=gen, dtype=torch.float64) * 3).requires_grad_(True)
    ones, zeros = torch.ones(B, 1, dtype=torch.float64), torch.zeros(B, 1, dtype=torch.float64)
    ld = 0.5 * (F.binary_cross_entropy_with_logits(r - f.mean(), ones) + F.binary_cross_entropy_with_logits(f - r.mean(), zeros))
    ld.backward()
    sr, sf = torch.sigmoid(r - f.mean()).detach(), torch.sigmoid(f - r.mean()).detach()
    assert torch.allclose(r.grad, ((sr - 1) - sf.mean()) / (2 * B), rtol=1e-12, atol=1e-15)
    assert torch.allclose(f.grad, (sf - (sr - 1).mean()) / (2 * B), rtol=1e-12, atol=1e-15)
    r2, f2 = r.detach().clone().requires_grad_(True), f.detach().clone().requires_grad_(True)
    l2 = rasgan_d_loss(r2, f2, ones, zeros)
    l2.backward()
    assert float(l2) == pytest.approx(float(ld), rel=1e-12) and torch.allclose(r2.grad, r.grad) and torch.allclose(f2.grad, f.grad)
    # generator side with the focal loss: product module (closed-form backward) vs the oracle's composite
    r3, f3 = r.detach().clone().requires_grad_(True), f.detach().clone().requires_grad_(True)
    lg = rasgan_g_loss(r3, f3, ones, zeros, FocalLoss(1.0))
    lg.backward()
    r4, f4 = r.detach().clone().requires_grad_(True), f.detach().clone().requires_grad_(True)
    lo = 0.5 * (OS.focal_loss(r4 - f4.mean(), zeros, 1.0) + OS.focal_loss(f4 - r4.mean(), ones, 1.0))
    lo.backward()
    assert float(lg) == pytest.approx(float(lo), rel=1e-12)
    assert torch.allclose(r3.grad, r4.grad, rtol=1e-10, atol=1e-14) and torch.allclose(f3.grad, f4.grad, rtol=1e-10, atol=1e-14)


def test_spectral_norm_restatement_vs_torch():
    """reference model/basic.py:25 wraps the Discriminator's convs in an undefined `spectral_norm` (NameError); the
    evident intent is torch.nn.utils.spectral_norm.  The oracle's restatement (oracle/model.py spectral_normalize) against
    torch's own implementation on a conv: two training forwards before one backward (the GAN pattern loss = D(real) - D(fake)),
    then an eval forward - outputs, the gradient of weight_orig, and the u / v buffers after every call."""
    torch.manual_seed(3)
    conv = torch.nn.utils.spectral_norm(torch.nn.Conv2d(5, 7, 3, padding=1, bias=False))
    w = conv.weight_orig.detach().clone().requires_grad_(True)
    u, v = conv.weight_u.clone(), conv.weight_v.clone()
    xa, xb = torch.randn(2, 5, 6, 6), torch.randn(2, 5, 6, 6)
    ya, yb = conv(xa), conv(xb)
    (ya.sum() - 2.0 * yb.square().sum()).backward()
    oa = F.conv2d(xa, OM.spectral_normalize(w, u, v, True), padding=1)
    ob = F.conv2d(xb, OM.spectral_normalize(w, u, v, True), padding=1)
    (oa.sum() - 2.0 * ob.square().sum()).backward()
    assert torch.allclose(oa, ya, rtol=1e-6, atol=1e-7) and torch.allclose(ob, yb, rtol=1e-6, atol=1e-7)
    assert torch.allclose(u, conv.weight_u, rtol=1e-6, atol=1e-8) and torch.allclose(v, conv.weight_v, rtol=1e-6, atol=1e-8)
    assert torch.allclose(w.grad, conv.weight_orig.grad, rtol=1e-5, atol=1e-7)
    conv.eval()
    ye = conv(xa)
    oe = F.conv2d(xa, OM.spectral_normalize(w, u, v, False), padding=1)
    assert torch.allclose(oe, ye, rtol=1e-6, atol=1e-7) and torch.allclose(u, conv.weight_u)
    # closed form of the gradient the HIP kernel implements: dW = (G - <G, W_hat> u v^T) / sigma
    w2 = w.detach().clone().requires_grad_(True)
    u2, v2 = u.clone(), v.clone()
    wh = OM.spectral_normalize(w2, u2, v2, True)
    G = torch.randn_like(wh)
    wh.backward(G)
    sigma = torch.dot(u2, torch.mv(w2.detach().reshape(7, -1), v2))
    closed = (G - (G * wh.detach()).sum() * torch.outer(u2, v2).view_as(G)) / sigma
    assert torch.allclose(w2.grad, closed, rtol=1e-5, atol=1e-7)
