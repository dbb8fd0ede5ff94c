"""CPU restatement of the reference's losses and train steps - TEST INFRASTRUCTURE.

Follows reference train.py:123-140 (optimizers, loss closures), :164-173 (pretrain step) and :194-259
(GAN step) and model/focal_loss.py:9-13, on the functional networks of oracle/model.py.
"""
import math

import torch
import torch.nn.functional as F

from . import model as M


# ------------------------------------------------------------------------------------------------
# losses
# ------------------------------------------------------------------------------------------------
def focal_loss(x, t, gamma):
    """model/focal_loss.py:9-13.  The reference passes a grad-requiring `w` as the BCE `weight`; under its
    pinned torch 0.4 BCE-with-logits was a Python composite and gradient flowed through BOTH w and the BCE
    term.  This composite restates exactly that; the forward value equals the reference's."""
    p = torch.sigmoid(x)
    pt = p * t + (1 - p) * (1 - t)
    w = (1 - pt).pow(gamma)
    return (w * F.binary_cross_entropy_with_logits(x, t, reduction="none")).mean()


def focal_loss_grad_closed_form(x, t, gamma):
    """dL/dx of focal_loss: [dw/dx * bce + w * (p - t)] / N with dw/dx = -gamma (1-pt)^(gamma-1) (2t-1) p (1-p)."""
    p = torch.sigmoid(x)
    pt = p * t + (1 - p) * (1 - t)
    bce = F.binary_cross_entropy_with_logits(x, t, reduction="none")
    w = (1 - pt).pow(gamma)
    if gamma == 0:
        dw = torch.zeros_like(x)
    else:
        dw = -gamma * (1 - pt).pow(gamma - 1) * (2 * t - 1) * p * (1 - p)
    return (dw * bce + w * (p - t)) / x.numel()


def tv_loss(y):
    """train.py:137-140: SUM (not mean) of absolute horizontal and vertical differences."""
    return torch.sum(torch.abs(y[:, :, :, :-1] - y[:, :, :, 1:])) + torch.sum(torch.abs(y[:, :, :-1, :] - y[:, :, 1:, :]))


# ------------------------------------------------------------------------------------------------
# train state
# ------------------------------------------------------------------------------------------------
def is_buffer(name):
    """state_dict entries that are buffers, not parameters: BatchNorm running statistics / counters and spectral norm's u, v."""
    return "running" in name or name.endswith(("num_batches_tracked", "weight_u", "weight_v"))


class TrainState:
    """Parameters as leaf tensors + torch.optim.Adam exactly as train.py:123-128 builds them."""

    def __init__(self, g_sd, d_sd, vgg_sd, cfg):
        self.cfg = dict(cfg)
        self.g = {k: v.clone().requires_grad_(True) for k, v in g_sd.items()}            # all of G trains (Q1)
        self.d = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and not is_buffer(k) else v.clone())
                  for k, v in (d_sd or {}).items()}
        self.vgg = {k: v.clone() for k, v in (vgg_sd or {}).items()}
        lr = cfg.get("learning_rate", 5e-5)
        self.optim_g = torch.optim.Adam(list(self.g.values()), betas=(0.9, 0.999), lr=lr)
        d_params = [v for v in self.d.values() if v.requires_grad]
        self.optim_d = torch.optim.Adam(d_params, betas=(0.9, 0.999), lr=lr) if d_params else None

    def G(self, x):
        return M.generator_forward(self.g, x, self.cfg["depth"], self.cfg["res_scale"])

    def D(self, x):
        """Discriminator forward.  cfg["dp_replicas"] = R > 1 restates what nn.DataParallel(D) (train.py:116) computes on R
        devices: the batch is scattered into R contiguous shards and every replica normalises with ITS shard's BatchNorm
        statistics (no SyncBN); only replica 0's running-stat update survives (its buffers are the module's own)."""
        R = int(self.cfg.get("dp_replicas", 1))
        if R <= 1:
            return M.discriminator_forward(self.d, x, update_running_stats=True)
        outs = [M.discriminator_forward(self.d, xs, update_running_stats=(i == 0)) for i, xs in enumerate(x.chunk(R, 0))]
        return torch.cat(outs, 0)


def pretrain_step(st: TrainState, lr, hr):
    """train.py:164-173."""
    sr = st.G(lr)
    st.optim_g.zero_grad()
    loss = F.l1_loss(sr, hr)
    loss.backward()
    st.optim_g.step()
    return {"l1": loss.item()}


def gradient_penalty(D, hr, sr, u):
    """train.py:216-226: 10 * mean((||dD(x_both)/dx_both||_2 - 1)^2) at x_both = hr*u + sr*(1-u), one u per sample; x_both is a
    NEW leaf (the reference wraps it in a fresh Variable), so nothing flows back into G.  `u` ([B,1,1,1], uniform in [0,1) in the
    reference) is an argument here so that runs can be reproduced.  The extra D forward updates the BatchNorm running stats."""
    x_both = (hr * u + sr.detach() * (1 - u)).detach().requires_grad_(True)
    out = D(x_both)
    grad = torch.autograd.grad(outputs=out, inputs=x_both, grad_outputs=torch.ones_like(out), retain_graph=True, create_graph=True,
                               only_inputs=True)[0]
    return 10 * ((grad.norm(2, 1).norm(2, 1).norm(2, 1) - 1) ** 2).mean()


def gan_step(st: TrainState, lr, hr, gp_u=None):
    """train.py:194-259 with the defaults' branches kept selectable (gan_type, focal_loss, GP).  cfg["GP"] = True adds the
    gradient penalty (train.py:216-226) with the interpolation weights gp_u."""
    c = st.cfg
    B = lr.size(0)
    target_real = torch.ones(B, 1)
    target_fake = torch.zeros(B, 1)
    d_leaves = [v for k, v in st.d.items() if v.is_floating_point() and not is_buffer(k)]

    # ---- discriminator phase (:202-229)
    for p in d_leaves:
        p.requires_grad_(True)
    st.optim_d.zero_grad()
    pred_real = st.D(hr)
    sr = st.G(lr)
    pred_fake = st.D(sr.detach())
    if c.get("gan_type", "RSGAN") == "SGAN":
        d_loss = F.binary_cross_entropy_with_logits(pred_real, target_real) + \
            F.binary_cross_entropy_with_logits(pred_fake, target_fake)
    elif c.get("gan_type") == "RaSGAN":
        # NOT in the reference (train.py:210-213 has SGAN / RSGAN): Jolicoeur-Martineau 2018, relativistic average standard GAN,
        # as in the paper's published code - restated here as the checker of the product's extension; batch means are over the
        # whole (global) batch
        d_loss = 0.5 * (F.binary_cross_entropy_with_logits(pred_real - pred_fake.mean(), target_real) +
                        F.binary_cross_entropy_with_logits(pred_fake - pred_real.mean(), target_fake))
    else:
        d_loss = F.binary_cross_entropy_with_logits(pred_real - pred_fake, target_real)
    gp = None
    if c.get("GP", False):
        gp = gradient_penalty(st.D, hr, sr, gp_u)
        d_loss = d_loss + gp
    d_loss.backward()
    st.optim_d.step()

    # ---- generator phase (:234-259)
    for p in d_leaves:
        p.requires_grad_(False)
    st.optim_g.zero_grad()
    pred_fake = st.D(sr)
    pred_real = st.D(hr)
    l1 = F.l1_loss(sr, hr) * c.get("alpha_l1", 0.0)
    f_sr, f_hr = M.vgg_forward(st.vgg, sr, hr)
    vgg = F.mse_loss(f_sr, f_hr) * c.get("alpha_vgg", 50.0)
    tv = tv_loss(sr) * c.get("alpha_tv", 1e-6)
    gamma = c.get("fl_gamma", 1.0)
    use_focal = c.get("focal_loss", True)
    lf = (lambda z, t: focal_loss(z, t, gamma)) if use_focal else F.binary_cross_entropy_with_logits
    if c.get("gan_type") == "RaSGAN":       # extension, see the discriminator phase
        g_loss = 0.5 * (lf(pred_real - pred_fake.mean(), target_fake) + lf(pred_fake - pred_real.mean(), target_real))
    else:
        z = pred_fake if c.get("gan_type", "RSGAN") == "SGAN" else pred_fake - pred_real
        g_loss = lf(z, target_real)
    g_loss = g_loss * c.get("alpha_gan", 1.0)
    total = l1 + vgg + g_loss + tv
    total.backward()
    st.optim_g.step()
    out = {"l1": l1.item(), "vgg": vgg.item(), "g": g_loss.item(), "tv": tv.item(), "d": d_loss.item()}
    if gp is not None:
        out["gp"] = gp.item()
    return out


def step_lr(base_lr, epoch, lr_step, gamma=0.5):
    """StepLR stepped at EPOCH START (train.py:156,185-186) under the reference's pinned torch 0.4 (README.md:22), whose
    `_LRScheduler.__init__` runs step(0) and then resets last_epoch to -1: the step() at the start of epoch e (1-based)
    gives last_epoch = e-1, so epoch e trains at base*gamma^floor((e-1)/lr_step) - epochs 1..120 at lr, 121..240 at lr/2.
    (Run unchanged on torch >= 1.1 the same script halves one epoch earlier, because the constructor already counts a step.)"""
    return base_lr * math.pow(gamma, (epoch - 1) // lr_step)
