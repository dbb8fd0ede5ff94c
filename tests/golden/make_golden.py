"""Generate the golden fixtures under tests/golden/ by IMPORTING THE REFERENCE (build container only).

Run:  python tests/golden/make_golden.py
Needs /root/reference (read-only) - never runs on the GPU box.  `torchvision` is absent here, so a stub
providing torchvision.models.vgg19().features with the public cfg-"E" topology is injected before the
import; VGG weight VALUES are therefore build-generated ("parity unpinned" for the
pretrained values, pinned for topology / slicing / MeanShift constants).
Only data (inputs' seeds, sampled outputs) is written - no reference source text.
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import detrand  # noqa: E402
from oracle import model as OM  # noqa: E402  (only for shape tables / cfg list)


def import_reference():
    tv, tvm = types.ModuleType("torchvision"), types.ModuleType("torchvision.models")

    class _V(nn.Module):
        def __init__(self):
            super().__init__()
            layers, c = [], 3
            for v in OM.VGG_CFG_E:
                if v == "M":
                    layers.append(nn.MaxPool2d(2, 2))
                else:
                    layers += [nn.Conv2d(c, v, 3, padding=1), nn.ReLU(True)]
                    c = v
            self.features = nn.Sequential(*layers)

    tvm.vgg19 = lambda pretrained=False, **k: _V()
    tv.models = tvm
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.models"] = tvm
    sys.path.insert(0, "/root/reference")
    import model as ref_model  # noqa
    import utils as ref_utils  # noqa
    return ref_model, ref_utils


def sample_idx(n, k, seed):
    if n <= k:
        return np.arange(n)
    u = detrand.uniform01(k, seed)
    return np.unique((u.astype(np.float64) * n).astype(np.int64))


def load_det(module, seed, scheme="default", skip=("sub_mean", "add_mean")):
    sd = module.state_dict()
    shapes = {k: tuple(v.shape) for k, v in sd.items() if not k.startswith(skip)}
    vals = detrand.fill_state_dict(shapes, seed, scheme)
    sd.update(vals)
    module.load_state_dict(sd)
    return module


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        out[k] = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, {k: out[k].shape for k in out})


def main():
    torch.set_num_threads(8)
    R, U = import_reference()

    # ---- GV1: small generator, full output + all param grads under L1 ------------------------------
    opt = {"num_channels": 16, "depth": 2, "res_scale": 0.1}
    G = load_det(R.Generator(opt), seed=0)
    lr = detrand.image_batch((2, 3, 12, 12), 1234)
    hr = detrand.image_batch((2, 3, 48, 48), 1235)
    sr = G(lr)
    loss = F.l1_loss(sr, hr)
    loss.backward()
    arrs = {"sr": sr, "loss": loss}
    for k, p in G.named_parameters():
        arrs["grad." + k] = p.grad
    arrs["keys"] = np.array(list(G.state_dict().keys()))
    save("gv1_generator_small", **arrs)

    # ---- GV2: full generator (256 ch, 32 blocks) at [2,3,48,48]: sampled output + sampled grads ------
    opt = {"num_channels": 256, "depth": 32, "res_scale": 0.1}
    G = load_det(R.Generator(opt), seed=0)
    lr = detrand.image_batch((2, 3, 48, 48), 1234)
    hr = detrand.image_batch((2, 3, 192, 192), 1235)
    sr = G(lr)
    loss = F.l1_loss(sr, hr)
    loss.backward()
    idx = sample_idx(sr.numel(), 4096, 99)
    arrs = {"sr_idx": idx, "sr_val": sr.detach().reshape(-1)[idx], "sr_sum": sr.sum(), "sr_abs_sum": sr.abs().sum(),
            "loss": loss}
    for k in ["sub_mean.weight", "sub_mean.bias", "embed.weight", "embed.bias", "body.0.body.0.weight",
              "body.0.body.0.bias", "body.31.body.2.weight", "body.32.weight", "upsample.0.weight",
              "upsample.0.bias", "upsample.2.weight", "upsample.4.weight", "upsample.4.bias", "add_mean.weight",
              "add_mean.bias"]:
        g = dict(G.named_parameters())[k].grad.reshape(-1)
        gi = sample_idx(g.numel(), 2048, 7)
        arrs["gidx." + k] = gi
        arrs["gval." + k] = g[gi]
        arrs["gmax." + k] = g.abs().max()
    save("gv2_generator_full", **arrs)

    # ---- GV3: PixelShuffle(2) on arange, forward and backward (bit-exact) ---------------------------
    x = torch.arange(2 * 16 * 3 * 5, dtype=torch.float32).reshape(2, 16, 3, 5).requires_grad_(True)
    y = nn.PixelShuffle(2)(x)
    gy = torch.arange(y.numel(), dtype=torch.float32).reshape(y.shape) * 0.5
    y.backward(gy)
    save("gv3_pixel_shuffle", y=y, gx=x.grad)

    # ---- GV4: discriminator at patch_size 8 (32x32 in), train-mode BN, two consecutive calls ---------
    D = load_det(R.Discriminator({"patch_size": 8, "spectral_norm": False}), seed=1)
    a = detrand.image_batch((4, 3, 32, 32), 21)
    b = detrand.image_batch((4, 3, 32, 32), 22).requires_grad_(True)
    o1 = D(a)
    o2 = D(b)
    l = F.binary_cross_entropy_with_logits(o1 - o2, torch.ones(4, 1))
    l.backward()
    arrs = {"o1": o1, "o2": o2, "loss": l, "gin": b.grad}
    for k, p in D.named_parameters():
        g = p.grad.reshape(-1)
        gi = sample_idx(g.numel(), 2048, 5)
        arrs["gidx." + k] = gi
        arrs["gval." + k] = g[gi]
        arrs["gmax." + k] = g.abs().max()
    for k, v in D.state_dict().items():
        if "running" in k or "num_batches" in k:
            arrs["buf." + k] = v
    save("gv4_discriminator_small", **arrs)

    # ---- GV5: FocalLoss forward from the reference on a grid -----------------------------------------
    xs = torch.linspace(-8, 8, 33).reshape(-1, 1)
    arrs = {"x": xs}
    with torch.no_grad():
        for gamma in (0, 1, 2):
            for t in (0, 1):
                tt = torch.full_like(xs, float(t))
                vals = torch.stack([R.FocalLoss(gamma)(xs[i:i + 1], tt[i:i + 1]) for i in range(xs.size(0))])
                arrs[f"f_g{gamma}_t{t}"] = vals
                arrs[f"mean_g{gamma}_t{t}"] = R.FocalLoss(gamma)(xs, tt)
    save("gv5_focal", **arrs)

    # ---- GV7: VGG topology with generated weights -----------------------------------------------------
    V = R.VGG()
    load_det(V, seed=2, scheme="vgg", skip=("sub_mean",))
    a = detrand.image_batch((2, 3, 32, 32), 31).requires_grad_(True)
    b = detrand.image_batch((2, 3, 32, 32), 32)
    fa, fb = V(a, b)
    m = F.mse_loss(fa, fb)
    m.backward()
    save("gv7_vgg_small", f_sr=fa, f_hr=fb, mse=m, gin=a.grad, keys=np.array(list(V.state_dict().keys())),
         sub_w=V.sub_mean.weight, sub_b=V.sub_mean.bias)

    # ---- GV8: two consecutive GAN steps at the small config with the reference's modules --------------
    opt = {"patch_size": 8, "num_channels": 16, "depth": 2, "res_scale": 0.1, "spectral_norm": False}
    G = load_det(R.Generator(opt), seed=0)
    D = load_det(R.Discriminator(opt), seed=1)
    V = load_det(R.VGG(), seed=2, scheme="vgg", skip=("sub_mean",))
    oG = torch.optim.Adam([p for p in G.parameters() if p.requires_grad], betas=(0.9, 0.999), lr=5e-5)
    oD = torch.optim.Adam(D.parameters(), betas=(0.9, 0.999), lr=5e-5)
    ones = torch.ones(4, 1)
    logs = []
    for it in range(2):
        lr = detrand.image_batch((4, 3, 8, 8), 100 + it)
        hr = detrand.image_batch((4, 3, 32, 32), 200 + it)
        for p in D.parameters():
            p.requires_grad = True
        oD.zero_grad()
        pr = D(hr)
        sr = G(lr)
        pf = D(sr.detach())
        dl = F.binary_cross_entropy_with_logits(pr - pf, ones)
        dl.backward()
        oD.step()
        for p in D.parameters():
            p.requires_grad = False
        oG.zero_grad()
        pf = D(sr)
        pr = D(hr)
        l1 = F.l1_loss(sr, hr) * 0.0
        fs, fh = V(sr, hr)
        vg = F.mse_loss(fs, fh) * 50.0
        d = sr
        tv = (torch.sum(torch.abs(d[:, :, :, :-1] - d[:, :, :, 1:])) + torch.sum(torch.abs(d[:, :, :-1, :] - d[:, :, 1:, :]))) * 1e-6
        z = pf - pr
        # FocalLoss (gamma 1) value from the reference module; gradient through the torch-0.4-semantics composite (Q4)
        with torch.no_grad():
            ref_val = R.FocalLoss(1)(z, ones)
        p_ = torch.sigmoid(z)
        w_ = (1 - (p_ * ones + (1 - p_) * (1 - ones))).pow(1)
        gl = (w_ * F.binary_cross_entropy_with_logits(z, ones, reduction="none")).mean()
        assert abs(gl.item() - ref_val.item()) <= 1e-6 * max(1.0, abs(ref_val.item()))
        tot = l1 + vg + gl + tv
        tot.backward()
        oG.step()
        logs.append([l1.item(), vg.item(), gl.item(), tv.item(), dl.item()])
    arrs = {"losses": np.array(logs, dtype=np.float64)}
    for k, v in G.state_dict().items():
        v = v.reshape(-1)
        gi = sample_idx(v.numel(), 512, 3)
        arrs["G.idx." + k] = gi
        arrs["G.val." + k] = v[gi]
    for k, v in D.state_dict().items():
        v = v.reshape(-1).float()
        gi = sample_idx(v.numel(), 512, 3)
        arrs["D.idx." + k] = gi
        arrs["D.val." + k] = v[gi]
    save("gv8_gan_steps_small", **arrs)

    # ---- GV11: one GAN step WITH gradient penalty (train.py:216-226) at the small config, reference modules ----------
    opt = {"patch_size": 8, "num_channels": 16, "depth": 2, "res_scale": 0.1, "spectral_norm": False}
    G = load_det(R.Generator(opt), seed=0)
    D = load_det(R.Discriminator(opt), seed=1)
    oD = torch.optim.Adam(D.parameters(), betas=(0.9, 0.999), lr=5e-5)
    lr = detrand.image_batch((4, 3, 8, 8), 100)
    hr = detrand.image_batch((4, 3, 32, 32), 200)
    u = detrand.uniform((4, 1, 1, 1), 77, 0.0, 1.0)
    ones = torch.ones(4, 1)
    oD.zero_grad()
    pr = D(hr)
    sr = G(lr)
    pf = D(sr.detach())
    dl = F.binary_cross_entropy_with_logits(pr - pf, ones)
    x_both = (hr * u + sr * (1 - u)).detach().requires_grad_(True)          # Variable(x_both, requires_grad=True): a new leaf
    grad = torch.autograd.grad(outputs=D(x_both), inputs=x_both, grad_outputs=torch.ones(4, 1), retain_graph=True,
                               create_graph=True, only_inputs=True)[0]
    gp = 10 * ((grad.norm(2, 1).norm(2, 1).norm(2, 1) - 1) ** 2).mean()
    tot = dl + gp
    tot.backward()
    gi = sample_idx(grad.numel(), 2048, 9)
    arrs = {"d_loss": dl, "gp": gp, "total": tot, "gx_idx": gi, "gx_val": grad.detach().reshape(-1)[gi], "gx_max": grad.abs().max()}
    for k, p in D.named_parameters():
        g_ = p.grad.reshape(-1)
        gi = sample_idx(g_.numel(), 1024, 5)
        arrs["gidx." + k], arrs["gval." + k], arrs["gmax." + k] = gi, g_[gi], g_.abs().max()
    oD.step()
    for k, v in D.state_dict().items():
        v = v.reshape(-1).float()
        gi = sample_idx(v.numel(), 256, 3)
        arrs["pidx." + k], arrs["pval." + k] = gi, v[gi]
    save("gv11_gradient_penalty_small", **arrs)

    # ---- GV9: utils known answers ------------------------------------------------------------------------
    a = detrand.image_batch((1, 3, 16, 20), 41)
    b = (a + detrand.uniform((1, 3, 16, 20), 42, -20, 20))
    psnr = U.compute_PSNR(a.clone(), b.clone())
    [img] = U.tensors_to_imgs([b.clone() * 1.1 - 5])
    y = U.rgb2y(img.astype(np.float64))
    save("gv9_utils", psnr=np.float64(psnr), img=img, y=y)

    # ---- GV6: plain loss values / grads (F.* calls of train.py:131-140 on reference-shaped data) --------
    s = (detrand.image_batch((2, 3, 10, 12), 51) + detrand.uniform((2, 3, 10, 12), 52, -0.5, 0.5)).requires_grad_(True)
    h = detrand.image_batch((2, 3, 10, 12), 53)
    l1 = nn.L1Loss()(s, h)
    tv = torch.sum(torch.abs(s[:, :, :, :-1] - s[:, :, :, 1:])) + torch.sum(torch.abs(s[:, :, :-1, :] - s[:, :, 1:, :]))
    (l1 + tv).backward()
    z = detrand.uniform((6, 1), 54, -3, 3).requires_grad_(True)
    bce = nn.BCEWithLogitsLoss()(z, torch.ones(6, 1))
    bce.backward()
    save("gv6_losses", l1=l1, tv=tv, g_l1_tv=s.grad, bce=bce, g_bce=z.grad)


if __name__ == "__main__":
    main()
