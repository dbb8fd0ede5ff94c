"""Full-size golden fixtures (the benchmarked shapes) made by IMPORTING THE REFERENCE - build container only.

Run:  python tests/golden/make_golden_full.py            (about 3 minutes on 8 cores, ~12 GB RSS)
SURVEY.md 8(c) asks for the reference's outputs at the sizes the bench runs, not only at toy sizes:
  GV2b  Generator 256 ch x 32 blocks at [16,3,48,48]   (reference model/pesr.py:28-38): sampled sr, sums, sampled grads
  GV4b  Discriminator at patch_size 48, batch 16        (model/pesr.py:50,69-81): two train-mode calls, sampled grads
  GV7b  VGG features[:35] at 192x192, batch 2           (model/vgg.py:8-28): sampled features + input gradient
  GV8b  ONE full-size GAN step (B=16, 256x32, ps=48) with the reference's modules (train.py:194-259):
        the 5 losses, sampled gradients of every G and D tensor, sampled post-Adam parameters
Only data is written (sample indices, values, maxima) - no reference source text.
"""
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (reference import stub, deterministic fills, save)
from oracle import detrand  # noqa: E402

G_GRAD_KEYS = ["sub_mean.weight", "sub_mean.bias", "embed.weight", "embed.bias", "body.0.body.0.weight",
               "body.0.body.0.bias", "body.15.body.2.weight", "body.31.body.2.weight", "body.31.body.2.bias", "body.32.weight",
               "upsample.0.weight", "upsample.0.bias", "upsample.2.weight", "upsample.2.bias", "upsample.4.weight",
               "upsample.4.bias", "add_mean.weight", "add_mean.bias"]


def sampled_grads(arrs, prefix, named_params, keys=None, k=2048, seed=7):
    for name, p in named_params:
        if keys is not None and name not in keys:
            continue
        g = p.grad.reshape(-1)
        gi = MG.sample_idx(g.numel(), k, seed)
        arrs[f"{prefix}gidx.{name}"] = gi
        arrs[f"{prefix}gval.{name}"] = g[gi]
        arrs[f"{prefix}gmax.{name}"] = g.abs().max()


def main():
    torch.set_num_threads(8)
    R, _ = MG.import_reference()
    full = {"patch_size": 48, "num_channels": 256, "depth": 32, "res_scale": 0.1, "spectral_norm": False}

    # ---- GV2b: full generator at the benchmarked batch -------------------------------------------------
    t0 = time.time()
    G = MG.load_det(R.Generator(full), seed=0)
    lr = detrand.image_batch((16, 3, 48, 48), 1234)
    hr = detrand.image_batch((16, 3, 192, 192), 1235)
    sr = G(lr)
    loss = F.l1_loss(sr, hr)
    loss.backward()
    idx = MG.sample_idx(sr.numel(), 8192, 99)
    arrs = {"sr_idx": idx, "sr_val": sr.detach().reshape(-1)[idx], "sr_sum": sr.sum(), "sr_abs_sum": sr.abs().sum(), "loss": loss}
    sampled_grads(arrs, "", G.named_parameters(), G_GRAD_KEYS)
    MG.save("gv2b_generator_full_b16", **arrs)
    print(f"  GV2b {time.time() - t0:.0f} s", flush=True)
    del G, sr, loss

    # ---- GV4b: discriminator at patch_size 48 (192x192 in, Linear(73728, 1024)), two train-mode calls -----
    t0 = time.time()
    D = MG.load_det(R.Discriminator(full), seed=1)
    a = detrand.image_batch((16, 3, 192, 192), 21)
    b = detrand.image_batch((16, 3, 192, 192), 22).requires_grad_(True)
    o1 = D(a)
    o2 = D(b)
    l = F.binary_cross_entropy_with_logits(o1 - o2, torch.ones(16, 1))
    l.backward()
    gi = MG.sample_idx(b.grad.numel(), 8192, 11)
    arrs = {"o1": o1, "o2": o2, "loss": l, "gin_idx": gi, "gin_val": b.grad.reshape(-1)[gi], "gin_max": b.grad.abs().max()}
    sampled_grads(arrs, "", D.named_parameters(), None, 2048, 5)
    for k, v in D.state_dict().items():
        if "running" in k or "num_batches" in k:
            arrs["buf." + k] = v
    MG.save("gv4b_discriminator_ps48", **arrs)
    print(f"  GV4b {time.time() - t0:.0f} s", flush=True)
    del D, o1, o2, l

    # ---- GV7b: VGG at 192x192 --------------------------------------------------------------------------
    t0 = time.time()
    V = R.VGG()
    MG.load_det(V, seed=2, scheme="vgg", skip=("sub_mean",))
    a = detrand.image_batch((2, 3, 192, 192), 31).requires_grad_(True)
    b = detrand.image_batch((2, 3, 192, 192), 32)
    fa, fb = V(a, b)
    m = F.mse_loss(fa, fb)
    m.backward()
    fi = MG.sample_idx(fa.numel(), 8192, 13)
    gi = MG.sample_idx(a.grad.numel(), 8192, 17)
    MG.save("gv7b_vgg_192", f_idx=fi, f_sr=fa.detach().reshape(-1)[fi], f_hr=fb.reshape(-1)[fi], f_sr_abs_sum=fa.abs().sum(),
            mse=m, gin_idx=gi, gin_val=a.grad.reshape(-1)[gi], gin_max=a.grad.abs().max())
    print(f"  GV7b {time.time() - t0:.0f} s", flush=True)
    del V, fa, fb

    # ---- GV8b: one full-size GAN step with the reference's modules (train.py:194-259 restated on CPU tensors) ---
    t0 = time.time()
    G = MG.load_det(R.Generator(full), seed=0)
    D = MG.load_det(R.Discriminator(full), seed=1)
    V = MG.load_det(R.VGG(), seed=2, scheme="vgg", skip=("sub_mean",))
    oG = torch.optim.Adam([p for p in G.parameters() if p.requires_grad], betas=(0.9, 0.999), lr=5e-5)
    oD = torch.optim.Adam(D.parameters(), betas=(0.9, 0.999), lr=5e-5)
    ones = torch.ones(16, 1)
    lr = detrand.image_batch((16, 3, 48, 48), 100)
    hr = detrand.image_batch((16, 3, 192, 192), 200)
    for p in D.parameters():
        p.requires_grad = True
    oD.zero_grad()
    pr = D(hr)
    sr = G(lr)
    pf = D(sr.detach())
    dl = F.binary_cross_entropy_with_logits(pr - pf, ones)
    dl.backward()
    arrs = {}
    sampled_grads(arrs, "D.", D.named_parameters(), None, 1024, 3)      # D's gradients (before its Adam step)
    oD.step()
    for p in D.parameters():
        p.requires_grad = False
    oG.zero_grad()
    pf = D(sr)
    pr = D(hr)
    l1 = F.l1_loss(sr, hr) * 0.0
    fs, fh = V(sr, hr)
    vg = F.mse_loss(fs, fh) * 50.0
    tv = (torch.sum(torch.abs(sr[:, :, :, :-1] - sr[:, :, :, 1:])) + torch.sum(torch.abs(sr[:, :, :-1, :] - sr[:, :, 1:, :]))) * 1e-6
    z = pf - pr
    with torch.no_grad():
        ref_val = R.FocalLoss(1)(z, ones)          # the reference module's forward value
    p_ = torch.sigmoid(z)                          # gradient through the torch-0.4-semantics composite
    w_ = (1 - (p_ * ones + (1 - p_) * (1 - ones))).pow(1)
    gl = (w_ * F.binary_cross_entropy_with_logits(z, ones, reduction="none")).mean()
    assert abs(gl.item() - ref_val.item()) <= 1e-6 * max(1.0, abs(ref_val.item()))
    tot = l1 + vg + gl + tv
    tot.backward()
    sampled_grads(arrs, "G.", G.named_parameters(), None, 512, 3)
    oG.step()
    arrs["losses"] = np.array([l1.item(), vg.item(), gl.item(), tv.item(), dl.item()], dtype=np.float64)
    arrs["pred"] = torch.cat([pf.detach(), pr.detach()], 1)
    for name, net in (("G", G), ("D", D)):
        for k, v in net.state_dict().items():
            v = v.reshape(-1).float()
            gi = MG.sample_idx(v.numel(), 256, 3)
            arrs[f"{name}.idx.{k}"] = gi
            arrs[f"{name}.val.{k}"] = v[gi]
    MG.save("gv8b_gan_step_full", **arrs)
    print(f"  GV8b {time.time() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    main()
