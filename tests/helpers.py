"""Shared test helpers: deterministic state dicts (oracle/detrand) and fixture loading."""
import os

import numpy as np
import torch

from oracle import detrand
from oracle import model as OM

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def gen_sd(C, depth, seed=0):
    shapes = {k: v for k, v in OM.generator_shapes(C, depth).items() if not k.startswith(("sub_mean", "add_mean"))}
    sd = detrand.fill_state_dict(shapes, seed)
    OM.set_meanshift(sd, "G")
    return {k: sd[k] for k in OM.generator_shapes(C, depth)}


def dis_sd(ps, seed=1):
    return detrand.fill_state_dict(OM.discriminator_shapes(ps), seed)


def vgg_sd(seed=2):
    shapes = {k: v for k, v in OM.vgg_shapes().items() if not k.startswith("sub_mean")}
    sd = detrand.fill_state_dict(shapes, seed, "vgg")
    return OM.set_meanshift(sd, "V")


def close(a, b, rtol=1e-6, atol=0.0, what=""):
    a = np.asarray(a.detach().cpu() if torch.is_tensor(a) else a, dtype=np.float64)
    b = np.asarray(b.detach().cpu() if torch.is_tensor(b) else b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = np.abs(b).max() + 1e-30
    err = np.abs(a - b).max()
    assert err <= rtol * scale + atol, f"{what}: max err {err:.3e}, scale {scale:.3e}, rel {err / scale:.3e} > {rtol}"
