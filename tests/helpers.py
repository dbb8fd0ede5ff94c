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


def adam_close(got, ref, lr, steps, what=""):
    """Post-Adam parameter comparison.  Adam's first steps move every element by ~lr in the direction of sign(g), so an
    element whose gradient is at the fp32 noise level can legitimately differ by up to 2*lr per step (the reference's
    own fp32-vs-fp64 runs differ by 2.0*lr after two steps on some elements).  What parity
    can and does assert: the worst element stays within that sign-flip bound, at most a few elements (3 % / 3) are
    "flipped" (off by > 0.5 lr), and the MEAN error of all others stays far below lr (a tensor that is not trained, or
    trained with a wrong gradient, is off by >= lr on average)."""
    a = np.asarray(got.detach().cpu() if torch.is_tensor(got) else got, dtype=np.float64).reshape(-1)
    b = np.asarray(ref.detach().cpu() if torch.is_tensor(ref) else ref, dtype=np.float64).reshape(-1)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a - b)
    scale = np.abs(b).max() + 1e-30
    assert err.max() <= 2.2 * lr * steps + 2e-5 * scale, f"{what}: max err {err.max():.3e} = {err.max() / lr:.2f} lr"
    flipped = err > 0.5 * lr                       # elements whose near-zero gradient changed sign somewhere
    assert flipped.sum() <= max(3, 0.03 * err.size), f"{what}: {int(flipped.sum())} of {err.size} elements off by > 0.5 lr"
    rest = err[~flipped]
    if rest.size:
        assert rest.mean() <= 0.06 * lr + 2e-6 * scale, f"{what}: mean err of the rest {rest.mean():.3e} = {rest.mean() / lr:.3f} lr"
