"""Build-owned deterministic number generator (counter-based splitmix64 -> uniform fp32).

Used to fill BOTH the reference modules (in the build container, when golden vectors are made) and
the modules under test, so no fixture depends on torch.manual_seed streams.
"""
import math
import re

import numpy as np
import torch

_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(z):
    z = (z + np.uint64(0x9E3779B97F4A7C15)) & _M
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M
    return z ^ (z >> np.uint64(31))


def uniform01(n: int, seed: int) -> np.ndarray:
    """n fp32 values in [0, 1): 24 random bits each, a pure function of (seed, index)."""
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + _splitmix64(np.uint64(seed & 0xFFFFFFFFFFFFFFFF)) * np.uint64(2)
        z = _splitmix64(idx)
    return ((z >> np.uint64(40)).astype(np.float64) / float(1 << 24)).astype(np.float32)


def uniform(shape, seed: int, lo: float = -1.0, hi: float = 1.0) -> torch.Tensor:
    n = int(np.prod(shape)) if len(shape) else 1
    u = uniform01(n, seed)
    return torch.from_numpy((lo + (hi - lo) * u.astype(np.float64)).astype(np.float32)).reshape(tuple(shape))


def image_batch(shape, seed: int) -> torch.Tensor:
    """Integers 0..255 as fp32 - what reference data.py:123-126 yields (uint8 HWC -> float CHW)."""
    n = int(np.prod(shape))
    return torch.from_numpy(np.floor(uniform01(n, seed).astype(np.float64) * 256.0).astype(np.float32)).reshape(tuple(shape))


def _key_seed(key: str, seed: int) -> int:
    h = 1469598103934665603
    for ch in key.encode():
        h = ((h ^ ch) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return (h ^ (seed * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF


_BN = re.compile(r"features\.\d+\.1\.")


def fill_state_dict(shapes: dict, seed: int, scheme: str = "default") -> dict:
    """Deterministic values for every entry of `shapes` (name -> shape).

    scheme "default": conv/linear weights and biases ~ U(+-1/sqrt(fan_in)) (the distribution of torch's
    default init); BatchNorm weight in [0.5, 1.5], bias in +-0.1; running stats at their defaults.
    scheme "vgg": conv weights with the variance of kaiming_normal_(fan_out, relu), zero bias (GV7).
    MeanShift entries (sub_mean/add_mean) are NOT touched here - they have fixed analytic values.
    """
    out = {}
    for name, shape in shapes.items():
        s = _key_seed(name, seed)
        shape = tuple(shape)
        if name.endswith("num_batches_tracked"):
            out[name] = torch.zeros((), dtype=torch.long)
        elif name.endswith("running_mean"):
            out[name] = torch.zeros(shape)
        elif name.endswith("running_var"):
            out[name] = torch.ones(shape)
        elif name.endswith(("weight_u", "weight_v")):          # spectral norm's singular-vector estimates: unit vectors
            t = uniform(shape, s, -1.0, 1.0)
            out[name] = t / t.norm()
        elif len(shape) == 1 and _BN.search(name) and name.endswith("weight"):  # BatchNorm gamma (features.N.1.weight)
            out[name] = uniform(shape, s, 0.5, 1.5)
        elif len(shape) == 1 and _BN.search(name) and name.endswith("bias"):    # BatchNorm beta
            out[name] = uniform(shape, s, -0.1, 0.1)
        elif len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            if scheme == "vgg":
                fan_out = shape[0] * int(np.prod(shape[2:]))
                bound = math.sqrt(2.0 / fan_out) * math.sqrt(3.0)
            else:
                bound = 1.0 / math.sqrt(fan_in)
            out[name] = uniform(shape, s, -bound, bound)
        else:  # bias of a conv / linear: needs the fan_in of its weight
            wname = name[: -len("bias")] + "weight"
            if scheme == "vgg":
                out[name] = torch.zeros(shape)
            else:
                fan_in = int(np.prod(tuple(shapes[wname])[1:]))
                bound = 1.0 / math.sqrt(fan_in)
                out[name] = uniform(shape, s, -bound, bound)
    return out
