"""Golden fixtures for the two index-only pieces of plumbing and the batch-1 Generator, made by IMPORTING THE REFERENCE
(build container only; needs /root/reference, never runs on the GPU box).

Run:  python tests/golden/make_golden_plumbing.py          (about 20 s)

  GV10  x8 self-ensemble: the reference's own `x8_forward` (reference test.py:45-74), imported from test.py itself with
        `sys.argv`, `imageio` and `Tensor.cuda` stubbed, run on a NON-equivariant toy model (tests/helpers.py `x8_toy_model`: an asymmetric
        3x3 conv + PixelShuffle(2), dyadic weights from the build's counter-based generator - exact in fp32) on square and non-square inputs.
  GV12  crop / augment: the reference's `SRDataset._crop`, `_aug_data`, `_to_tensor` (reference data.py:79-126), called on a
        bare instance under `random.seed(s)` - every one of the 8 `aug_idx` values, corner crops, both crop types, images
        whose sizes are not multiples of anything.  The three `random.randint` draws of each case are re-drawn from the same
        seed and stored next to the outputs, so a checker can be given them explicitly.
  GV2c  Generator 256 ch x 32 blocks at [1,3,48,48] -> [1,3,192,192] (reference model/pesr.py:28-38; the shape of reference
        test.py:100-106 and BASELINE config 1): sampled output, sums, sampled gradients under L1, and the same gradients
        in float64 (the reference module itself, `.double()`).
Only data is written (seeds, shapes, draws, outputs) - no reference source text.
"""
import os
import random
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import make_golden as MG  # noqa: E402  (torchvision stub + reference import, deterministic fills, save)
from helpers import x8_toy_model  # noqa: E402
from oracle import detrand  # noqa: E402

G_GRAD_KEYS = ["sub_mean.weight", "sub_mean.bias", "embed.weight", "embed.bias", "body.0.body.0.weight",
               "body.0.body.0.bias", "body.15.body.2.weight", "body.31.body.2.weight", "body.31.body.2.bias", "body.32.weight",
               "upsample.0.weight", "upsample.0.bias", "upsample.2.weight", "upsample.2.bias", "upsample.4.weight",
               "upsample.4.bias", "add_mean.weight", "add_mean.bias"]

X8_CASES = [((1, 3, 10, 14), 301), ((1, 3, 12, 12), 302), ((2, 3, 7, 9), 303)]       # (input shape, data seed)
X8_WEIGHT_SEED, X8_BIAS_SEED = 310, 311

# (LR height, LR width, patch, first seed to try): HR images are 4x; sizes chosen so that corner crops are frequent
CROP_IMAGES = [(13, 17, 12, 0), (11, 14, 8, 100), (16, 16, 16, 200), (21, 20, 20, 300)]


def uint8_image(h, w, seed):
    """HWC uint8 image from the build's counter-based generator."""
    return detrand.image_batch((1, 3, h, w), seed)[0].permute(1, 2, 0).contiguous().numpy().astype(np.uint8)


def import_reference_entrypoints():
    """reference data.py and test.py as modules.  test.py parses sys.argv and prints its banner at import; data.py and test.py
    import imageio (absent here: an empty stub - neither function under test touches it); x8_forward calls `.cuda()` on CPU
    tensors (no GPU here: Tensor.cuda returns the tensor itself)."""
    MG.import_reference()                                    # torchvision stub, /root/reference on sys.path, `model`, `utils`
    sys.modules.setdefault("imageio", types.ModuleType("imageio"))
    torch.Tensor.cuda = lambda self, *a, **k: self
    argv, sys.argv = sys.argv, ["test.py"]
    try:
        import data as ref_data
        import test as ref_test
    finally:
        sys.argv = argv
    assert ref_test.__file__.startswith("/root/reference/") and ref_data.__file__.startswith("/root/reference/")
    return ref_data, ref_test


def main():
    torch.set_num_threads(8)
    ref_data, ref_test = import_reference_entrypoints()

    # ---- GV10: x8 self-ensemble --------------------------------------------------------------------------------
    model = x8_toy_model(X8_WEIGHT_SEED, X8_BIAS_SEED)     # exact arithmetic: the fixture is the same bits on any machine
    arrs = {"shapes": np.array([s for s, _ in X8_CASES], dtype=np.int64), "seeds": np.array([s for _, s in X8_CASES], dtype=np.int64),
            "weight_seed": np.int64(X8_WEIGHT_SEED), "bias_seed": np.int64(X8_BIAS_SEED)}
    with torch.no_grad():
        for i, (shape, seed) in enumerate(X8_CASES):
            x = detrand.image_batch(shape, seed)
            out = ref_test.x8_forward(x, model)
            plain = model(x)
            assert out.shape == plain.shape and (out - plain).abs().max() > 1.0     # the toy model is not equivariant
            arrs[f"out{i}"] = out
    MG.save("gv10_x8", **arrs)

    # ---- GV12: crop / augment ------------------------------------------------------------------------------------
    ds = object.__new__(ref_data.SRDataset)                 # bare instance: __init__ reads the dataset directory
    ds.scale = 4
    rows, n = [], 0
    arrs = {}
    for (ih, iw, ps, seed0) in CROP_IMAGES:
        ds.patch_size = ps
        inp = uint8_image(ih, iw, 400 + seed0)
        lbl = uint8_image(4 * ih, 4 * iw, 401 + seed0)
        seen, edges, s = set(), set(), seed0
        want_edges = {e for e, room in (("top", ih > ps), ("bottom", ih > ps), ("left", iw > ps), ("right", iw > ps)) if room}
        while (len(seen) < 8 or edges != want_edges or s < seed0 + 10) and s < seed0 + 64:
            random.seed(s)
            h = random.randint(0, ih - ps); w = random.randint(0, iw - ps); aug = random.randint(0, 7)
            random.seed(s)
            ci, cl = ds._crop(inp, lbl, "random")
            ai, al = ds._aug_data(ci, cl)
            ti, tl = ds._to_tensor(ai, al)
            edges |= {e for e, hit in (("top", h == 0), ("bottom", h == ih - ps), ("left", w == 0), ("right", w == iw - ps))
                      if hit and e in want_edges}
            seen.add(aug)
            rows.append([ih, iw, ps, 400 + seed0, 401 + seed0, s, 1, h, w, aug])
            arrs[f"inp{n}"], arrs[f"lbl{n}"] = ti, tl
            n += 1
            s += 1
        assert len(seen) == 8 and edges == want_edges, (ih, iw, ps, seen, edges)
        # crop_type 'fixed' (the validation set's, reference train.py:92-93): no crop draw, one aug draw
        random.seed(s)
        aug = random.randint(0, 7)
        random.seed(s)
        ci, cl = ds._crop(inp, lbl, "fixed")
        ti, tl = ds._to_tensor(*ds._aug_data(ci, cl))
        rows.append([ih, iw, ps, 400 + seed0, 401 + seed0, s, 0, 0, 0, aug])
        arrs[f"inp{n}"], arrs[f"lbl{n}"] = ti, tl
        n += 1
    arrs["cases"] = np.array(rows, dtype=np.int64)      # ih, iw, patch, lr seed, hr seed, random.seed, random crop?, y, x, aug
    for i in range(n):          # the reference hands out float32 tensors holding the uint8 values: stored as uint8 (lossless, asserted)
        for k in (f"inp{i}", f"lbl{i}"):
            t = arrs[k]
            assert t.dtype == torch.float32 and torch.equal(t, t.to(torch.uint8).float())
            arrs[k] = t.to(torch.uint8).contiguous()
    arrs["out_dtype"] = np.array("float32")
    MG.save("gv12_crop_aug", **arrs)
    print("   crop/augment cases:", n, "aug values", sorted(set(r[9] for r in rows)))

    # ---- GV2c: full generator at batch 1 (BASELINE config 1's shape) -------------------------------------------
    R, _ = MG.import_reference()
    full = {"num_channels": 256, "depth": 32, "res_scale": 0.1}
    G = MG.load_det(R.Generator(full), seed=0)
    lr = detrand.image_batch((1, 3, 48, 48), 1234)
    hr = detrand.image_batch((1, 3, 192, 192), 1235)
    sr = G(lr)
    loss = F.l1_loss(sr, hr)
    loss.backward()
    idx = MG.sample_idx(sr.numel(), 8192, 99)
    arrs = {"sr_idx": idx, "sr_val": sr.detach().reshape(-1)[idx], "sr_sum": sr.sum(), "sr_abs_sum": sr.abs().sum(), "loss": loss}
    params = dict(G.named_parameters())
    for k in G_GRAD_KEYS:
        g = params[k].grad.reshape(-1)
        gi = MG.sample_idx(g.numel(), 2048, 7)
        arrs["gidx." + k], arrs["gval." + k], arrs["gmax." + k] = gi, g[gi], g.abs().max()
    G64 = MG.load_det(R.Generator(full), seed=0).double()
    sr64 = G64(lr.double())
    loss64 = F.l1_loss(sr64, hr.double())
    loss64.backward()
    arrs["sr_val64"], arrs["loss64"] = sr64.detach().reshape(-1)[idx], loss64
    worst = []
    p64 = dict(G64.named_parameters())
    for k in G_GRAD_KEYS:
        g = p64[k].grad.reshape(-1)
        arrs["g64." + k], arrs["gmax64." + k] = g[arrs["gidx." + k]], g.abs().max()
        worst.append(float((arrs["gval." + k].double() - arrs["g64." + k]).abs().max() / g.abs().max()))
    arrs["floor_worst"] = np.float64(max(worst))
    MG.save("gv2c_generator_full_b1", **arrs)
    print("   GV2c reference fp32 vs fp64 gradient error per tensor: worst %.2e median %.2e" % (max(worst), sorted(worst)[len(worst) // 2]))


if __name__ == "__main__":
    main()
