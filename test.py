#!/usr/bin/env python3
"""x4 SR inference entry point (counterpart of reference test.py): perceptual model, optional PSNR model with x8
self-ensemble, image-space blend `alpha*out + (1-alpha)*out_psnr`, PNG output.  Same flags as the reference
(test.py:13-33); device-agnostic plumbing; the Generator itself runs on the MI355X kernels.
"""
import argparse
import glob
import os

import numpy as np
import torch

from utils import default_device, imgs_to_tensors, tensors_to_imgs


# (flag, type, default, help) - the reference's flags and defaults (reference test.py:15-33)
_FLAGS = [
    ("dataset", str, "Set5", "folder under data/origin/test/ with an LR/ sub-folder of PNGs"),
    ("perceptual_model", str, "check_point/PESR/train/PERC_model.pt", "GAN-phase generator checkpoint"),
    ("psnr_model", str, "check_point/PESR/pretrain/PSNR_model.pt", "L1-pretrained generator checkpoint (used when alpha != 1)"),
    ("num_channels", int, 256, "generator width"),
    ("num_blocks", int, 32, "generator depth (residual blocks)"),
    ("res_scale", float, 0.1, "residual scaling"),
    ("alpha", float, 1, "image-space blend: alpha * perceptual + (1 - alpha) * x8-ensembled PSNR output"),
    ("save_path", str, "results", "output folder"),
]


def build_parser():
    parser = argparse.ArgumentParser(description="x4 super-resolution of a folder of LR images")
    for name, typ, default, text in _FLAGS:
        parser.add_argument("--" + name, type=typ, default=default, help=text)
    # an addition (not a reference flag): the optional bf16-operand mode of the build, DESIGN.md section 4c
    parser.add_argument("--precision", type=str, default="fp32", choices=["fp32", "bf16", "split-bf16"],
                        help="bf16: the generator's 3x3 convs round their operands to bf16 (fp32 accumulation and tensors): ~3x faster, "
                             "pixel values differ from the fp32 result by <= 1 grey level on a small fraction of pixels")
    return parser


# the three generators of the 8-element dihedral group, in the reference's order (test.py:58-60), as tensor ops
_TRANSFORMS = {
    "vflip": lambda t: t.flip(3),         # reference 'vflip' reverses the W axis
    "hflip": lambda t: t.flip(2),         # reference 'hflip' reverses the H axis
    "transpose": lambda t: t.transpose(2, 3),
}


def x8_forward(img, model):
    """Self-ensemble over the 8 flips/transposes (reference test.py:45-74), done on-device instead of through
    numpy round-trips.  Inputs are built by applying vflip, hflip, transpose cumulatively; output i is mapped back
    with transpose (i > 3), hflip (i % 4 > 1), vflip (i odd), then the 8 are averaged."""
    inputs = [img]
    for name in ("vflip", "hflip", "transpose"):
        inputs.extend([_TRANSFORMS[name](t).contiguous() for t in inputs])
    outputs = [model(t) for t in inputs]
    for i in range(len(outputs)):
        o = outputs[i]
        if i > 3:
            o = _TRANSFORMS["transpose"](o)
        if i % 4 > 1:
            o = _TRANSFORMS["hflip"](o)
        if (i % 4) % 2 == 1:
            o = _TRANSFORMS["vflip"](o)
        outputs[i] = o
    total = outputs[0]
    for o in outputs[1:]:
        total = total + o
    return total / len(outputs)


def _read_png(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGB"))


def _write_png(path, img):
    from PIL import Image
    Image.fromarray(img).save(path)


def main(argv=None):
    args = build_parser().parse_args(argv)
    from model import Generator
    if args.precision != "fp32":
        from pesr_amd import ops as _ops
        _ops.set_precision(args.precision)
    device = default_device()
    lr_paths = sorted(glob.glob(os.path.join("data/origin/test/", args.dataset, "LR", "*.png")))
    opt = {"num_channels": args.num_channels, "depth": args.num_blocks, "res_scale": args.res_scale}
    model = Generator(opt)
    model.load_state_dict(torch.load(args.perceptual_model, map_location="cpu"))
    model = model.to(device)
    print("Number of parameters:", sum(p.nelement() for p in model.parameters()))
    model_psnr = None
    if args.alpha != 1:
        model_psnr = Generator(opt)
        model_psnr.load_state_dict(torch.load(args.psnr_model, map_location="cpu"))
        model_psnr = model_psnr.to(device)
    save_path = os.path.join(args.save_path, args.dataset)
    os.makedirs(save_path, exist_ok=True)
    with torch.no_grad():
        for i, lr_path in enumerate(lr_paths):
            [inp] = imgs_to_tensors([_read_png(lr_path)], device)
            out = model(inp)
            if model_psnr is not None:
                out = args.alpha * out + (1 - args.alpha) * x8_forward(inp, model_psnr)
            [img] = tensors_to_imgs([out])
            _write_png(os.path.join(save_path, os.path.basename(lr_path)), img)
            print("Tested %d img(s)" % (i + 1))
    print("Finish")


if __name__ == "__main__":
    main()
