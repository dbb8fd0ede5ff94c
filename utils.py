"""Image <-> tensor helpers and the Y-channel PSNR of the reference (reference utils.py:10-41), device-agnostic.

Tensors carry raw 0..255 values (no /255 anywhere).  Metrics run in numpy on the host exactly as the
reference does; nothing here is on the hot path.
"""
import os
import shutil

import numpy as np
import torch

_Y_COEF = np.array([65.738, 129.057, 25.064]) / 256.0


def default_device():
    return torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")


def rgb2y(rgb):
    """ITU-R BT.601 luma of an HWC RGB array, offset 16 (reference utils.py:10-11)."""
    return np.dot(rgb[..., :3], _Y_COEF) + 16


def tensors_to_imgs(tensors):
    """[1,3,H,W] float tensors -> uint8 HWC arrays: clip to 0..255, round half-to-even, cast (reference utils.py:13-18)."""
    out = []
    for t in tensors:
        a = t.detach().squeeze(0).float().cpu().numpy()
        out.append(np.clip(a, 0, 255).round().transpose(1, 2, 0).astype(np.uint8))
    return out


def imgs_to_tensors(imgs, device=None):
    """uint8 HWC arrays -> [1,3,H,W] float tensors on `device` (reference utils.py:20-25 hard-codes .cuda())."""
    device = default_device() if device is None else device
    return [torch.from_numpy(np.ascontiguousarray(i.transpose(2, 0, 1))[None].astype(np.float32)).to(device) for i in imgs]


def normalize(tensors):
    return [t.clamp(0, 255) / 255 for t in tensors]


def compute_PSNR(out, lbl):
    """PSNR on the rounded Y channel of two [1,3,H,W] tensors (reference utils.py:32-41).  GPU tensors are measured on the
    device (pesr_amd.ops.psnr_y, bit-identical: integer-valued terms in double); only the scalar comes back."""
    if torch.is_tensor(out) and torch.is_tensor(lbl) and out.is_cuda and lbl.is_cuda and out.dtype == torch.float32 \
            and lbl.dtype == torch.float32 and out.dim() == 4 and out.shape[0] == 1 and out.shape == lbl.shape:
        from pesr_amd import ops
        return float(ops.psnr_y(out.detach(), lbl.detach())[1])
    o, l = tensors_to_imgs([out, lbl])
    yo = np.clip(rgb2y(o), 0, 255).round()
    yl = np.clip(rgb2y(l), 0, 255).round()
    rmse = np.sqrt(np.mean((yo - yl) ** 2))
    return 20 * np.log10(255 / rmse)


def update_tensorboard(epoch, tb, img_idx, inp, out, lbl):
    if tb is None:
        return
    inp, out, lbl = normalize([inp, out, lbl])
    if epoch == 1:
        tb.add_image(f"{img_idx}_LR", inp, epoch)
        tb.add_image(f"{img_idx}_HR", lbl, epoch)
    tb.add_image(f"{img_idx}_SR", out, epoch)


def clean_and_mk_dir(path):
    if os.path.exists(path):
        shutil.rmtree(path)
    os.makedirs(path)
