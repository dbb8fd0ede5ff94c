"""CPU restatement of the reference's inference plumbing - TEST INFRASTRUCTURE.
x8 self-ensemble as reference test.py:45-74 does it (numpy round-trips), PSNR / image conversion as reference
utils.py:10-41."""
from functools import reduce

import numpy as np
import torch


def _transform(v, op):
    a = v.detach().cpu().numpy()
    if op == "vflip":
        a = a[:, :, :, ::-1].copy()
    elif op == "hflip":
        a = a[:, :, ::-1, :].copy()
    elif op == "transpose":
        a = a.transpose((0, 1, 3, 2)).copy()
    return torch.from_numpy(a)


def x8_forward(img, model):
    """test.py:45-74"""
    inputs = [img]
    for tf in ("vflip", "hflip", "transpose"):
        inputs.extend([_transform(t, tf) for t in inputs])
    outs = [model(a) for a in inputs]
    for i in range(len(outs)):
        if i > 3:
            outs[i] = _transform(outs[i], "transpose")
        if i % 4 > 1:
            outs[i] = _transform(outs[i], "hflip")
        if (i % 4) % 2 == 1:
            outs[i] = _transform(outs[i], "vflip")
    return reduce(lambda x, y: x + y, outs) / len(outs)


def rgb2y(rgb):
    """utils.py:10-11"""
    return np.dot(rgb[..., :3], [65.738 / 256, 129.057 / 256, 25.064 / 256]) + 16


def tensor_to_img(t):
    """utils.py:13-18"""
    a = t.squeeze(0).detach().cpu().numpy()
    return a.clip(0, 255).round().transpose(1, 2, 0).astype(np.uint8)


def psnr_y(out, lbl):
    """utils.py:32-41"""
    o = rgb2y(tensor_to_img(out)).clip(0, 255).round()
    l = rgb2y(tensor_to_img(lbl)).clip(0, 255).round()
    return 20 * np.log10(255 / np.sqrt(np.mean((o - l) ** 2)))


def crop_augment(inp, lbl, patch, y, x, aug, scale=4):
    """reference data.py:79-126 for one sample with the random draws made explicit: _crop at LR (y, x) / HR (4y, 4x),
    _aug_data (bit 2 transpose, bit 1 vertical flip, bit 0 horizontal flip, in that order), _to_tensor (HWC -> CHW float)."""
    inp = inp[y:y + patch, x:x + patch, :]
    lbl = lbl[scale * y:scale * (y + patch), scale * x:scale * (x + patch), :]
    if (aug >> 2) & 1:
        inp, lbl = inp.transpose((1, 0, 2)).copy(), lbl.transpose((1, 0, 2)).copy()
    if (aug >> 1) & 1:
        inp, lbl = inp[::-1, :, :].copy(), lbl[::-1, :, :].copy()
    if aug & 1:
        inp, lbl = inp[:, ::-1, :].copy(), lbl[:, ::-1, :].copy()
    return torch.FloatTensor(inp.transpose(2, 0, 1).copy()), torch.FloatTensor(lbl.transpose(2, 0, 1).copy())
