"""Op-level CPU restatements (logical NCHW tensors, fp32) - the checker for each HIP kernel.

Every function names the reference line whose ATen op it restates.
"""
import torch
import torch.nn.functional as F


def conv3x3(x, w, b=None, stride=1):
    """reference model/basic.py:4-7 `Conv`: nn.Conv2d(k=3, padding=1, stride, bias)."""
    return F.conv2d(x, w, b, stride=stride, padding=1)


def conv3x3_grads(x, w, dy, stride=1, need_bias=True):
    """(dx, dw, db) of conv3x3 via autograd on CPU (ATen convolution_backward)."""
    x = x.detach().clone().requires_grad_(True)
    w = w.detach().clone().requires_grad_(True)
    b = torch.zeros(w.shape[0], dtype=w.dtype, requires_grad=True) if need_bias else None
    y = F.conv2d(x, w, b, stride=stride, padding=1)
    y.backward(dy)
    return x.grad, w.grad, (b.grad if need_bias else None)


def pixel_shuffle(x, r=2):
    """reference model/basic.py:57,59 nn.PixelShuffle(2): out[n,c,2h+i,2w+j] = in[n,4c+2i+j,h,w]."""
    return F.pixel_shuffle(x, r)


def pixel_unshuffle(x, r=2):
    return F.pixel_unshuffle(x, r)
