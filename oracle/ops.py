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


# ---- the OPTIONAL bf16-operand mode of the build (no counterpart in the reference, which is fp32 only) ---------------
def round_bf16(t):
    """Round to the nearest bfloat16 (ties to even), returned in t's dtype: what v_cvt_pk_bf16_f32 does to an MFMA operand."""
    return t.to(torch.float32).to(torch.bfloat16).to(t.dtype)


def conv3x3_bf16(x, w, b=None, stride=1):
    """model/basic.py:4-7 `Conv` as the bf16 mode computes its forward: BOTH operands of every product rounded to bf16, the
    products (exact in fp32) summed - here in float64, so the kernel is checked to fp32-accumulation accuracy - bias in fp32."""
    y = F.conv2d(round_bf16(x).double(), round_bf16(w).double(), None, stride=stride, padding=1).float()
    return y if b is None else y + b.view(1, -1, 1, 1)


def conv3x3_bf16_grads(x, w, dy, need_bias=True):
    """(dx, dw, db) of the bf16 mode: dx from round(dy) and round(w), dw from round(x) and round(dy), both summed in float64;
    db = sum of the UN-rounded dy (the kernel adds it up in fp32 on the vector unit)."""
    xr = round_bf16(x).double().requires_grad_(True)
    wr = round_bf16(w).double().requires_grad_(True)
    F.conv2d(xr, wr, None, padding=1).backward(round_bf16(dy).double())
    return xr.grad.float(), wr.grad.float(), (dy.double().sum(dim=(0, 2, 3)).float() if need_bias else None)
