"""Focal-weighted BCE-with-logits (reference model/focal_loss.py), mean-reduced.

The reference multiplies BCE by w = (1-pt)^gamma passed as the `weight` argument; under its pinned
torch 0.4 the gradient flowed through BOTH w and the BCE term (on torch >= 1.x that call raises in
training).  This module reproduces the torch-0.4 semantics with a closed-form backward.
The operands are [B, 1] logits - scalar-sized host-side torch ops, no kernel needed.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class _FocalFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, t, gamma):
        p = x.sigmoid()
        pt = p * t + (1 - p) * (1 - t)
        w = (1 - pt).pow(gamma)
        bce = F.binary_cross_entropy_with_logits(x, t, reduction='none')
        ctx.gamma = gamma
        ctx.save_for_backward(p, pt, w, bce, t)
        return (w * bce).mean()

    @staticmethod
    def backward(ctx, g):
        p, pt, w, bce, t = ctx.saved_tensors
        gamma = ctx.gamma
        if gamma == 0:
            dw = torch.zeros_like(p)
        else:
            dw = -gamma * (1 - pt).pow(gamma - 1) * (2 * t - 1) * p * (1 - p)
        return g * (dw * bce + w * (p - t)) / p.numel(), None, None


class FocalLoss(nn.Module):
    def __init__(self, gamma):
        super().__init__()
        self.gamma = gamma

    def forward(self, x, t):
        return _FocalFn.apply(x, t, self.gamma)
