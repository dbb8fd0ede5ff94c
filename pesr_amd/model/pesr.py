"""Generator (EDSR-style trunk + x4 sub-pixel upsampler) and Discriminator (strided VGG-style stack + 2 Linear layers)
on the HIP kernels, API- and checkpoint-compatible with reference model/pesr.py: same `opt` keys, same attribute names
(`sub_mean, embed, body, upsample, add_mean` / `features, classifier`), hence the same state_dict keys and shapes, and the
same order of parameter-initialisation RNG draws.
"""
import torch
import torch.nn as nn

from .. import functional as PF
from .. import ops
from .basic import BasicBlock, Conv, MeanShift, ResBlock, Upsampler, nchw, nhwc

_DIV2K_MEAN = (0.4488, 0.4371, 0.4040)      # reference model/pesr.py:13
_UNIT_STD = (1.0, 1.0, 1.0)


def discriminator_plan(first_width=64, stages=7):
    """(cin, cout, stride) of the BasicBlocks after the 3->64 stem: stride 2 on even stages, width doubling on odd ones
    (reference model/pesr.py:56-65) -> 64s2, 128, 128s2, 256, 256s2, 512, 512s2."""
    plan, width = [], first_width
    for stage in range(stages):
        if stage % 2:
            plan.append((width, 2 * width, 1))
            width *= 2
        else:
            plan.append((width, width, 2))
    return plan


class Generator(nn.Module):
    def __init__(self, opt):
        super().__init__()
        width, depth, res_scale = opt['num_channels'], opt['depth'], opt['res_scale']
        # The reference builds the trunk before everything else (RNG order) but registers it third (state_dict order).
        trunk = [ResBlock(width, 3, act=nn.ReLU(True), res_scale=res_scale) for _ in range(depth)]
        trunk.append(Conv(width, width, 3))
        self.sub_mean = MeanShift(255, _DIV2K_MEAN, _UNIT_STD)
        self.embed = Conv(3, width, 3)
        self.body = nn.Sequential(*trunk)
        self.upsample = Upsampler(width)
        self.add_mean = MeanShift(255, _DIV2K_MEAN, _UNIT_STD, 1)

    def forward(self, x):
        feat = self.embed(self.sub_mean(x))
        h = feat
        for block in self.body[:-1]:
            h = block(h)
        last = self.body[-1]
        # trunk tail conv and the global skip (`res += x`, reference model/pesr.py:32-33) in one kernel
        h = nchw(PF.ConvAddFn.apply(nhwc(h), nhwc(feat), last.weight, last.bias, last.packed))
        return self.add_mean(self.upsample(h))


class Discriminator(nn.Module):
    def __init__(self, opt):
        super().__init__()
        lrelu = nn.LeakyReLU(negative_slope=0.2, inplace=True)
        use_sn = opt['spectral_norm']
        blocks = [BasicBlock(3, 64, 3, bn=True, act=lrelu, sn=use_sn)]
        for cin, cout, stride in discriminator_plan():
            blocks.append(BasicBlock(cin, cout, 3, stride=stride, bn=True, act=lrelu, sn=use_sn))
        self.features = nn.Sequential(*blocks)
        # the last block emits NCHW-contiguous so that .view(B, -1) has the reference's column order (c, y, x)
        self.features[-1].flatten_output = True
        side = (opt['patch_size'] * 4) // 16                 # four stride-2 stages on the HR patch
        self.classifier = nn.Sequential(nn.Linear(512 * side * side, 1024), lrelu, nn.Linear(1024, 1))

    def forward_features(self, x):
        """The eight conv + BatchNorm + LeakyReLU blocks: [B, 3, H, W] -> [B, 512 * side * side] in the reference's NCHW column order
        (reference model/pesr.py:78-79).  One call = one set of BatchNorm batch statistics, as in the reference."""
        # the blocks are chained here (not through nn.Sequential.forward) so that each block knows its producer: the tensors in between
        # are seen by nobody else, which lets a block's input-gradient kernel do the BatchNorm reductions of the block in front of it
        flat, link = x, None
        for blk in self.features:
            flat, link = blk.forward_linked(flat, link) if isinstance(blk, BasicBlock) else (blk(flat), None)
        return flat.view(flat.size(0), -1)

    def classify(self, flat, grad_rows=None):
        """Linear -> LeakyReLU -> Linear (reference model/pesr.py:69-75,80) on the flattened features.  The classifier has no BatchNorm,
        so its rows are independent: the train step hands it the features of TWO calls at once ([hr; sr]) and the 302 MB weight matrix
        of classifier.0 is streamed once instead of twice, forward and backward (pesr_amd.step.Trainer.gan_step).  grad_rows: only the
        first grad_rows rows need an input gradient (functional.LinearFn)."""
        fc1, act, fc2 = self.classifier
        hidden = PF.LinearFn.apply(flat, fc1.weight, fc1.bias, ops.ACT_LRELU, act.negative_slope, grad_rows)
        return PF.LinearFn.apply(hidden, fc2.weight, fc2.bias, ops.ACT_NONE, 0.0)

    PAIR_ROWS = 16   # rows per call inside a paired classifier pass: the Linear kernels add rows in chains of sixteen (linear.hip)

    def classify_pair(self, fa, fb, grad_first_only=False):
        """classify() of two calls' features in ONE pass over the weights: -> (logits of fa, logits of fb).  Each call takes a block of
        sixteen rows (shorter batches are padded with zero rows, which add exact zeros), and the kernels sum every block as a chain of its
        own before adding the blocks - so forward values, input gradients, weight and bias gradients are bit for bit those of two
        separate calls.  Batches above sixteen: call classify() twice.  grad_first_only: fb came from a no_grad pass."""
        B, R = fa.shape[0], self.PAIR_ROWS
        assert fb.shape == fa.shape and B <= R
        if B < R:
            pad = fa.new_zeros((R - B, fa.shape[1]))
            x = torch.cat([fa, pad, fb, pad])
        else:
            x = torch.cat([fa, fb])
        preds = self.classify(x, grad_rows=R if grad_first_only else None)
        return PF.SplitRowsFn.apply(preds, B, R)

    def forward(self, x):
        return self.classify(self.forward_features(x))


def _forward_second_order(self, x):
    """D(x) through twice-differentiable, un-fused functions (pesr_amd.functional *2Fn): the extra forward of the gradient
    penalty (reference train.py:216-226), whose result goes into torch.autograd.grad(..., create_graph=True).  Same values as
    forward() up to fp32 rounding; BatchNorm runs in training mode and updates its running statistics like any other call."""
    h = nhwc(x)
    for blk in self.features:
        conv, bn, act = blk[0], blk[1], blk[2]
        z = PF.Conv2Fn.apply(h, conv.effective_weight(), conv.packed, conv.stride)
        u = PF.Bn2Fn.apply(z, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, bn.eps, bn.momentum)
        h = PF.LRelu2Fn.apply(u, float(act.negative_slope))
    flat = nchw(h).contiguous().view(h.size(0), -1)          # NCHW flatten (reference model/pesr.py:79)
    fc1, act, fc2 = self.classifier
    hidden = PF.LRelu2Fn.apply(PF.Linear2Fn.apply(flat, fc1.weight, fc1.bias), float(act.negative_slope))
    return PF.Linear2Fn.apply(hidden, fc2.weight, fc2.bias)


Discriminator.forward_second_order = _forward_second_order
