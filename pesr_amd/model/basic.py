"""Building blocks with the reference's constructors and state_dict schema (reference model/basic.py),
running on the HIP kernels of libpesr_hip.so.

Every module takes and returns logical NCHW fp32 tensors; outputs are in torch.channels_last memory
format (NHWC physically), so chaining modules never copies.  CPU tensors raise: there is no fallback.
Parameters keep the reference's names, shapes (OIHW) and default initialisation (drawn through the very
torch initialisers the reference's nn.Conv2d / nn.Linear would call, in the same order, so a seeded
construction consumes the RNG identically).
"""
import torch
import torch.nn as nn

from .. import functional as PF
from .. import ops


def nhwc(x: torch.Tensor) -> torch.Tensor:
    """logical NCHW -> NHWC view (zero-copy for channels_last inputs)."""
    return x.permute(0, 2, 3, 1)


def nchw(y: torch.Tensor) -> torch.Tensor:
    """NHWC tensor -> logical NCHW view (channels_last strides)."""
    return y.permute(0, 3, 1, 2)


def spectral_norm(conv, eps=1e-12):
    """reference model/basic.py:25 calls `spectral_norm(Conv(...))` without importing it (NameError when --spectral_norm true,
    SURVEY Q3); the evident intent - torch.nn.utils.spectral_norm, one power iteration per training forward - is what this
    does to a pesr_amd Conv: same parameter / buffer names (`weight_orig`, `weight_u`, `weight_v`: torch's state_dict schema),
    same RNG draws in the same order (u then v, standard normal, normalised), the normalised weight computed by the HIP
    kernels of spectral_norm.hip at every forward."""
    import torch.nn.functional as F
    w = conv.weight
    with torch.no_grad():
        u = F.normalize(w.new_empty(w.shape[0]).normal_(0, 1), dim=0, eps=eps)
        v = F.normalize(w.new_empty(w[0].numel()).normal_(0, 1), dim=0, eps=eps)
    del conv.weight
    conv.register_parameter("weight_orig", w)
    conv.register_buffer("weight_u", u)
    conv.register_buffer("weight_v", v)
    conv._sn_eps = eps
    return conv


class Conv(nn.Module):
    """k x k conv, padding k//2 (reference model/basic.py:4-7).  kernel_size 3 - the only size the reference's networks use -
    runs on the MFMA / Winograd kernels; any other odd size on the generic kernels of conv_kxk.hip (complete, untuned)."""

    def __init__(self, in_planes, out_planes, kernel_size, stride=1, bias=True):
        super().__init__()
        if kernel_size % 2 == 0:
            raise NotImplementedError("pesr_amd Conv: even kernel sizes change the output size under padding k//2; odd sizes only")
        init = nn.Conv2d(in_planes, out_planes, kernel_size, padding=kernel_size // 2, stride=stride, bias=bias)
        self.weight = nn.Parameter(init.weight.data)
        self.bias = nn.Parameter(init.bias.data) if bias else None
        self.in_channels, self.out_channels, self.stride, self.kernel_size = in_planes, out_planes, stride, kernel_size
        self.packed = PF.PackedConvWeights(ps=False)
        self._sn_eps = None          # set by spectral_norm(): the weight is then weight_orig / sigma, recomputed per forward

    def effective_weight(self):
        """The weight this forward convolves with: the parameter itself, or - under spectral_norm - weight_orig / sigma."""
        if self._sn_eps is None:
            return self.weight
        return PF.spectral_normalize(self.weight_orig, self.weight_u, self.weight_v, self.training, self._sn_eps)

    def forward(self, x, act=ops.ACT_NONE, relu_in=False, relu_grad_by_consumer=False):
        if self.kernel_size != 3:
            assert act == ops.ACT_NONE and not relu_in, "fused activations exist for the 3x3 kernels only"
            return nchw(PF.ConvKxKFn.apply(nhwc(x), self.effective_weight(), self.bias, self.stride))
        return nchw(PF.conv3x3(nhwc(x), self.effective_weight(), self.bias, self.packed, self.stride, act, relu_in,
                               relu_grad_by_consumer))

    def extra_repr(self):
        return (f"{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, stride={self.stride}, "
                f"bias={self.bias is not None}")


class MeanShift(nn.Module):
    """1x1 conv 3->3 initialised to (x -/+ rgb_range*mean)/std (reference model/basic.py:9-17).  As in the
    reference, `self.requires_grad = False` is a no-op: weight and bias ARE trainable."""

    def __init__(self, rgb_range, rgb_mean, rgb_std, sign=-1):
        super().__init__()
        init = nn.Conv2d(3, 3, kernel_size=1)  # draws (then discards) an init, as the reference does
        std = torch.Tensor(rgb_std)
        w = torch.eye(3).view(3, 3, 1, 1)
        w.div_(std.view(3, 1, 1, 1))
        b = sign * rgb_range * torch.Tensor(rgb_mean)
        b.div_(std)
        self.weight = nn.Parameter(init.weight.data.copy_(w))
        self.bias = nn.Parameter(init.bias.data.copy_(b))
        self.requires_grad = False

    def forward(self, x):
        if x.dim() == 4 and x.is_contiguous() and not x.is_contiguous(memory_format=torch.channels_last):
            # NCHW-contiguous network input: the layout change is folded into the kernel
            return nchw(PF.MeanShiftFn.apply(x, self.weight, self.bias, True, False))
        return nchw(PF.MeanShiftFn.apply(nhwc(x), self.weight, self.bias, False, False))


class PixelShuffle(nn.Module):
    """nn.PixelShuffle(2) as a standalone kernel (Upsampler fuses it into the conv epilogue instead)."""

    def __init__(self, upscale_factor=2):
        super().__init__()
        assert upscale_factor == 2
        self.upscale_factor = upscale_factor

    def forward(self, x):
        return nchw(_PixelShuffleFn.apply(nhwc(x)))


class _PixelShuffleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return ops.pixel_shuffle_fwd(x.contiguous())

    @staticmethod
    def backward(ctx, gy):
        return ops.pixel_shuffle_bwd(gy.contiguous())


def _act_slope(act):
    """negative slope that stands for an activation module in the fused BN / conv epilogues (None: identity)."""
    if act is None:
        return 1.0
    if isinstance(act, nn.LeakyReLU):
        return float(act.negative_slope)
    if isinstance(act, nn.ReLU):
        return 0.0
    raise NotImplementedError(f"activation {type(act).__name__}: the HIP epilogues implement ReLU, LeakyReLU and none")


def _conv_act(conv, x_nhwc, act):
    """conv (+bias) followed by an activation module (or None), un-fused with whatever comes next."""
    slope = _act_slope(act)
    w = conv.effective_weight()
    if conv.kernel_size != 3:       # generic kernel, activation as its own (differentiable) elementwise op
        z = PF.ConvKxKFn.apply(x_nhwc, w, conv.bias, conv.stride)
        return z if slope == 1.0 else PF.LRelu2Fn.apply(z, slope)
    if slope == 1.0:
        return PF.conv3x3(x_nhwc, w, conv.bias, conv.packed, conv.stride)
    if slope == 0.0:
        return PF.conv3x3(x_nhwc, w, conv.bias, conv.packed, conv.stride, act=ops.ACT_RELU)
    return PF.ConvLReluFn.apply(x_nhwc, w, conv.bias, conv.packed, conv.stride, slope)


def _conv_bn_act(conv, bn, x_nhwc, act, y_nchw=False, prev_link=None, link=None):
    if conv.kernel_size != 3:       # generic conv -> BatchNorm -> activation, un-fused (training-mode statistics only)
        if not (bn.training or bn.running_mean is None):
            raise NotImplementedError("eval-mode BatchNorm behind a conv with kernel_size != 3")
        z = PF.ConvKxKFn.apply(x_nhwc, conv.effective_weight(), conv.bias, conv.stride)
        u = PF.Bn2Fn.apply(z, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, bn.eps, bn.momentum)
        slope = _act_slope(act)
        y = u if slope == 1.0 else PF.LRelu2Fn.apply(u, slope)
        return nchw(y).contiguous() if y_nchw else y
    return PF.ConvBnLReluFn.apply(x_nhwc, conv.effective_weight(), conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                  bn.num_batches_tracked, conv.packed, conv.stride, bn.eps, bn.momentum, _act_slope(act), y_nchw,
                                  bn.training or bn.running_mean is None, prev_link, link)


class BasicBlock(nn.Sequential):
    """conv -> [BatchNorm2d] -> [act] (reference model/basic.py:19-31); the Discriminator's unit (no bias, BN in training mode,
    LeakyReLU(0.2)).  Every combination the constructor accepts runs on the HIP kernels: conv bias, bn on / off, BatchNorm in
    training or eval mode, act = ReLU / LeakyReLU / None."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, bias=False, bn=True, act=nn.ReLU(True),
                 sn=True):
        if sn:
            conv = spectral_norm(Conv(in_channels, out_channels, kernel_size, stride, bias))
        else:
            conv = Conv(in_channels, out_channels, kernel_size, stride, bias)
        m = [conv]
        if bn:
            m.append(nn.BatchNorm2d(out_channels))
        if act is not None:
            m.append(act)
        super().__init__(*m)
        self._has_bn, self._has_act = bn, act is not None
        self.flatten_output = False  # set by Discriminator on its last block: emit NCHW-contiguous for .view(B, -1)

    def forward_linked(self, x, prev_link):
        """forward() for a caller that chains blocks itself and keeps the tensors in between to itself (Discriminator.forward):
        -> (output, BnLink | None).  prev_link: the link returned for the block that produced x.  With the links the backward pass
        folds each block's BatchNorm reductions into the input-gradient kernel of the block behind it (functional.BnLink)."""
        conv = self[0]
        if not (self._has_bn and conv.kernel_size == 3 and self._has_act) or (self._has_bn and self[1].momentum is None):
            return self.forward(x), None
        link = None if self.flatten_output else PF.BnLink()
        y = _conv_bn_act(conv, self[1], nhwc(x), self[-1], self.flatten_output, prev_link, link)
        return (y if self.flatten_output else nchw(y)), link

    def forward(self, x):
        conv = self[0]
        bn = self[1] if self._has_bn else None
        act = self[-1] if self._has_act else None
        if bn is not None:
            y = _conv_bn_act(conv, bn, nhwc(x), act, self.flatten_output)
            return y if self.flatten_output else nchw(y)
        y = _conv_act(conv, nhwc(x), act)
        if self.flatten_output:
            return nchw(y).contiguous()
        return nchw(y)


class ResBlock(nn.Module):
    """x + res_scale * conv(act(conv(x))) (reference model/basic.py:33-52).  The reference's only configuration (bias, no BN,
    ReLU) is one fused autograd node; the other constructor branches (bn=True, bias=False, LeakyReLU) run un-fused on the
    same kernels."""

    def __init__(self, n_feats, kernel_size, bias=True, bn=False, act=nn.ReLU(True), res_scale=1):
        super().__init__()
        _act_slope(act)       # rejects activations the epilogues do not implement
        modules_body = []
        for i in range(2):
            modules_body.append(Conv(n_feats, n_feats, kernel_size, bias=bias))
            if bn:
                modules_body.append(nn.BatchNorm2d(n_feats))
            if i == 0:
                modules_body.append(act)
        self.body = nn.Sequential(*modules_body)
        self.res_scale = res_scale
        self._bn = bn
        # the fused node runs the 3x3 kernels only: any other (odd) kernel size takes the un-fused path, whose _conv_act
        # routes it to the generic kernels (a [C,C,5,5] weight read by a 3x3 packer would be silently wrong)
        self._fused = kernel_size == 3 and bias and not bn and isinstance(act, nn.ReLU) and not isinstance(act, nn.LeakyReLU)

    def forward(self, x):
        if self._fused:
            c1, c2 = self.body[0], self.body[2]
            return nchw(PF.ResBlockFn.apply(nhwc(x), c1.weight, c1.bias, c2.weight, c2.bias, c1.packed, c2.packed,
                                            float(self.res_scale)))
        h = nhwc(x)
        if self._bn:
            c1, b1, act, c2, b2 = self.body
            r = _conv_bn_act(c1, b1, h, act)
            r = _conv_bn_act(c2, b2, r, None)
        else:
            c1, act, c2 = self.body
            r = _conv_act(c1, h, act)
            r = _conv_act(c2, r, None)
        return nchw(PF.ScaleAddFn.apply(r, h, float(self.res_scale)))


class Upsampler(nn.Sequential):
    """conv C->4C, PixelShuffle(2), conv C->4C, PixelShuffle(2), conv C->3 (reference model/basic.py:54-60).
    Both PixelShuffles are fused into the conv epilogue (forward) and the dgrad/wgrad loaders (backward)."""

    def __init__(self, n_feats):
        super().__init__(Conv(n_feats, 4 * n_feats, 3), PixelShuffle(2), Conv(n_feats, 4 * n_feats, 3), PixelShuffle(2),
                         Conv(n_feats, 3, 3))
        self[0].packed = PF.PackedConvWeights(ps=True)
        self[2].packed = PF.PackedConvWeights(ps=True)

    def forward(self, x):
        h = nhwc(x)
        h = PF.conv3x3(h, self[0].weight, self[0].bias, self[0].packed)   # -> [N, 2H, 2W, C]
        h = PF.conv3x3(h, self[2].weight, self[2].bias, self[2].packed)   # -> [N, 4H, 4W, C]
        h = PF.conv3x3(h, self[4].weight, self[4].bias, self[4].packed)   # -> [N, 4H, 4W, 3]
        return nchw(h)
