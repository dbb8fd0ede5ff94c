"""Generator (EDSR-style) and Discriminator (VGG-style, strided) with the reference's constructor
arguments, attribute names and state_dict schema (reference model/pesr.py), on the HIP kernels."""
import torch.nn as nn

from .. import functional as PF
from .basic import BasicBlock, Conv, MeanShift, ResBlock, Upsampler, nchw, nhwc


class Generator(nn.Module):
    def __init__(self, opt):
        super().__init__()
        n_resblock = opt['depth']
        n_feats = opt['num_channels']
        res_scale = opt['res_scale']
        kernel_size = 3
        rgb_mean = (0.4488, 0.4371, 0.4040)   # DIV2K800 (reference model/pesr.py:13)
        rgb_std = (1.0, 1.0, 1.0)
        rgb_range = 255
        # construction order follows the reference (body first), registration order is set below: a seeded
        # construction draws the RNG in the same sequence and state_dict() lists keys in the same order
        blocks = [ResBlock(n_feats, kernel_size, act=nn.ReLU(True), res_scale=res_scale) for _ in range(n_resblock)]
        blocks.append(Conv(n_feats, n_feats, kernel_size))
        self.sub_mean = MeanShift(rgb_range, rgb_mean, rgb_std)
        self.embed = Conv(3, n_feats, kernel_size)
        self.body = nn.Sequential(*blocks)
        self.upsample = Upsampler(n_feats)
        self.add_mean = MeanShift(rgb_range, rgb_mean, rgb_std, 1)

    def forward(self, x):
        x = self.sub_mean(x)
        x = self.embed(x)
        res = x
        for blk in self.body[:-1]:
            res = blk(res)
        tail = self.body[-1]
        # body tail conv + global skip (`res += x`) in one kernel
        res = nchw(PF.ConvAddFn.apply(nhwc(res), nhwc(x), tail.weight, tail.bias, tail.packed))
        x = self.upsample(res)
        return self.add_mean(x)


class Discriminator(nn.Module):
    def __init__(self, opt):
        super().__init__()
        in_channels = 3
        out_channels = 64
        depth = 7
        act = nn.LeakyReLU(negative_slope=0.2, inplace=True)
        n_colors = 3
        patch_size = opt['patch_size'] * 4
        sn = opt['spectral_norm']

        m_features = [BasicBlock(n_colors, out_channels, 3, bn=True, act=act, sn=sn)]
        for i in range(depth):
            in_channels = out_channels
            if i % 2 == 1:
                stride = 1
                out_channels *= 2
            else:
                stride = 2
            m_features.append(BasicBlock(in_channels, out_channels, 3, stride=stride, bn=True, act=act, sn=sn))
        self.features = nn.Sequential(*m_features)
        self.features[-1].flatten_output = True   # NCHW-contiguous so that .view(B, -1) has the reference's column order

        patch_size = patch_size // (2 ** ((depth + 1) // 2))
        self.classifier = nn.Sequential(nn.Linear(out_channels * patch_size ** 2, 1024), act, nn.Linear(1024, 1))

    def forward(self, x):
        features = self.features(x)
        flat = features.view(features.size(0), -1)
        fc1, act, fc2 = self.classifier[0], self.classifier[1], self.classifier[2]
        h = PF.LinearFn.apply(flat, fc1.weight, fc1.bias, PF.ops.ACT_LRELU, act.negative_slope)
        return PF.LinearFn.apply(h, fc2.weight, fc2.bias, PF.ops.ACT_NONE, 0.0)
