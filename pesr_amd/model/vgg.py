"""VGG19 features[:35] perceptual extractor (reference model/vgg.py) on the HIP kernels.

The reference loads torchvision's pretrained vgg19.  torchvision and its weight file are third-party
artefacts that are absent offline: pass `weights=<path to a vgg19 state_dict>` to load them, otherwise the
16 convs are initialised with torchvision's own non-pretrained scheme (kaiming_normal_, fan_out; zero
bias) and a warning is printed - topology, slicing and MeanShift constants are identical either way.
The reference's `self.vgg.requires_grad = False` is a no-op that leaves every VGG parameter trainable
but owned by no optimizer; their weight gradients are dead work and are skipped here by
marking the parameters requires_grad=False.
"""
import warnings

import torch
import torch.nn as nn

from .. import functional as PF
from .. import ops
from .basic import Conv, MeanShift, nchw, nhwc

_CFG_E = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M']


class _MaxPool(nn.Module):
    def forward(self, x, relu_in=True):
        return nchw(PF.MaxPoolFn.apply(nhwc(x), relu_in))


def _vgg19_feature_modules():
    layers, c = [], 3
    for v in _CFG_E:
        if v == 'M':
            layers.append(_MaxPool())
        else:
            conv = Conv(c, v, 3)
            nn.init.kaiming_normal_(conv.weight, mode='fan_out', nonlinearity='relu')
            nn.init.constant_(conv.bias, 0)
            layers += [conv, nn.ReLU(True)]
            c = v
    return layers


class VGG(nn.Module):
    def __init__(self, weights=None):
        super().__init__()
        modules = _vgg19_feature_modules()
        self.vgg = nn.Sequential(*modules[:35])  # through conv5_4, before its ReLU
        if weights is not None:
            sd = torch.load(weights, map_location='cpu')
            sd = {k[len('features.'):]: v for k, v in sd.items() if k.startswith('features.')}
            self.vgg.load_state_dict({k: v for k, v in sd.items() if int(k.split('.')[0]) < 35})
        else:
            warnings.warn("pesr_amd VGG: no pretrained vgg19 weights given (weights=...); using random "
                          "kaiming-normal features (perceptual-loss VALUES are then not the reference's)")
        rgb_range = 255
        vgg_mean = (0.485, 0.456, 0.406)
        vgg_std = (0.229 * rgb_range, 0.224 * rgb_range, 0.225 * rgb_range)
        self.sub_mean = MeanShift(rgb_range, vgg_mean, vgg_std)
        for p in self.parameters():
            p.requires_grad = False

    # features[19:35] = conv4_1 .. conv5_4 (+ pool4): the layers behind the third max-pool, small enough (24 x 24 and 12 x 12 at the
    # training shape) that the sr and hr passes run them as ONE batch (forward())
    TAIL_START = 19

    def _steps(self, lo, hi):
        """(kind, module, has_relu) for features[lo:hi], a conv's ReLU folded into it."""
        mods = list(self.vgg)
        out, i = [], lo
        while i < hi:
            m = mods[i]
            if isinstance(m, Conv):
                has_relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                out.append(("conv", m, has_relu))
                i += 2 if has_relu else 1
            else:
                out.append(("pool", m, False))
                i += 1
        return out

    def _run(self, h, lo, hi, prev_relu=False):
        for kind, m, has_relu in self._steps(lo, hi):
            if kind == "conv":
                h = m(h, act=ops.ACT_RELU if has_relu else ops.ACT_NONE, relu_in=prev_relu,
                      relu_grad_by_consumer=has_relu)   # every ReLU here feeds a conv or a pool that masks for it
                prev_relu = has_relu
            else:
                h = m(h, relu_in=prev_relu)
                prev_relu = False
        return h

    def features(self, x):
        """sub_mean -> vgg19.features[:35] of one image batch (reference model/vgg.py:19-22 `_forward`)."""
        return self._run(self.sub_mean(x), 0, len(self.vgg))

    def forward(self, sr, hr):
        """(features(sr), features(hr) without gradient) - reference model/vgg.py:24-26.  The two passes share conv4_1 .. conv5_4 as
        ONE batch (functional.VggTailFn; round 4): behind the third max-pool the feature maps are 24 x 24 and 12 x 12 at the training
        shape, a batch of 16 leaves those launches at 128 - 240 workgroups (split-K, 40 % of the matrix pipe on the 12 x 12 ones), and
        vgg19 has no BatchNorm that would tie samples together: 2076 -> 1768 us per step for these eight layers' forwards
        (profiles/r04_merge_probe.txt).  The input gradient runs on the sr half only.  Not in the bf16 mode (its oracle mirrors the
        per-pass dispatch), not when the tail's weights are trainable, not for unequal shapes."""
        n = len(self.vgg)
        merge = (sr.shape == hr.shape and n > self.TAIL_START and ops.PRECISION == "fp32" and sr.is_cuda
                 and not any(p.requires_grad for p in self.vgg[self.TAIL_START:].parameters()))
        if not merge:
            vgg_sr = self.features(sr)
            with torch.no_grad():
                vgg_hr = self.features(hr.detach())
            return vgg_sr, vgg_hr
        a = self._run(self.sub_mean(sr), 0, self.TAIL_START)
        with torch.no_grad():
            b = self._run(self.sub_mean(hr.detach()), 0, self.TAIL_START)
        fa, fb = PF.VggTailFn.apply(nhwc(a), nhwc(b), self._steps(self.TAIL_START, n))
        return nchw(fa), nchw(fb)
