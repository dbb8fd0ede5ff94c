"""Functional CPU restatement of the reference networks over plain state dicts - TEST INFRASTRUCTURE.

Each network is a pure function `forward(state_dict, x, ...)`; the state dict uses exactly the
reference's key/shape schema ("Checkpoint / resume"), so a reference checkpoint, the
product modules' `state_dict()` and these functions are interchangeable.  Arithmetic follows
reference model/basic.py, model/pesr.py, model/vgg.py line by line (cited per function); it is pinned
against outputs of the imported reference by tests/test_oracle_golden.py.
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F

# torchvision vgg19 cfg "E"; the reference keeps features[:35] = through conv5_4, pre-ReLU (model/vgg.py:8-10)
VGG_CFG_E = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]
G_RGB_MEAN = (0.4488, 0.4371, 0.4040)  # model/pesr.py:13
VGG_MEAN = (0.485, 0.456, 0.406)       # model/vgg.py:13
VGG_STD = (0.229, 0.224, 0.225)        # model/vgg.py:14 (times rgb_range 255)


# ------------------------------------------------------------------------------------------------
# shapes / fixed constants
# ------------------------------------------------------------------------------------------------
def meanshift_params(rgb_range, mean, std, sign):
    """model/basic.py:9-17: 1x1 conv, weight = I/std, bias = sign*rgb_range*mean/std (fp32 arithmetic order kept)."""
    std_t = torch.tensor(std, dtype=torch.float32)
    w = torch.eye(3).view(3, 3, 1, 1) / std_t.view(3, 1, 1, 1)
    b = sign * rgb_range * torch.tensor(mean, dtype=torch.float32)
    b = b / std_t
    return w, b


def generator_shapes(num_channels, depth):
    """model/pesr.py:18-26 in state_dict (attribute-assignment) order."""
    C = num_channels
    s = OrderedDict()
    s["sub_mean.weight"], s["sub_mean.bias"] = (3, 3, 1, 1), (3,)
    s["embed.weight"], s["embed.bias"] = (C, 3, 3, 3), (C,)
    for i in range(depth):
        for j in (0, 2):
            s[f"body.{i}.body.{j}.weight"], s[f"body.{i}.body.{j}.bias"] = (C, C, 3, 3), (C,)
    s[f"body.{depth}.weight"], s[f"body.{depth}.bias"] = (C, C, 3, 3), (C,)
    s["upsample.0.weight"], s["upsample.0.bias"] = (4 * C, C, 3, 3), (4 * C,)
    s["upsample.2.weight"], s["upsample.2.bias"] = (4 * C, C, 3, 3), (4 * C,)
    s["upsample.4.weight"], s["upsample.4.bias"] = (3, C, 3, 3), (3,)
    s["add_mean.weight"], s["add_mean.bias"] = (3, 3, 1, 1), (3,)
    return s


def discriminator_plan():
    """model/pesr.py:53-65: (cin, cout, stride) of the 8 BasicBlocks."""
    plan = [(3, 64, 1)]
    cin = cout = 64
    for i in range(7):
        cin = cout
        if i % 2 == 1:
            stride, cout = 1, cout * 2
        else:
            stride = 2
        plan.append((cin, cout, stride))
    return plan


def discriminator_shapes(patch_size, spectral_norm=False):
    """model/pesr.py:41-75 (patch_size is the LR size; the classifier sees 4*ps/16 squared x 512).  spectral_norm: the conv
    entries follow torch.nn.utils.spectral_norm's schema (weight_orig, weight_u [Cout], weight_v [Cin*9])."""
    s = OrderedDict()
    for i, (cin, cout, _) in enumerate(discriminator_plan()):
        if spectral_norm:
            s[f"features.{i}.0.weight_orig"] = (cout, cin, 3, 3)
            s[f"features.{i}.0.weight_u"], s[f"features.{i}.0.weight_v"] = (cout,), (cin * 9,)
        else:
            s[f"features.{i}.0.weight"] = (cout, cin, 3, 3)
        s[f"features.{i}.1.weight"], s[f"features.{i}.1.bias"] = (cout,), (cout,)
        s[f"features.{i}.1.running_mean"], s[f"features.{i}.1.running_var"] = (cout,), (cout,)
        s[f"features.{i}.1.num_batches_tracked"] = ()
    side = (patch_size * 4) // 16
    s["classifier.0.weight"], s["classifier.0.bias"] = (1024, 512 * side * side), (1024,)
    s["classifier.2.weight"], s["classifier.2.bias"] = (1, 1024), (1,)
    return s


def vgg_conv_indices():
    idx, out, c = 0, [], 3
    for v in VGG_CFG_E:
        if v == "M":
            idx += 1
        else:
            out.append((idx, c, v))
            c = v
            idx += 2
    return [(i, ci, co) for (i, ci, co) in out if i < 35]


def vgg_shapes():
    """model/vgg.py:8-15: `vgg.<idx>.weight/bias` for the 16 convs below index 35, then sub_mean."""
    s = OrderedDict()
    for i, ci, co in vgg_conv_indices():
        s[f"vgg.{i}.weight"], s[f"vgg.{i}.bias"] = (co, ci, 3, 3), (co,)
    s["sub_mean.weight"], s["sub_mean.bias"] = (3, 3, 1, 1), (3,)
    return s


def set_meanshift(sd, which):
    """Fill the MeanShift entries of a generator ('G') or VGG ('V') state dict with their fixed values."""
    if which == "G":
        sd["sub_mean.weight"], sd["sub_mean.bias"] = meanshift_params(255, G_RGB_MEAN, (1.0, 1.0, 1.0), -1)
        sd["add_mean.weight"], sd["add_mean.bias"] = meanshift_params(255, G_RGB_MEAN, (1.0, 1.0, 1.0), 1)
    else:
        std = tuple(s * 255 for s in VGG_STD)
        sd["sub_mean.weight"], sd["sub_mean.bias"] = meanshift_params(255, VGG_MEAN, std, -1)
    return sd


# ------------------------------------------------------------------------------------------------
# forwards
# ------------------------------------------------------------------------------------------------
def _conv3(x, w, b, stride=1, ps=False):
    """nn.Conv2d(k=3, padding=1) of model/basic.py:4-7.  (With oracle.bf16.enabled() - the build's optional bf16 mode, no
    counterpart in the reference - the convs that mode covers round their operands to bf16; `ps` marks the convs in front of
    nn.PixelShuffle, whose fused kernels have their own shape rules.)"""
    from . import bf16
    if bf16.ON:
        return bf16.conv3x3(x, w, b, stride, ps)
    return F.conv2d(x, w, b, stride=stride, padding=1)


def generator_forward(sd, x, depth, res_scale):
    """model/pesr.py:28-38 with ResBlock model/basic.py:48-52 and Upsampler model/basic.py:54-60."""
    x = F.conv2d(x, sd["sub_mean.weight"], sd["sub_mean.bias"])
    x = _conv3(x, sd["embed.weight"], sd["embed.bias"])
    h = x
    for i in range(depth):
        r = _conv3(h, sd[f"body.{i}.body.0.weight"], sd[f"body.{i}.body.0.bias"])
        r = F.relu(r)
        r = _conv3(r, sd[f"body.{i}.body.2.weight"], sd[f"body.{i}.body.2.bias"])
        h = r.mul(res_scale) + h                     # basic.py:49-50
    h = _conv3(h, sd[f"body.{depth}.weight"], sd[f"body.{depth}.bias"])
    h = h + x                                        # pesr.py:33
    h = _conv3(h, sd["upsample.0.weight"], sd["upsample.0.bias"], ps=True)
    h = F.pixel_shuffle(h, 2)
    h = _conv3(h, sd["upsample.2.weight"], sd["upsample.2.bias"], ps=True)
    h = F.pixel_shuffle(h, 2)
    h = _conv3(h, sd["upsample.4.weight"], sd["upsample.4.bias"])
    return F.conv2d(h, sd["add_mean.weight"], sd["add_mean.bias"])


def spectral_normalize(w, u, v, training, eps=1e-12):
    """torch.nn.utils.spectral_norm's compute_weight with n_power_iterations = 1 - what reference model/basic.py:25 evidently
    means by its (undefined) `spectral_norm`: in training mode v <- normalize(W^T u), u <- normalize(W v) IN PLACE under
    no_grad, then sigma = u^T W v on clones of u, v (constants for autograd) and w / sigma.  Pinned against torch's own
    implementation by tests/test_oracle_golden.py."""
    wm = w.reshape(w.shape[0], -1)
    if training:
        with torch.no_grad():
            v.copy_(F.normalize(torch.mv(wm.t(), u), dim=0, eps=eps))
            u.copy_(F.normalize(torch.mv(wm, v), dim=0, eps=eps))
    uc, vc = u.clone(), v.clone()
    sigma = torch.dot(uc, torch.mv(wm, vc))
    return w / sigma


def discriminator_forward(sd, x, update_running_stats=True):
    """model/pesr.py:77-81; every BasicBlock = conv(no bias) -> BatchNorm2d in TRAINING mode -> LeakyReLU(0.2)
    (model/basic.py:26-30; D is never put in eval mode).  Running stats are updated in place in
    `sd` when asked, as nn.BatchNorm2d would (momentum 0.1, unbiased variance)."""
    h = x
    for i, (_, _, stride) in enumerate(discriminator_plan()):
        if f"features.{i}.0.weight_orig" in sd:        # --spectral_norm true: torch's spectral_norm state_dict schema
            w = spectral_normalize(sd[f"features.{i}.0.weight_orig"], sd[f"features.{i}.0.weight_u"], sd[f"features.{i}.0.weight_v"], True)
        else:
            w = sd[f"features.{i}.0.weight"]
        h = _conv3(h, w, None, stride=stride)
        rm, rv = sd[f"features.{i}.1.running_mean"], sd[f"features.{i}.1.running_var"]
        if update_running_stats:
            h = F.batch_norm(h, rm, rv, sd[f"features.{i}.1.weight"], sd[f"features.{i}.1.bias"], True, 0.1, 1e-5)
            sd[f"features.{i}.1.num_batches_tracked"] += 1
        else:
            h = F.batch_norm(h, None, None, sd[f"features.{i}.1.weight"], sd[f"features.{i}.1.bias"], True, 0.1, 1e-5)
        h = F.leaky_relu(h, 0.2)
    h = h.reshape(h.size(0), -1)                     # NCHW flatten: index = c*side*side + y*side + x
    h = F.leaky_relu(F.linear(h, sd["classifier.0.weight"], sd["classifier.0.bias"]), 0.2)
    return F.linear(h, sd["classifier.2.weight"], sd["classifier.2.bias"])


def vgg_features(sd, x):
    """model/vgg.py:19-22: sub_mean -> vgg19.features[:35]."""
    h = F.conv2d(x, sd["sub_mean.weight"], sd["sub_mean.bias"])
    idx = 0
    for v in VGG_CFG_E:
        if idx >= 35:
            break
        if v == "M":
            h = F.max_pool2d(h, 2, 2)
            idx += 1
        else:
            h = _conv3(h, sd[f"vgg.{idx}.weight"], sd[f"vgg.{idx}.bias"])
            idx += 1
            if idx < 35:
                h = F.relu(h)
            idx += 1
    return h


def vgg_forward(sd, sr, hr):
    """model/vgg.py:18-28: features of sr with graph, of hr under no_grad on a detached input."""
    f_sr = vgg_features(sd, sr)
    with torch.no_grad():
        f_hr = vgg_features(sd, hr.detach())
    return f_sr, f_hr
