"""Per-op, per-shape GPU time of one GAN step: every pesr_amd.ops call is bracketed by events on one stream
(PESR_SIDE_STREAM=0), then grouped by (op, tensor shapes).  Conv rows also print TFLOP/s against the fp32-MFMA peak."""
import os, sys, warnings, collections
os.environ["PESR_SIDE_STREAM"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from model import Generator, Discriminator, VGG
from pesr_amd import ops
from pesr_amd.optim import FlatAdam
from pesr_amd.step import Trainer

REC = []
ON = [False]
def wrap(name):
    f = getattr(ops, name)
    def g(*a, **k):
        if not ON[0]:
            return f(*a, **k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = f(*a, **k); e1.record()
        shapes = tuple(tuple(t.shape) for t in a if isinstance(t, torch.Tensor))
        extra = tuple((kk, v) for kk, v in sorted(k.items()) if isinstance(v, (int, bool)) and kk in ("stride", "ps_out", "ps_in", "cout"))
        extra += tuple(x for x in a if isinstance(x, (int, bool)) and not isinstance(x, torch.Tensor))[:2]
        if name == "conv3x3_dgrad":
            extra += (("in", tuple(a[2]) if len(a) > 2 else tuple(k["in_shape"])),)
        REC.append((name, shapes, extra, e0, e1))
        return r
    setattr(ops, name, g)
for n in ["conv3x3_fwd", "conv3x3_dgrad", "conv3x3_wgrad", "conv3x3_wgrad_rgb", "conv3x3_rgb_dgrad", "bn_lrelu_fwd", "bn_lrelu_bwd", "linear_fwd", "linear_dgrad",
          "linear_wgrad", "maxpool2x2_fwd", "maxpool2x2_bwd", "meanshift_fwd", "meanshift_bwd", "relu_mask", "loss_l1_tv", "loss_mse",
          "adam_step", "pack_conv3x3", "pack_conv3x3_wino", "pixel_shuffle_fwd", "pixel_shuffle_bwd"]:
    wrap(n)

torch.manual_seed(0)
dev = torch.device("cuda")
opt = {"patch_size": 48, "num_channels": 256, "depth": 32, "res_scale": 0.1, "spectral_norm": False}
G, D = Generator(opt).to(dev), Discriminator(opt).to(dev)
with warnings.catch_warnings():
    warnings.simplefilter("ignore"); V = VGG().to(dev)
oG, oD = FlatAdam([p for p in G.parameters() if p.requires_grad], lr=5e-7), FlatAdam(D.parameters(), lr=5e-7)   # (bench.py --lr: keeps D's loss O(1))
tr = Trainer(G, D, V, oG, oD, gan_type="RSGAN", focal_loss=True, fl_gamma=1.0, alpha_vgg=50, alpha_gan=1, alpha_tv=1e-6, alpha_l1=0)
lr = torch.randint(0, 256, (16, 3, 48, 48)).float().to(dev)
hr = torch.randint(0, 256, (16, 3, 192, 192)).float().to(dev).contiguous(memory_format=torch.channels_last)
for _ in range(3): tr.gan_step(lr, hr)
torch.cuda.synchronize()
ON[0] = True
R = 3
e_a, e_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e_a.record()
for _ in range(R): tr.gan_step(lr, hr)
e_b.record(); torch.cuda.synchronize()
agg = collections.OrderedDict()
for name, shapes, extra, e0, e1 in REC:
    key = (name, shapes[:2], extra)
    t, c = agg.get(key, (0.0, 0))
    agg[key] = (t + e0.elapsed_time(e1), c + 1)
def conv_flops(name, shapes, extra):
    """ALGORITHMIC flops of a conv op: 2 * N * OH * OW * Cin * Cout * 9 with OH x OW the OUTPUT size of the forward conv (round 2
    used the input size: every stride-2 row was overstated 4 x).  `extra` = the call's keyword ints as (name, value) pairs, then
    its first two positional ints: (cout, stride) for conv3x3_fwd, (stride,) for conv3x3_dgrad / conv3x3_wgrad."""
    try:
        d = dict(e for e in extra if isinstance(e, tuple) and len(e) == 2 and isinstance(e[0], str))
        ints = [x for x in extra if isinstance(x, int) and not isinstance(x, bool)]
        if name == "conv3x3_fwd":
            N, H, W, Ci = shapes[0]
            co = d.get("cout") or ints[0]
            s = d.get("stride") or (ints[1] if len(ints) > 1 else 1)
        elif name == "conv3x3_dgrad":
            N, H, W, Ci = d["in"]
            s = d.get("stride") or (ints[0] if ints else 1)
            co = shapes[0][3] * (4 if d.get("ps_in") else 1)
        elif name == "conv3x3_wgrad":
            N, H, W, Ci = shapes[0]
            s = d.get("stride") or (ints[0] if ints else 1)
            co = shapes[1][3] * (4 if d.get("ps_in") else 1)
        else:
            return 0.0
        return 2.0 * N * ((H - 1) // s + 1) * ((W - 1) // s + 1) * Ci * co * 9
    except Exception:
        return 0.0
rows = []
for (name, shapes, extra), (t, c) in agg.items():
    fl = conv_flops(name, shapes, extra)
    rows.append((t / R, c / R, name, shapes, extra, fl))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"step {e_a.elapsed_time(e_b) / R:.2f} ms; sum of bracketed ops {tot:.2f} ms")
print(f"{'ms/step':>8} {'calls':>5} {'us/call':>8} {'TF/s':>6} {'eff':>5} {'lost ms':>7}  op / shapes")
for t, c, name, shapes, extra, fl in rows[:70]:
    us = t / c * 1e3
    tf = fl / (us * 1e-6) / 1e12 if fl else 0.0
    lost = t - c * fl / 157.3e12 * 1e3 if fl else 0.0
    print(f"{t:8.3f} {c:5.0f} {us:8.1f} {tf:6.1f} {tf / 1.573:5.1f} {lost:7.2f}  {name} {shapes} {extra}")
