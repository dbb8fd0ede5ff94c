"""Before building batch-merged launches (round 4): what ONE launch over 32 images buys against two launches over 16 for
 (a) the forward convs of vgg19 conv4_1 .. conv5_4 (the sr and hr passes could share them behind two batch-16 passes of the large layers),
 (b) the Discriminator's weight gradients (every layer is used twice per backward pass: the second use accumulates today),
 (c) Linear(73728, 1024): weight gradient and input gradient at 32 rows against two calls at 16.
Time only (random data), same box, interleaved medians."""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pesr_amd import functional as PF
from pesr_amd import ops

def med(f, n=7, it=5):
    for _ in range(2): f()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it): f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / it * 1e3)
    return statistics.median(ts)

torch.manual_seed(0)
R = lambda *s: torch.rand(*s, device="cuda") - 0.5
tot = {"a": [0.0, 0.0], "b": [0.0, 0.0], "c": [0.0, 0.0]}
print("(a) vgg19 forward convs, bias + ReLU: 2 x batch 16 vs 1 x batch 32 (us)")
for name, H, Cin, Cout, count in (("conv4_1", 24, 256, 512, 1), ("conv4_2..4", 24, 512, 512, 3), ("conv5_1..4", 12, 512, 512, 4)):
    w = R(Cout, Cin, 3, 3) * 0.05; b = R(Cout)
    x16, x32 = R(16, H, H, Cin), R(32, H, H, Cin)
    cache = PF.PackedConvWeights()
    f16 = lambda: ops.conv3x3_fwd(x16, cache.for_fwd(w, x16.shape), b, Cout, act=ops.ACT_RELU)
    f32_ = lambda: ops.conv3x3_fwd(x32, cache.for_fwd(w, x32.shape), b, Cout, act=ops.ACT_RELU)
    t16, t32 = med(f16), med(f32_)
    print(f"  {name:12s} {H}x{H} {Cin}->{Cout}: 2 x {t16:7.1f} = {2 * t16:7.1f}   1 x {t32:7.1f}   x{count} layers: saves {count * (2 * t16 - t32):7.1f}")
    tot["a"][0] += count * 2 * t16; tot["a"][1] += count * t32
print(f"  total {tot['a'][0]:.0f} -> {tot['a'][1]:.0f} us per step")
print("(b) Discriminator weight gradients: two uses (the second accumulating) vs one launch over 32 images (us)")
for name, H, Cin, Cout, s in (("f.1 s2", 192, 64, 64, 2), ("f.2", 96, 64, 128, 1), ("f.3 s2", 96, 128, 128, 2), ("f.4", 48, 128, 256, 1),
                              ("f.5 s2", 48, 256, 256, 2), ("f.6", 24, 256, 512, 1), ("f.7 s2", 24, 512, 512, 2)):
    OH = (H - 1) // s + 1
    x16, x32, d16, d32 = R(16, H, H, Cin), R(32, H, H, Cin), R(16, OH, OH, Cout), R(32, OH, OH, Cout)
    dw = torch.empty(Cout, Cin, 3, 3, device="cuda")
    def two():
        ops.conv3x3_wgrad(x16, d16, s, want_bias=False, dw_out=dw)
        ops.conv3x3_wgrad(x16, d16, s, want_bias=False, dw_out=dw, accumulate=True)
    one = lambda: ops.conv3x3_wgrad(x32, d32, s, want_bias=False, dw_out=dw)
    t2, t1 = med(two), med(one)
    print(f"  {name:8s} {H}x{H} {Cin}->{Cout}: two uses {t2:7.1f}   one launch {t1:7.1f}   saves {t2 - t1:7.1f}")
    tot["b"][0] += t2; tot["b"][1] += t1
    del x16, x32, d16, d32
print(f"  total {tot['b'][0]:.0f} -> {tot['b'][1]:.0f} us per step")
print("(c) Linear(73728, 1024) (us)")
w = R(1024, 73728) * 0.01
x16, x32, d16, d32 = R(16, 73728), R(32, 73728), R(16, 1024), R(32, 1024)
dw = torch.empty_like(w); db = torch.empty(1024, device="cuda")
def w2():
    ops.linear_wgrad(d16, x16, dw_out=dw, db_out=db)
    ops.linear_wgrad(d16, x16, dw_out=dw, db_out=db, accumulate=True)
t2, t1 = med(w2), med(lambda: ops.linear_wgrad(d32, x32, dw_out=dw, db_out=db))
print(f"  weight gradient: two uses {t2:7.1f}   one call at 32 rows {t1:7.1f}   saves {t2 - t1:7.1f}")
tg2, tg1 = med(lambda: (ops.linear_dgrad(d16, w), ops.linear_dgrad(d16, w))), med(lambda: ops.linear_dgrad(d32, w))
print(f"  input gradient : two calls {tg2:7.1f}   one call at 32 rows {tg1:7.1f}   saves {tg2 - tg1:7.1f}")
tf2, tf1 = med(lambda: (ops.linear_fwd(x16, w, db), ops.linear_fwd(x16, w, db))), med(lambda: ops.linear_fwd(x32, w, db))
print(f"  forward        : two calls {tf2:7.1f}   one call at 32 rows {tf1:7.1f}   saves {tf2 - tf1:7.1f}")
