"""Winograd F(4,3)-along-x conv (conv3x3_wino4.hip) vs torch conv2d in fp64 on the CPU, next to the F(2,3) and direct kernels:
accuracy table and interleaved timing at the G-body shape.   python scripts/wino4_test.py [quick]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from pesr_amd import ops

def nhwc(t): return t.permute(0, 2, 3, 1).contiguous().cuda()
def nchw(t): return t.permute(0, 3, 1, 2).cpu()
def rel(a, ref): return ((a.double() - ref).abs().max() / ref.abs().max()).item()

torch.manual_seed(0)
worst = 0.0
shapes = [(1, 12, 48, 16, 64), (2, 7, 12, 32, 64), (1, 48, 48, 256, 256), (1, 13, 96, 64, 128), (2, 5, 4, 16, 64), (1, 9, 196, 16, 64),
          (3, 24, 24, 256, 128), (1, 30, 20, 48, 192)]
for (N, H, W, Cin, Cout) in shapes:
    x = torch.rand(N, Cin, H, W) - 0.5; w = (torch.rand(Cout, Cin, 3, 3) - 0.5) * 0.1; b = torch.rand(Cout) - 0.5
    skip = torch.rand(N, Cout, H, W) - 0.5; mk = torch.rand(N, Cout, H, W) - 0.5
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    w4 = ops.pack_conv3x3_wino4(w.cuda(), 0)
    y = ops.conv3x3_fwd(nhwc(x), w4, b.cuda(), Cout)
    e4 = rel(nchw(y), ref)
    e0 = rel(nchw(ops.conv3x3_fwd(nhwc(x), ops.pack_conv3x3(w.cuda(), 0), b.cuda(), Cout)), ref)
    e2 = rel(nchw(ops.conv3x3_fwd(nhwc(x), ops.pack_conv3x3_wino(w.cuda(), 0), b.cuda(), Cout)), ref) if W % 2 == 0 and Cout % 128 == 0 else float("nan")
    ref2 = torch.relu(torch.where(mk.double() > 0, ref * 0.1, torch.zeros_like(ref)) + skip.double())
    y2 = ops.conv3x3_fwd(nhwc(x), w4, b.cuda(), Cout, alpha=0.1, act=ops.ACT_RELU, skip=nhwc(skip), mask=nhwc(mk))
    ee = rel(nchw(y2), ref2)
    e3 = float("nan")
    if Cin % 64 == 0:
        dy = torch.rand(N, Cout, H, W) - 0.5
        dref = F.conv_transpose2d(dy.double(), w.double(), padding=1)
        dx = ops.conv3x3_dgrad(nhwc(dy), ops.pack_conv3x3_wino4(w.cuda(), 1), (N, H, W, Cin))
        e3 = rel(nchw(dx), dref)
    print(f"{N}x{H}x{W} {Cin}->{Cout}: rel err F(4,3) {e4:.2e} | F(2,3) {e2:.2e} | direct {e0:.2e} | fused epilogue {ee:.2e} | dgrad {e3:.2e}", flush=True)
    worst = max(worst, e4, ee, 0.0 if e3 != e3 else e3)
# fused PixelShuffle store / pixel-unshuffle load (the upsampler convs)
N, H, W, C = 2, 12, 24, 64
x = torch.rand(N, C, H, W) - 0.5; w = (torch.rand(4 * C, C, 3, 3) - 0.5) * 0.1; b = torch.rand(4 * C) - 0.5
ref = F.pixel_shuffle(F.conv2d(x.double(), w.double(), b.double(), padding=1), 2)
y = ops.conv3x3_fwd(nhwc(x), ops.pack_conv3x3_wino4(w.cuda(), 0, ps=True), ops.pack_bias_ps(b.cuda()), 4 * C, ps_out=True)
e_ps = rel(nchw(y), ref)
dys = torch.rand(N, C, 2 * H, 2 * W) - 0.5
dref = F.conv_transpose2d(F.pixel_unshuffle(dys.double(), 2), w.double(), padding=1)
dx = ops.conv3x3_dgrad(nhwc(dys), ops.pack_conv3x3_wino4(w.cuda(), 1, ps=True), (N, H, W, C), ps_in=True)
e_pi = rel(nchw(dx), dref)
print(f"fused PixelShuffle store {e_ps:.2e}, fused pixel-unshuffle load (dgrad) {e_pi:.2e}")
worst = max(worst, e_ps, e_pi)
print("worst", worst)
assert worst < 1e-5

if len(sys.argv) > 1 and sys.argv[1] == "quick":
    sys.exit(0)
# timing, K1 shape, interleaved
N, H, W, C = 16, 48, 48, 256
x = torch.rand(N, H, W, C, device="cuda") - 0.5; w = (torch.rand(C, C, 3, 3, device="cuda") - 0.5) * 0.1; b = torch.rand(C, device="cuda")
sk = torch.rand(N, H, W, C, device="cuda")
packs = {"direct": ops.pack_conv3x3(w, 0), "F(2,3)": ops.pack_conv3x3_wino(w, 0), "F(4,3)": ops.pack_conv3x3_wino4(w, 0)}
def t(fn, it=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
res = {k: [] for k in packs}
for k, p in packs.items():
    for _ in range(3): ops.conv3x3_fwd(x, p, b, C, act=ops.ACT_RELU)
for rnd in range(7):
    for k, p in packs.items():
        res[k].append(t(lambda: ops.conv3x3_fwd(x, p, b, C, alpha=0.1, skip=sk)))
for k, v in res.items():
    v.sort()
    print(f"K1 shape fwd (scale+skip epilogue) {k}: median {v[len(v)//2]:.1f} us, min {v[0]:.1f} us  ({43.487e9 / v[0] / 1e6:.1f} algorithmic TFLOP/s)")
