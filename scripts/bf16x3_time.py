"""Split-bf16 forward / input-gradient kernel against the fp32 F(4,3) kernel on the step's layer shapes (same box, interleaved
medians; rotating inputs), with the error of both against a float64 conv on one image:   python scripts/bf16x3_time.py"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pesr_amd import ops

def med(f, n=6, it=10):
    for _ in range(2): f()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it): f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / it * 1e3)
    return statistics.median(ts)

SHAPES = [("G body 256->256 @48", 16, 48, 48, 256, 256, False), ("upsample.0 256->1024 @48 ps", 16, 48, 48, 256, 1024, True),
          ("upsample.2 256->1024 @96 ps", 16, 96, 96, 256, 1024, True), ("vgg conv2_2 128->128 @96", 16, 96, 96, 128, 128, False),
          ("vgg conv3_1 128->256 @48", 16, 48, 48, 128, 256, False), ("vgg conv4_2 512->512 @24 (N=32)", 32, 24, 24, 512, 512, False)]
torch.manual_seed(0)
for name, N, H, W, Cin, Cout, ps in SHAPES:
    xs = [torch.rand(N, H, W, Cin, device="cuda") - 0.5 for _ in range(3)]
    w = (torch.rand(Cout, Cin, 3, 3, device="cuda") - 0.5) * 0.1
    b = torch.rand(Cout, device="cuda")
    w4, w3 = ops.pack_conv3x3_wino4(w, 0, ps), ops.pack_conv3x3_bf16x3(w, 0, ps)
    bp = ops.pack_bias_ps(b) if ps else b
    k = [0]
    def run(wp):
        k[0] += 1
        return ops.conv3x3_fwd(xs[k[0] % 3], wp, bp, Cout, act=ops.ACT_NONE if ps else ops.ACT_RELU, ps_out=ps)
    t4, t3 = med(lambda: run(w4)), med(lambda: run(w3))
    gf = 18.0 * N * H * W * Cin * Cout / 1e9
    # error vs fp64 on the first image (no activation)
    x1 = xs[0][:1]
    ref = torch.nn.functional.conv2d(x1.permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=1)
    if ps: ref = torch.nn.functional.pixel_shuffle(ref, 2)
    ref = ref.permute(0, 2, 3, 1)
    e4 = float((ops.conv3x3_fwd(x1, w4, bp, Cout, ps_out=ps).double() - ref).abs().max() / ref.abs().max())
    e3 = float((ops.conv3x3_fwd(x1, w3, bp, Cout, ps_out=ps).double() - ref).abs().max() / ref.abs().max())
    print(f"{name:34s} F(4,3) fp32 {t4:8.1f} us ({gf / t4 * 1e3:6.1f} alg TF/s, err {e4:.1e})   split-bf16 {t3:8.1f} us ({gf / t3 * 1e3:6.1f} alg TF/s = "
          f"{3 * gf / t3 * 1e3:6.1f} issued, err {e3:.1e})   x{t4 / t3:.2f}")
