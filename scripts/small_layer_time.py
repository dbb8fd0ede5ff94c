"""F(4,3) (stacked images, dense LDS layout) vs the direct kernel on the small-spatial layers (VGG conv5_x: 16 x 12 x 12 x 512)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pesr_amd import ops

def timeit(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]

torch.manual_seed(0)
for (N, H, W, Ci, Co) in [(16, 12, 12, 512, 512), (16, 24, 24, 512, 512), (16, 24, 24, 256, 512), (32, 12, 12, 512, 512)]:
    x = torch.rand(N, H, W, Ci, device="cuda") - 0.5
    w = (torch.rand(Co, Ci, 3, 3, device="cuda") - 0.5) * 0.05
    b = torch.rand(Co, device="cuda")
    wd, w4 = ops.pack_conv3x3(w, 0), ops.pack_conv3x3_wino4(w, 0)
    yd = ops.conv3x3_fwd(x, wd, b, Co, act=ops.ACT_RELU); y4 = ops.conv3x3_fwd(x, w4, b, Co, act=ops.ACT_RELU)
    err = (yd - y4).abs().max().item() / yd.abs().max().item()
    td = timeit(lambda: ops.conv3x3_fwd(x, wd, b, Co, act=ops.ACT_RELU)); t4 = timeit(lambda: ops.conv3x3_fwd(x, w4, b, Co, act=ops.ACT_RELU))
    gf = 2 * N * H * W * 9 * Ci * Co / 1e9
    print(f"{N}x{H}x{W} {Ci}->{Co}: direct {td:7.1f} us ({gf / td * 1e3:6.1f} TF/s)   F(4,3) {t4:7.1f} us ({gf / t4 * 1e3:6.1f} algorithmic TF/s)   "
          f"eligible {ops.wino4_eligible(N, H, W, Ci, Co)}  rel diff {err:.2e}")
