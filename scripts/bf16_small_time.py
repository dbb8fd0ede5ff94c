"""The small 512-channel layers (12^2, 24^2 at batch 16) on the F(4,3) fp32 kernel vs the bf16 kernel (whatever the workgroup-count rule says)."""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pesr_amd import ops
def t(fn, iters=20):
    for _ in range(3): fn()
    r = []
    for _ in range(5):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        r.append(e0.elapsed_time(e1) / iters * 1e3)
    return statistics.median(r)
for (N, H, W, Ci, Co) in [(16, 12, 12, 512, 512), (16, 24, 24, 512, 256), (16, 24, 24, 256, 512), (16, 24, 24, 512, 512), (16, 48, 48, 256, 128)]:
    x = torch.rand(N, H, W, Ci, device="cuda") - 0.5
    w = (torch.rand(Co, Ci, 3, 3, device="cuda") - 0.5) * 0.1
    w4 = ops.pack_conv3x3_wino4(w, 0) if ops.wino4_eligible(N, H, W, Ci, Co) else ops.pack_conv3x3(w, 0)
    wb = ops.pack_conv3x3_bf16(w, 0)
    a = t(lambda: ops.conv3x3_fwd(x, w4, None, Co, 1)); b = t(lambda: ops.conv3x3_fwd(x, wb, None, Co, 1))
    print(f"fwd {N}x{H}x{W}x{Ci}->{Co}: fp32 {type(w4).__name__:12s} {a:7.1f} us   bf16 {b:7.1f} us")
