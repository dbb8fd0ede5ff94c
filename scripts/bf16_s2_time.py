"""Stride-2 forward and input gradient at the Discriminator's four down-sampling layers: fp32 direct kernel vs the stride-2 forms of the bf16 kernel."""
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

for (N, H, W, Ci, Co) in [(16, 192, 192, 64, 64), (16, 96, 96, 128, 128), (16, 48, 48, 256, 256), (16, 24, 24, 512, 512)]:
    x = torch.rand(N, H, W, Ci, device="cuda") - 0.5
    w = (torch.rand(Co, Ci, 3, 3, device="cuda") - 0.5) * 0.1
    wp, wb = ops.pack_conv3x3(w, 0), ops.pack_conv3x3_bf16(w, 0)
    a = t(lambda: ops.conv3x3_fwd(x, wp, None, Co, 2))
    b = t(lambda: ops.conv3x3_fwd(x, wb, None, Co, 2))
    gf = 2.0 * N * (H // 2) * (W // 2) * Ci * Co * 9 / 1e9
    d = (ops.conv3x3_fwd(x, wp, None, Co, 2) - ops.conv3x3_fwd(x, wb, None, Co, 2)).abs().max().item()
    print(f"s2 fwd   {N}x{H}x{W}x{Ci}->{Co}: fp32 {a:7.1f} us   bf16 {b:7.1f} us   max|diff| {d:.2e}")
    dy = torch.rand(N, H // 2, W // 2, Co, device="cuda") - 0.5
    wpd, wbd = ops.pack_conv3x3(w, 1), ops.pack_conv3x3_bf16(w, 1)
    a = t(lambda: ops.conv3x3_dgrad(dy, wpd, (N, H, W, Ci), 2))
    b = t(lambda: ops.conv3x3_dgrad(dy, wbd, (N, H, W, Ci), 2))
    d = (ops.conv3x3_dgrad(dy, wpd, (N, H, W, Ci), 2) - ops.conv3x3_dgrad(dy, wbd, (N, H, W, Ci), 2)).abs().max().item()
    print(f"s2 dgrad {N}x{H}x{W}x{Ci}<-{Co}: fp32 {a:7.1f} us   bf16 {b:7.1f} us   max|diff| {d:.2e}")
