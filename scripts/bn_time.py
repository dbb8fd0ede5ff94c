"""BatchNorm + LeakyReLU forward / backward kernels at the Discriminator's largest layers: time and HBM rate (forward reads x
twice and writes y: 3 passes over the tensor; backward reads x and dy twice and writes dx: 5 passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pesr_amd import ops

def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]

torch.manual_seed(0)
for (N, H, W, C) in [(16, 96, 96, 64), (16, 96, 96, 128), (16, 48, 48, 256), (16, 12, 12, 512)]:
    x = torch.rand(N, H, W, C, device="cuda") - 0.5
    dy = torch.rand(N, H, W, C, device="cuda") - 0.5
    g, b = torch.rand(C, device="cuda") + 0.5, torch.rand(C, device="cuda") - 0.5
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    nb = torch.zeros((), dtype=torch.long, device="cuda")
    y, mi = ops.bn_lrelu_fwd(x, g, b, rm, rv, nb, 1e-5, 0.1, 0.2)
    mb = x.numel() * 4 / 1e6
    tf = timeit(lambda: ops.bn_lrelu_fwd(x, g, b, rm, rv, nb, 1e-5, 0.1, 0.2))
    tb = timeit(lambda: ops.bn_lrelu_bwd(x, dy, g, b, mi, 0.2))
    print(f"{N}x{H}x{W}x{C} ({mb:5.1f} MB): forward {tf:6.1f} us = {3 * mb / tf:5.2f} TB/s   backward {tb:6.1f} us = {5 * mb / tb:5.2f} TB/s")
