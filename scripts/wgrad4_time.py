"""Interleaved timing of the three weight-gradient algorithms at the G-body shape (batch 16, 48x48, 256 -> 256)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pesr_amd import ops
N, H, W, C = 16, 48, 48, 256
x = torch.rand(N, H, W, C, device="cuda") - 0.5; dy = torch.rand(N, H, W, C, device="cuda") - 0.5
def t(algo, it=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): ops.conv3x3_wgrad(x, dy, 1, algo=algo)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
names = {ops.WGRAD_AUTO: "F(4,3)", ops.WGRAD_WINO23: "F(2,3)", ops.WGRAD_DIRECT: "direct"}
for a in names: t(a, 3)
res = {a: [] for a in names}
for _ in range(6):
    for a in names: res[a].append(t(a))
for a, v in res.items():
    print(f"wgrad {names[a]} (main kernel + reduce): median {statistics.median(v):.1f} us, min {min(v):.1f} us ({43.487e9 / min(v) / 1e6:.1f} algorithmic TFLOP/s)")
