"""Interleaved timing of the weight-gradient algorithms at the G-body shape (batch 16, 48x48, 256 -> 256)."""
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
names = {ops.WGRAD_AUTO: "F(4,3) 32x32x2", ops.WGRAD_WINO4_16X16: "F(4,3) 16x16x4", ops.WGRAD_WINO23: "F(2,3)", ops.WGRAD_DIRECT: "direct"}
for a in names: t(a, 3)
res = {a: [] for a in names}
for _ in range(6):
    for a in names: res[a].append(t(a))
for a, v in res.items():
    print(f"wgrad {names[a]} (main kernel + reduce): median {statistics.median(v):.1f} us, min {min(v):.1f} us ({43.487e9 / min(v) / 1e6:.1f} algorithmic TFLOP/s)")
# agreement of the two F(4,3) forms (same transform, other MFMA shape / summation order)
a, _ = ops.conv3x3_wgrad(x, dy, 1, algo=ops.WGRAD_AUTO); b, _ = ops.conv3x3_wgrad(x, dy, 1, algo=ops.WGRAD_WINO4_16X16)
print("max |16x16x4 - 32x32x2| / max |dw| =", float((a - b).abs().max() / a.abs().max()))
r0 = ops.conv3x3_wgrad(x, dy, 1, algo=ops.WGRAD_AUTO)[0].clone()
print("32x32x2 form, 10 repeats bit-identical:", all(torch.equal(ops.conv3x3_wgrad(x, dy, 1, algo=ops.WGRAD_AUTO)[0], r0) for _ in range(10)))
