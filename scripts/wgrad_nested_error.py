"""Error of the three forms of the transposed-Winograd weight gradient against an fp64 oracle at the full G-body shape (16 x 48 x 48, 256 -> 256),
uniform and post-ReLU inputs:  python scripts/wgrad_nested_error.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from pesr_amd import ops
torch.manual_seed(0)
N, H, W, C = 16, 48, 48, 256
for kind in ("uniform", "relu"):
    x = torch.rand(N, C, H, W, dtype=torch.float64) * 2 - 1
    if kind == "relu":
        x = x.clamp(min=0) * 2
    dy = torch.rand(N, C, H, W, dtype=torch.float64) * 2 - 1
    ref = torch.nn.grad.conv2d_weight(x, (C, C, 3, 3), dy, padding=1)
    xg = x.float().permute(0, 2, 3, 1).contiguous().cuda(); dg = dy.float().permute(0, 2, 3, 1).contiguous().cuda()
    ref32 = torch.nn.grad.conv2d_weight(x.float(), (C, C, 3, 3), dy.float(), padding=1).double()
    out = [f"{kind:8s} CPU fp32 {float((ref32 - ref).abs().max() / ref.abs().max()):.2e}"]
    for name, algo in (("direct", ops.WGRAD_DIRECT), ("16x16x4 1-D", ops.WGRAD_WINO4_16X16), ("32x32x2 1-D", ops.WGRAD_WINO4_1D), ("32x32x2 y-nested", ops.WGRAD_AUTO)):
        dw, _ = ops.conv3x3_wgrad(xg, dg, 1, want_bias=False, algo=algo)
        out.append(f"{name} {float((dw.cpu().double() - ref).abs().max() / ref.abs().max()):.2e}")
    print("  |  ".join(out))
