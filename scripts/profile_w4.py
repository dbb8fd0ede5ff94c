"""Launch the F(4,3) conv kernel at the G-body shape a few times (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pesr_amd import ops
torch.manual_seed(0)
N, H, W, C = 16, 48, 48, 256
x = torch.rand(N, H, W, C, device="cuda") - 0.5
w = (torch.rand(C, C, 3, 3, device="cuda") - 0.5) * 0.1
b = torch.rand(C, device="cuda")
dy = torch.rand(N, H, W, C, device="cuda") - 0.5
w4, w4d = ops.pack_conv3x3_wino4(w, 0), ops.pack_conv3x3_wino4(w, 1)
for _ in range(6):
    ops.conv3x3_fwd(x, w4, b, C, act=ops.ACT_RELU)
    ops.conv3x3_dgrad(dy, w4d, (N, H, W, C), mask=x)
    ops.conv3x3_wgrad(x, dy)
torch.cuda.synchronize()
