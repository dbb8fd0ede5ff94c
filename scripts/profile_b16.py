"""Launch the bf16-mode conv kernels at the G-body shape a few times (for rocprofv3 --pmc passes)."""
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
wb, wbd = ops.pack_conv3x3_bf16(w, 0), ops.pack_conv3x3_bf16(w, 1)
for _ in range(6):
    ops.conv3x3_fwd(x, wb, b, C, act=ops.ACT_RELU)
    ops.conv3x3_dgrad(dy, wbd, (N, H, W, C), mask=x)
    ops.conv3x3_wgrad_bf16(x, dy)
torch.cuda.synchronize()
