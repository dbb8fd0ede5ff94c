"""Launch the stride-2 conv kernels (forward, input gradient, weight gradient) at the Discriminator's four layer shapes a few
times (for the rocprofv3 --pmc passes of scripts/gpu_job.sh pmc: matrix-pipe share, LDS bank conflicts, waits, HBM traffic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pesr_amd import ops
torch.manual_seed(0)
for (N, H, W, Ci, Co) in [(16, 192, 192, 64, 64), (16, 96, 96, 128, 128), (16, 48, 48, 256, 256), (16, 24, 24, 512, 512)]:
    x = torch.rand(N, H, W, Ci, device="cuda") - 0.5
    w = (torch.rand(Co, Ci, 3, 3, device="cuda") - 0.5) * 0.1
    dy = torch.rand(N, H // 2, W // 2, Co, device="cuda") - 0.5
    wf, wd = ops.pack_conv3x3(w, 0), ops.pack_conv3x3(w, 1)
    for _ in range(4):
        ops.conv3x3_fwd(x, wf, None, Co, stride=2)
        ops.conv3x3_dgrad(dy, wd, (N, H, W, Ci), stride=2)
        ops.conv3x3_wgrad(x, dy, 2, want_bias=False)
torch.cuda.synchronize()
