"""Launch the stride-2 forward convs of the Discriminator a few times (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pesr_amd import ops
torch.manual_seed(0)
for (H, C) in ((192, 64), (96, 128), (48, 256), (24, 512)):
    x = torch.rand(16, H, H, C, device="cuda") - 0.5
    w = (torch.rand(C, C, 3, 3, device="cuda") - 0.5) * 0.1
    wp = ops.pack_conv3x3(w, 0)
    for _ in range(4):
        ops.conv3x3_fwd(x, wp, None, C, stride=2)
torch.cuda.synchronize()
