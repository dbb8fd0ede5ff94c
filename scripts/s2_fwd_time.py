"""Time the four stride-2 forward convs of the Discriminator (direct kernel), for library A/B runs: PESR_HIP_LIB=exp/libX.so python scripts/s2_fwd_time.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pesr_amd import ops
torch.manual_seed(0)
for (H, C) in ((192, 64), (96, 128), (48, 256), (24, 512)):
    x = torch.rand(16, H, H, C, device="cuda") - 0.5
    w = (torch.rand(C, C, 3, 3, device="cuda") - 0.5) * 0.1
    wp = ops.pack_conv3x3(w, 0)
    for _ in range(3): ops.conv3x3_fwd(x, wp, None, C, stride=2)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.conv3x3_fwd(x, wp, None, C, stride=2)
    e1.record(); torch.cuda.synchronize()
    print(f"s2 fwd {C}->{C} @{H}: {e0.elapsed_time(e1)/20*1e3:.1f} us")
