"""The C -> 3 forward kernel at the Generator's last conv (16 x 192 x 192 x 256 -> 3), four calls: driver of the TCC request-counter pass
(scripts/gpu_job.sh tcc) that shows whether the halo rows neighbouring bands share are fetched from HBM once or twice (4.72 M lines of x;
5.45 M with every halo row twice)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pesr_amd import ops
torch.manual_seed(0)
N, H, W, C = 16, 192, 192, 256
x = torch.rand(N, H, W, C, device="cuda") - 0.5
w = (torch.rand(3, C, 3, 3, device="cuda") - 0.5) * 0.1
b = torch.rand(3, device="cuda")
for _ in range(4):
    ops.conv3x3_fwd(x, None, b, 3, w_oihw=w)
torch.cuda.synchronize()
