"""Launch the HBM-bound kernels of the train step a few times each (for the separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE /
GRBM_GUI_ACTIVE passes of scripts/profile_round3.sh): the C <-> 3 layer at 16 x 192^2 x 256 (forward, input gradient, weight
gradient), Linear(73728, 1024) at batch 16 (forward, input gradient, weight gradient), BatchNorm + LeakyReLU at 16 x 192^2 x 64,
the 3 -> 64 input conv.  Algorithmic bytes per call are printed for the summary table."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pesr_amd import ops
torch.manual_seed(0)
N, H, W, C = 16, 192, 192, 256
x = torch.rand(N, H, W, C, device="cuda") - 0.5
w3 = (torch.rand(3, C, 3, 3, device="cuda") - 0.5) * 0.1
b3 = torch.rand(3, device="cuda")
dy3 = torch.rand(N, H, W, 3, device="cuda") - 0.5
M, NF, K = 16, 1024, 73728
xl = torch.rand(M, K, device="cuda") - 0.5
wl = (torch.rand(NF, K, device="cuda") - 0.5) * 0.01
bl = torch.rand(NF, device="cuda")
dyl = torch.rand(M, NF, device="cuda") - 0.5
dwl = torch.empty_like(wl)
z = torch.rand(N, H, W, 64, device="cuda") - 0.5
ga, be = torch.rand(64, device="cuda") + 0.5, torch.rand(64, device="cuda") - 0.5
rm, rv, nb = torch.zeros(64, device="cuda"), torch.ones(64, device="cuda"), torch.zeros((), dtype=torch.long, device="cuda")
x3 = torch.rand(N, H, W, 3, device="cuda") * 255
w64 = (torch.rand(64, 3, 3, 3, device="cuda") - 0.5) * 0.1
flush = torch.empty(128 * 1024 * 1024, device="cuda")     # 512 MB: evicts the 256 MB Infinity Cache between calls
for _ in range(4):
    for f in (lambda: ops.conv3x3_fwd(x, None, b3, 3, w_oihw=w3),
              lambda: ops.conv3x3_rgb_dgrad(dy3, w3, (N, H, W, C)),
              lambda: ops.conv3x3_wgrad_rgb(x, dy3, 1),
              lambda: ops.linear_fwd(xl, wl, bl, act=ops.ACT_LRELU, slope=0.2),
              lambda: ops.linear_dgrad(dyl, wl),
              lambda: ops.linear_wgrad(dyl, xl, dw_out=dwl),
              lambda: ops.bn_lrelu_fwd(z, ga, be, rm, rv, nb),
              lambda: ops.conv3x3_fwd(x3, None, None, 64, w_oihw=w64),
              lambda: ops.conv3x3_rgb_in_dgrad(z, w64, (N, H, W, 3)),             # round 4: dx of the 3 -> 64 convs, streaming kernel
              lambda: ops.conv3x3_wgrad_rgb(z, x3, 0, want_bias=False)):          # round 4: their weight gradient on the MFMA form
        flush.zero_()
        f()
torch.cuda.synchronize()
print("algorithmic MB per call: C->3 forward / C<-3 input gradient / its weight gradient 604; Linear fwd / dgrad / wgrad 302; "
      "BN+LReLU forward 151 read twice + 151 written; 3->64 forward 151 written; 3->64 input gradient and weight gradient 151 read")
