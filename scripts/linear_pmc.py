"""The three Linear kernels at the Discriminator's classifier.0 shape (16 x 73728 -> 1024), four calls each behind a 512 MB flush: the
driver of the TCC request-counter pass (scripts/gpu_job.sh tcc) that settles how much of the forward's counter-vs-algorithmic traffic
(464.9 MB against 307 MB) is 32-byte requests / L2 misses.  Algorithmic bytes: 302 MB of weights per call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pesr_amd import ops
torch.manual_seed(0)
M, NF, K = 16, 1024, 73728
xl = torch.rand(M, K, device="cuda") - 0.5
wl = (torch.rand(NF, K, device="cuda") - 0.5) * 0.01
bl = torch.rand(NF, device="cuda")
dyl = torch.rand(M, NF, device="cuda") - 0.5
dwl = torch.empty_like(wl)
flush = torch.empty(128 * 1024 * 1024, device="cuda")
for _ in range(4):
    for f in (lambda: ops.linear_fwd(xl, wl, bl, act=ops.ACT_LRELU, slope=0.2), lambda: ops.linear_dgrad(dyl, wl),
              lambda: ops.linear_wgrad(dyl, xl, dw_out=dwl)):
        flush.zero_()
        f()
torch.cuda.synchronize()
