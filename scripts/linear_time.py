"""Time the three Linear kernels at the Discriminator's classifier.0 shape (16 x 73728 -> 1024: the 302 MB weight matrix is the
traffic) and report their HBM rate against the 6.29 TB/s SURVEY.md quotes as achievable."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pesr_amd import ops

def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]

def main(M=16):
    torch.manual_seed(0)
    N, K = 1024, 73728
    print(f"M = {M} rows")
    x = torch.rand(M, K, device="cuda") - 0.5
    w = (torch.rand(N, K, device="cuda") - 0.5) * 0.01
    b = torch.rand(N, device="cuda")
    dy = torch.rand(M, N, device="cuda") - 0.5
    dw = torch.empty_like(w)
    mb = w.numel() * 4 / 1e6
    # a 512 MB write between calls would evict the matrix from the 256 MB MALL; back-to-back calls of one kernel re-read what
    # the previous call left there, so the step's rate (other tensors in between) lies between the two columns
    flush = torch.empty(128 * 1024 * 1024, device="cuda")
    for name, f in (("forward", lambda: ops.linear_fwd(x, w, b, act=ops.ACT_LRELU, slope=0.2)),
                    ("input gradient", lambda: ops.linear_dgrad(dy, w)),
                    ("weight gradient", lambda: ops.linear_wgrad(dy, x, dw_out=dw))):
        warm = timeit(f)
        cold = timeit(lambda: (flush.zero_(), f())[1]) - timeit(lambda: flush.zero_())
        print(f"{name:<16} back-to-back {warm:7.1f} us = {mb / warm:5.2f} TB/s ({100 * mb / warm / 6.29:5.1f} %)   after a 512 MB flush {cold:7.1f} us = {mb / cold:5.2f} TB/s ({100 * mb / cold / 6.29:5.1f} %)")

if __name__ == "__main__":
    main(16)
    main(32)      # the classifier on [hr; sr]: one pass over W for two of the Discriminator's calls (two 16-row MFMA tiles)
