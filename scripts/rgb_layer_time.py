"""Time the C <-> 3 layer kernels (forward, input gradient, weight gradient) at the Generator's last-conv shape and report
their HBM rate (the 604 MB activation tensor is the traffic; 6.29 TB/s is the measured-achievable HBM rate SURVEY.md quotes)."""
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

def main():
    torch.manual_seed(0)
    N, H, W, C = 16, 192, 192, 256
    x = torch.rand(N, H, W, C, device="cuda") - 0.5
    w = (torch.rand(3, C, 3, 3, device="cuda") - 0.5) * 0.1
    b = torch.rand(3, device="cuda")
    dy = torch.rand(N, H, W, 3, device="cuda") - 0.5
    mb = x.numel() * 4 / 1e6
    wp = ops.pack_conv3x3(w, 0)
    rows = []
    rows.append(("forward, dedicated kernel", timeit(lambda: ops.conv3x3_fwd(x, wp, b, 3, w_oihw=w))))
    ops.USE_RGB_OUT = False
    rows.append(("forward, implicit GEMM (Cout padded to 16)", timeit(lambda: ops.conv3x3_fwd(x, wp, b, 3, w_oihw=w))))
    ops.USE_RGB_OUT = True
    rows.append(("input gradient (conv3x3_rgb_dgrad)", timeit(lambda: ops.conv3x3_rgb_dgrad(dy, w, (N, H, W, C)))))
    rows.append(("weight gradient (conv3x3_wgrad_rgb)", timeit(lambda: ops.conv3x3_wgrad_rgb(x, dy, 1))))
    for name, us in rows:
        print(f"{name:<46} {us:8.1f} us   {mb / us:6.2f} TB/s  ({100 * mb / us / 6.29:5.1f} % of 6.29 TB/s)")
    # the 3 -> 64 layers (Discriminator features.0, vgg19 features.0) at 16 x 192 x 192: the 151 MB 64-channel tensor is the traffic
    C = 64
    x3 = torch.rand(N, H, W, 3, device="cuda") * 255
    w3 = (torch.rand(C, 3, 3, 3, device="cuda") - 0.5) * 0.2
    b3 = torch.rand(C, device="cuda")
    dy64 = torch.rand(N, H, W, C, device="cuda") - 0.5
    mb = dy64.numel() * 4 / 1e6
    wpd = ops.pack_conv3x3(w3, 1)
    rows = [("3->64 forward (conv_rgb_in)", timeit(lambda: ops.conv3x3_fwd(x3, None, b3, C, w_oihw=w3))),
            ("3->64 input gradient, streaming (rgb_in_dgrad)", timeit(lambda: ops.conv3x3_rgb_in_dgrad(dy64, w3, (N, H, W, 3)))),
            ("3->64 input gradient, implicit GEMM (round 3)", timeit(lambda: ops.conv3x3_dgrad(dy64, wpd, (N, H, W, 3)))),
            ("3->64 weight gradient (conv3x3_wgrad_rgb)", timeit(lambda: ops.conv3x3_wgrad_rgb(dy64, x3, 0)))]
    for name, us in rows:
        print(f"{name:<46} {us:8.1f} us   {mb / us:6.2f} TB/s  ({100 * mb / us / 6.29:5.1f} % of 6.29 TB/s)")

if __name__ == "__main__":
    main()
