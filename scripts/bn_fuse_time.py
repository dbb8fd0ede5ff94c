"""Time the Discriminator's conv kernels with and without the BatchNorm-sums epilogue (round 6), interleaved, for library A/B runs:
    PESR_HIP_LIB=exp/libX.so python scripts/bn_fuse_time.py
forward: plain conv vs conv + (sum z, sum z^2); input gradient: plain vs masked gradient + (sum g', sum g' xhat)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pesr_amd import ops
torch.manual_seed(0)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


# (name, N, H, W, Cin, Cout, stride, wino4)
LAYERS = [("f1", 16, 192, 192, 64, 64, 2, False), ("f2", 16, 96, 96, 64, 128, 1, True), ("f3", 16, 96, 96, 128, 128, 2, False),
          ("f4", 16, 48, 48, 128, 256, 1, True), ("f5", 16, 48, 48, 256, 256, 2, False), ("f7", 16, 24, 24, 512, 512, 2, False)]
flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
for name, N, H, W, Cin, Cout, s, w4 in LAYERS:
    x = torch.rand(N, H, W, Cin, device="cuda") - 0.5
    w = (torch.rand(Cout, Cin, 3, 3, device="cuda") - 0.5) * 0.1
    pack = ops.pack_conv3x3_wino4 if w4 else ops.pack_conv3x3
    wp, wpd = pack(w, 0), pack(w, 1)
    OH, OW = (H - 1) // s + 1, (W - 1) // s + 1
    dy = torch.rand(N, OH, OW, Cout, device="cuda") - 0.5
    gamma, beta = torch.rand(Cin, device="cuda") + 0.5, torch.rand(Cin, device="cuda") - 0.5
    stats = torch.stack([x.mean((0, 1, 2)), 1.0 / x.var((0, 1, 2)).sqrt()]).contiguous()
    t_f = timed(lambda: ops.conv3x3_fwd(x, wp, None, Cout, s))
    r = ops.conv3x3_fwd_bn_stats(x, wp, None, Cout, s)
    t_fs = timed(lambda: ops.conv3x3_fwd_bn_stats(x, wp, None, Cout, s)) if r is not None else float("nan")
    t_d = timed(lambda: ops.conv3x3_dgrad(dy, wpd, (N, H, W, Cin), s))
    r = ops.conv3x3_dgrad_bn_sums(dy, wpd, (N, H, W, Cin), s, x, stats, gamma, beta, 0.2)
    t_ds = timed(lambda: ops.conv3x3_dgrad_bn_sums(dy, wpd, (N, H, W, Cin), s, x, stats, gamma, beta, 0.2)) if r is not None else float("nan")
    # what the fused forms replace: the statistics pass over z (forward), the reduction pass over (z, dy) (backward)
    z = torch.rand(N, OH, OW, Cout, device="cuda")
    rm, rv, nb = torch.zeros(Cout, device="cuda"), torch.ones(Cout, device="cuda"), torch.zeros((), dtype=torch.long, device="cuda")
    g2, b2 = torch.ones(Cout, device="cuda"), torch.zeros(Cout, device="cuda")
    t_bn = timed(lambda: ops.bn_lrelu_fwd(z, g2, b2, rm, rv, nb))
    gx = torch.rand(N, H, W, Cin, device="cuda")
    t_bb = timed(lambda: ops.bn_lrelu_bwd(x, gx, gamma, beta, stats))
    print(f"{name}: fwd {t_f:6.1f} us, + sums {t_fs:6.1f} | dgrad {t_d:6.1f} us, + masked sums {t_ds:6.1f} | un-fused BN fwd (3 launches) {t_bn:6.1f}, BN bwd of the input tensor (3 launches) {t_bb:6.1f}")
