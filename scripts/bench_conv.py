"""Scratch micro-benchmark of the conv kernels on the shapes of SURVEY.md Appendix A (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pesr_amd import ops

def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

shapes = [  # name, N, H, W, Cin, Cout, stride, ps
    ("G body 256->256 @48", 16, 48, 48, 256, 256, 1, False),
    ("G up0 256->1024 @48 (ps)", 16, 48, 48, 256, 1024, 1, True),
    ("G up2 256->1024 @96 (ps)", 16, 96, 96, 256, 1024, 1, True),
    ("VGG 64->64 @192", 16, 192, 192, 64, 64, 1, False),
    ("VGG 128->128 @96", 16, 96, 96, 128, 128, 1, False),
    ("VGG 512->512 @24", 16, 24, 24, 512, 512, 1, False),
    ("VGG 512->512 @12", 16, 12, 12, 512, 512, 1, False),
    ("RGB-out 256->3 @192", 16, 192, 192, 256, 3, 1, False),
    ("RGB-out 64->3 @192", 16, 192, 192, 64, 3, 1, False),
    ("D 64->64 s2 @192", 16, 192, 192, 64, 64, 2, False),
    ("D 128->128 s2 @96", 16, 96, 96, 128, 128, 2, False),
    ("D 256->256 s2 @48", 16, 48, 48, 256, 256, 2, False),
    ("D 512->512 s2 @24", 16, 24, 24, 512, 512, 2, False),
]
only = sys.argv[1] if len(sys.argv) > 1 else None
for name, N, H, W, Cin, Cout, s, ps in shapes:
    if only and only not in name: continue
    x = torch.rand(N, H, W, Cin, device="cuda") - 0.5
    w = (torch.rand(Cout, Cin, 3, 3, device="cuda") - 0.5) * 0.1
    b = torch.rand(Cout, device="cuda")
    OH, OW = (H - 1) // s + 1, (W - 1) // s + 1
    wp = ops.pack_conv3x3(w, 0, ps); wpd = ops.pack_conv3x3(w, 1, ps)
    bp = ops.pack_bias_ps(b) if ps else b
    dy = torch.rand((N, 2 * OH, 2 * OW, Cout // 4) if ps else (N, OH, OW, Cout), device="cuda") - 0.5
    flop = 2.0 * N * OH * OW * Cout * Cin * 9
    t = timeit(lambda: ops.conv3x3_fwd(x, wp, bp, Cout, s, ps_out=ps))
    line = f"{name:28s} fwd {t*1e3:8.1f} us {flop/t/1e9:7.1f} TF/s"
    if Cout == 3:
        print(line, flush=True); continue
    t = timeit(lambda: ops.conv3x3_dgrad(dy, wpd, (N, H, W, Cin), s, ps_in=ps))
    line += f" | dgrad {t*1e3:8.1f} us {flop/t/1e9:7.1f} TF/s"
    t = timeit(lambda: ops.conv3x3_wgrad(x, dy, s, ps_in=ps))
    line += f" | wgrad {t*1e3:8.1f} us {flop/t/1e9:7.1f} TF/s"
    t = timeit(lambda: ops.pack_conv3x3(w, 0, ps))
    line += f" | pack {t*1e3:7.1f} us"
    print(line, flush=True)
