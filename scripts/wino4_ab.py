"""Same-session interleaved A/B of builds of the F(4,3) conv kernel (each a separate libpesr_hip*.so, scripts/build_variant.sh):
    python scripts/wino4_ab.py [fwd|skip|mask|both|wgrad] pesr_amd/libpesr_hip.so exp/libX.so ...     (G-body shape, batch 16)
wgrad / wgrad1d / wgrad16 = the weight gradient (auto / 1-D F(4,3) on 32x32x2 / on 16x16x4); fwd = bias + ReLU (ResBlock conv1); skip = bias, x 0.1, + skip (conv2, and the input gradient of conv1 without bias); mask = x 0.1 and
ReLU mask (input gradient of conv2); both = mask + skip (tests only)."""
import ctypes, os, statistics, sys
import torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from pesr_amd import _lib
what = "fwd"
args = sys.argv[1:]
if args and args[0] in ("fwd", "skip", "mask", "both", "wgrad", "wgrad1d", "wgrad16", "clock"):
    what = args.pop(0)
libs = args
N, H, W, C = 16, 48, 48, 256
x = torch.rand(N, H, W, C, device="cuda") - 0.5
w = (torch.rand(C, C, 3, 3, device="cuda") - 0.5) * 0.1
b = torch.rand(C, device="cuda"); y = torch.empty(N, H, W, C, device="cuda"); sk = torch.rand(N, H, W, C, device="cuda") - 0.5
wp = torch.empty(18 * C * C, device="cuda")
handles = []
for path in libs:
    l = ctypes.CDLL(os.path.join(R, path))
    for name, (res, a) in _lib.SIGNATURES.items():
        if not hasattr(l, name): continue      # (an older build)
        f = getattr(l, name); f.restype = res; f.argtypes = a
    handles.append(l)
s = torch.cuda.current_stream().cuda_stream
handles[0].pesr_pack_conv3x3_wino4(w.data_ptr(), wp.data_ptr(), C, C, 0, 0, s)
dw = torch.empty(C, C, 3, 3, device="cuda"); dbias = torch.empty(C, device="cuda")
ws = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
def run(l, iters=20):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        if what.startswith("wgrad"):   # transposed Winograd weight gradient + its reduce kernel (algo 0 = auto; wgrad1d: 4 = the 1-D F(4,3) form; wgrad16: 3)
            rc = l.pesr_conv3x3_wgrad(x.data_ptr(), sk.data_ptr(), dw.data_ptr(), dbias.data_ptr(), N, H, W, C, C, 1, 1.0, 0,
                                      {"wgrad": 0, "wgrad1d": 4, "wgrad16": 3}[what], 0, ws.data_ptr(), ws.numel(), s)
        elif what == "both":   # ReLU mask + residual add
            rc = l.pesr_conv3x3_wino4(x.data_ptr(), wp.data_ptr(), None, sk.data_ptr(), x.data_ptr(), y.data_ptr(), N, H, W, C, C, 0.1, 0, 0.0, 0, 0, None, 0, s)
        elif what == "skip":
            rc = l.pesr_conv3x3_wino4(x.data_ptr(), wp.data_ptr(), b.data_ptr(), sk.data_ptr(), None, y.data_ptr(), N, H, W, C, C, 0.1, 0, 0.0, 0, 0, None, 0, s)
        elif what == "mask":
            rc = l.pesr_conv3x3_wino4(x.data_ptr(), wp.data_ptr(), None, None, sk.data_ptr(), y.data_ptr(), N, H, W, C, C, 0.1, 0, 0.0, 0, 0, None, 0, s)
        else:
            rc = l.pesr_conv3x3_wino4(x.data_ptr(), wp.data_ptr(), b.data_ptr(), None, None, y.data_ptr(), N, H, W, C, C, 1.0, 1, 0.0, 0, 0, None, 0, s)
        assert rc == 0, rc
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
if what == "clock":   # diagnostic builds with -DW4D_CLOCK: in-kernel shader clock of the forward kernel's main loop (bias + ReLU)
    import numpy as np
    clk = torch.zeros(1 << 16, dtype=torch.int64, device="cuda")
    n_launch = int(os.environ.get("W4_CLOCK_LAUNCHES", "60"))     # 60: a 10 ms burst; 3000: half a second of sustained load (the stamps of the LAST launch are read)
    for p, l in zip(libs, handles):
        for it in range(n_launch):
            rc = l.pesr_conv3x3_wino4(x.data_ptr(), wp.data_ptr(), b.data_ptr(), None, None, y.data_ptr(), N, H, W, C, C, 1.0, 1, 0.0, 0, 0,
                                      clk.data_ptr(), clk.numel() * 8, s)
            assert rc == 0
        torch.cuda.synchronize()
        c = clk[:512].cpu().numpy().reshape(256, 2).astype(np.float64)
        ok = c[:, 1] > 0
        ghz = c[ok, 0] / c[ok, 1] * 0.1
        us = c[ok, 1] / 100.0
        print(f"{p:34s} main loop: {np.median(us):7.1f} us (real-time counter), shader clock median {np.median(ghz):.3f} GHz (min {ghz.min():.3f} max {ghz.max():.3f}) over {int(ok.sum())} workgroups")
        clk.zero_()
    sys.exit(0)
for l in handles: run(l, 5)
res = {p: [] for p in libs}
for rnd in range(6):
    for p, l in zip(libs, handles):
        res[p].append(run(l))
for p in libs:
    print(f"{p:34s} median {statistics.median(res[p]):7.1f} us  min {min(res[p]):7.1f}  all {[round(v) for v in res[p]]}")
