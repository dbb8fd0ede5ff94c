"""Phase stamps of the y-nested weight-gradient kernel at the G-body shape (a build with -DX4_STAMPS: bash scripts/build_variant.sh x4st
conv3x3_wgrad_wino4.hip -DX4_STAMPS): python scripts/x4_phases.py exp/libx4st.so"""
import ctypes, os, sys
import numpy as np
import torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from pesr_amd import _lib
l = ctypes.CDLL(os.path.join(R, sys.argv[1]))
for name, (res, a) in _lib.SIGNATURES.items():
    if hasattr(l, name):
        f = getattr(l, name); f.restype = res; f.argtypes = a
N, H, W, C = 16, 48, 48, 256
x = torch.rand(N, H, W, C, device="cuda") - 0.5
dy = torch.rand(N, H, W, C, device="cuda") - 0.5
dw = torch.empty(C, C, 3, 3, device="cuda"); db = torch.empty(C, device="cuda")
ws = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for _ in range(int(os.environ.get('X4_LAUNCHES', '3000'))):
    rc = l.pesr_conv3x3_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), db.data_ptr(), N, H, W, C, C, 1, 1.0, 0, 0, 0, ws.data_ptr(), ws.numel(), s)
    assert rc == 0
torch.cuda.synchronize()
st = (ctypes.c_ulonglong * (1024 * 8))(); sg = (ctypes.c_ulonglong * (8 * 64))()
l.pesr_debug_x4_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
assert l.pesr_debug_x4_stamps(st, sg) == 0
t = np.frombuffer(st, dtype=np.uint64).reshape(1024, 8)[:256].astype(np.int64)
t0 = t[:, 0].min()
us = (t - t0) / 100.0
labels = ["entry", "prologue (first segment staged)", "main loop", "final barrier", "bias + G2^T + park", "G4^T + slab stores issued"]
print(f"kernel span (first entry -> last exit of wave 0): {us[:, 5].max():.2f} us; entry skew max {us[:, 0].max():.2f}")
for i in range(1, 6):
    d = us[:, i] - us[:, i - 1]
    print(f"  {labels[i]:<36} median {np.median(d):7.2f}  min {d.min():7.2f}  max {d.max():7.2f}")
ghz = (t[:, 7] - t[:, 6]) / np.maximum(t[:, 2] - t[:, 1], 1) * 0.1
print(f"  in-kernel shader clock over the main loop: median {np.median(ghz):.3f} GHz (min {ghz.min():.3f}, max {ghz.max():.3f})")
g = np.frombuffer(sg, dtype=np.uint64).reshape(8, 64).astype(np.int64)
for w in (0, 3, 7):
    d = np.diff(g[w, :48]) / 100.0
    print(f"  workgroup 0, wave {w}: per-segment us:", " ".join(f"{v:.2f}" for v in d))

if hasattr(l, "pesr_debug_x4_bar"):
    bb = (ctypes.c_ulonglong * (12 * 8 * 2))()
    l.pesr_debug_x4_bar.argtypes = [ctypes.c_void_p]
    assert l.pesr_debug_x4_bar(bb) == 0
    b = np.frombuffer(bb, dtype=np.uint64).reshape(12, 8, 2).astype(np.int64)
    print("  barrier of workgroup 0, segments 8..15 (shader cycles): arrival relative to the earliest wave / wait until release")
    for si in range(8):
        a0 = b[:, si, 0].min()
        print(f"   seg {8 + si}: segment length {int(b[:, si, 1].max() - (b[:, si - 1, 1].max() if si else a0)):6d}  "
              + " ".join(f"w{w}:{int(b[w, si, 0] - a0)}/{int(b[w, si, 1] - b[w, si, 0])}" for w in range(12)))

if hasattr(l, "pesr_debug_x4_stg"):
    gg = (ctypes.c_ulonglong * (12 * 8 * 4))()
    l.pesr_debug_x4_stg.argtypes = [ctypes.c_void_p]
    assert l.pesr_debug_x4_stg(gg) == 0
    g4 = np.frombuffer(gg, dtype=np.uint64).reshape(12, 8, 4).astype(np.int64)
    print("  staging block of workgroup 0 (shader cycles): wait for the loads / transform + LDS stores / issue of the next loads; start relative to the barrier release of the previous segment")
    for si in (2, 3):
        rel = b[:, si - 1, 1].max()
        print(f"   seg {8 + si}: " + " ".join(f"w{w}:@{int(g4[w, si, 0] - rel)} {int(g4[w, si, 1] - g4[w, si, 0])}/{int(g4[w, si, 2] - g4[w, si, 1])}/{int(g4[w, si, 3] - g4[w, si, 2])}" for w in range(12)))
