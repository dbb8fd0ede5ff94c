"""Time the Winograd kernel on the K1 shape for several library builds (same-session, interleaved)."""
import ctypes, os, sys, statistics
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from pesr_amd import _lib
libs = sys.argv[1:]
hs = []
for path in libs:
    l = ctypes.CDLL(os.path.join(R, path))
    for name, (res, args) in _lib.SIGNATURES.items():
        f = getattr(l, name); f.restype = res; f.argtypes = args
    hs.append(l)
N, H, W, C = 16, 48, 48, 256
x = torch.rand(N, H, W, C, device="cuda") - 0.5; w = (torch.rand(C, C, 3, 3, device="cuda") - 0.5) * 0.1; b = torch.rand(C, device="cuda")
wp = torch.empty(12 * C * C, device="cuda"); y = torch.empty(N, H, W, C, device="cuda")
s = torch.cuda.current_stream().cuda_stream
hs[0].pesr_pack_conv3x3_wino(w.data_ptr(), wp.data_ptr(), C, C, 0, 0, s)
def run(l, it=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        l.pesr_conv3x3_wino(x.data_ptr(), wp.data_ptr(), b.data_ptr(), None, None, y.data_ptr(), N, H, W, C, C, 1.0, 1, 0.0, 0, 0, None, 0, s)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
for l in hs: run(l, 5)
res = {p: [] for p in libs}
for _ in range(6):
    for p, l in zip(libs, hs): res[p].append(run(l))
for p in libs:
    print(f"{p:24s} median {statistics.median(res[p]):7.1f} us  min {min(res[p]):7.1f}")

if os.environ.get("WN_CLOCK"):
    l = hs[-1]; run(l, 2); torch.cuda.synchronize()
    t = y.view(-1)[:8].view(torch.int64).tolist()
    print(f"block 7 of {libs[-1]}: prologue {t[0] / 100:.1f} us, main loop {t[1] / 100:.1f} us, epilogue {t[2] / 100:.1f} us")
