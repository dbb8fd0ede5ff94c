"""Stride-2 conv input gradient (four parity-class launches) and forward at the Discriminator's layer shapes, interleaved over
the libraries given:  python scripts/s2_dgrad_time.py pesr_amd/libpesr_hip.so exp/libX.so"""
import ctypes, os, statistics, sys
import torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from pesr_amd import _lib
libs = sys.argv[1:] or ["pesr_amd/libpesr_hip.so"]
handles = []
for path in libs:
    l = ctypes.CDLL(os.path.join(R, path))
    for name, (res, a) in _lib.SIGNATURES.items():
        f = getattr(l, name, None)          # (an older variant library lacks the newest entry points)
        if f is not None:
            f.restype = res; f.argtypes = a
    handles.append(l)
s = torch.cuda.current_stream().cuda_stream
ws = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
for (N, H, W, Ci, Co) in [(16, 192, 192, 64, 64), (16, 96, 96, 128, 128), (16, 48, 48, 256, 256), (16, 24, 24, 512, 512)]:
    dy = torch.rand(N, H // 2, W // 2, Co, device="cuda") - 0.5
    x = torch.rand(N, H, W, Ci, device="cuda") - 0.5
    w = (torch.rand(Co, Ci, 3, 3, device="cuda") - 0.5) * 0.1
    wpd = torch.empty(9 * Ci * Co, device="cuda"); wpf = torch.empty(9 * Ci * Co, device="cuda")
    handles[0].pesr_pack_conv3x3(w.data_ptr(), wpd.data_ptr(), Co, Ci, 1, 0, s)
    handles[0].pesr_pack_conv3x3(w.data_ptr(), wpf.data_ptr(), Co, Ci, 0, 0, s)
    dx = [torch.empty(N, H, W, Ci, device="cuda") for _ in handles]
    y = [torch.empty(N, H // 2, W // 2, Co, device="cuda") for _ in handles]
    def run(i, what, iters=20):
        l = handles[i]
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            if what == "dgrad":
                rc = l.pesr_conv3x3_dgrad(dy.data_ptr(), wpd.data_ptr(), None, None, dx[i].data_ptr(), N, H, W, Ci, Co, 2, 1.0, 0, None, 0, s)
            else:
                rc = l.pesr_conv3x3_fwd(x.data_ptr(), wpf.data_ptr(), None, None, None, y[i].data_ptr(), N, H, W, Ci, Co, 2, 1.0, 0, 0.0, 0, ws.data_ptr(), ws.numel(), s)
            assert rc == 0, rc
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3
    gf = 2.0 * N * (H // 2) * (W // 2) * Ci * Co * 9 / 1e9
    for what in ("fwd", "dgrad"):
        for i in range(len(handles)): run(i, what, 3)
        res = [[] for _ in handles]
        for _ in range(5):
            for i in range(len(handles)): res[i].append(run(i, what))
        out = y if what == "fwd" else dx
        for i, p in enumerate(libs):
            m = statistics.median(res[i])
            same = "" if i == 0 else f"  max|diff vs first| {float((out[i] - out[0]).abs().max()):.2e}"
            print(f"s2 {what:5s} {N}x{H}x{W}x{Ci}->{Co}  {p:28s} {m:7.1f} us  {gf / m:6.1f} TF/s ({100 * gf / m / 157.3:4.1f} %){same}")
