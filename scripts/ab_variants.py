"""Same-session interleaved A/B of conv-kernel builds (each a separate libpesr_hip*.so loaded through ctypes)."""
import ctypes, sys, os, statistics
import torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from pesr_amd import _lib
what = "fwd"
args = sys.argv[1:]
if args and args[0] in ("fwd", "fwdskip", "wgrad"):
    what = args.pop(0)
libs = args
N, H, W, C = 16, 48, 48, 256
x = torch.rand(N, H, W, C, device="cuda") - 0.5
w = (torch.rand(C, C, 3, 3, device="cuda") - 0.5) * 0.1
b = torch.rand(C, device="cuda"); y = torch.empty(N, H, W, C, device="cuda")
wp = torch.empty(9 * C * C, device="cuda")
handles = []
for path in libs:
    l = ctypes.CDLL(os.path.join(R, path))
    for name, (res, args) in _lib.SIGNATURES.items():
        f = getattr(l, name); f.restype = res; f.argtypes = args
    handles.append(l)
s = torch.cuda.current_stream().cuda_stream
handles[0].pesr_pack_conv3x3(w.data_ptr(), wp.data_ptr(), C, C, 0, 0, s)
dy = torch.rand(N, H, W, C, device="cuda") - 0.5
dw = torch.empty(C, C, 3, 3, device="cuda"); dbias = torch.empty(C, device="cuda")
ws = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
COLD = int(os.environ.get("AB_COLD", "0"))      # rotate over this many x/dy sets (> 256 MB MALL in total => HBM-cold inputs)
xs = [x] + [torch.rand_like(x) - 0.5 for _ in range(max(COLD - 1, 0))]
dys = [dy] + [torch.rand_like(dy) - 0.5 for _ in range(max(COLD - 1, 0))]
def run(l, iters=20):
    if what == "wgrad":
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters):
            xi, di = xs[i % len(xs)], dys[i % len(dys)]
            l.pesr_conv3x3_wgrad(xi.data_ptr(), di.data_ptr(), dw.data_ptr(), dbias.data_ptr(), N, H, W, C, C, 1, 1.0, 0, ws.data_ptr(), ws.numel(), s)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3
    return run_fwd(l, iters)
def run_fwd(l, iters=20):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        if what == "fwdskip":   # the dgrad-of-conv1 epilogue: ReLU mask + residual add
            l.pesr_conv3x3_fwd(x.data_ptr(), wp.data_ptr(), None, dy.data_ptr(), x.data_ptr(), y.data_ptr(), N, H, W, C, C, 1, 0.1, 0, 0.0, 0, None, 0, s)
        else:
            l.pesr_conv3x3_fwd(x.data_ptr(), wp.data_ptr(), b.data_ptr(), None, None, y.data_ptr(), N, H, W, C, C, 1, 1.0, 1, 0.0, 0, None, 0, s)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for l in handles: run(l, 5)
res = {p: [] for p in libs}
for rnd in range(6):
    for p, l in zip(libs, handles):
        res[p].append(run(l))
if os.environ.get("AB_CLOCK"):
    for p, l in zip(libs, handles):
        run(l, 3); torch.cuda.synchronize()
        t = ws[:16].view(torch.int64).tolist()
        print(f"{p:28s} block 5: {t[0] / 100.0:.1f} us inside the kernel, shader clock {t[1] / max(t[0], 1) / 10.0:.3f} GHz")
for p in libs:
    print(f"{p:28s} median {statistics.median(res[p]):7.1f} us  min {min(res[p]):7.1f}  all {[round(v) for v in res[p]]}")
