"""Experiment: a chain of K1 convs on the full batch (one stream) vs. two half-batch chains on two streams."""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from pesr_amd import _lib
def load(path):
    l = ctypes.CDLL(os.path.join(R, path))
    for name, (res, args) in _lib.SIGNATURES.items():
        f = getattr(l, name); f.restype = res; f.argtypes = args
    return l
head, split = load(sys.argv[1]), load(sys.argv[2])
N, H, W, C = 16, 48, 48, 256
a = torch.rand(N, H, W, C, device="cuda") - 0.5
b = torch.empty_like(a)
w = (torch.rand(C, C, 3, 3, device="cuda") - 0.5) * 0.02
bias = torch.zeros(C, device="cuda")
wp = torch.empty(9 * C * C, device="cuda")
s0 = torch.cuda.current_stream()
head.pesr_pack_conv3x3(w.data_ptr(), wp.data_ptr(), C, C, 0, 0, s0.cuda_stream)
s1 = torch.cuda.Stream()
LAYERS = 32
def chain(lib, x, y, n, stream):
    for i in range(LAYERS):
        src, dst = (x, y) if i % 2 == 0 else (y, x)
        lib.pesr_conv3x3_fwd(src.data_ptr(), wp.data_ptr(), bias.data_ptr(), None, None, dst.data_ptr(), n, H, W, C, C, 1, 1.0, 1, 0.0, 0, None, 0, stream.cuda_stream)
def full(lib):
    chain(lib, a, b, N, s0)
def halves(lib):
    s1.wait_stream(s0)
    chain(lib, a[:8], b[:8], 8, s0)
    chain(lib, a[8:], b[8:], 8, s1)
    s0.wait_stream(s1)
for name, fn, lib in (("full batch, 8-wave tiles, 1 stream", full, head), ("full batch, 4-wave split-N tiles, 1 stream", full, split),
                      ("2 half-batch chains, 8-wave tiles, 2 streams", halves, head), ("2 half-batch chains, 4-wave split-N tiles, 2 streams", halves, split)):
    for _ in range(2): fn(lib)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record(); fn(lib); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3 / LAYERS)
    print(f"{name:55s} {min(ts):7.1f} us per layer (full-batch equivalent)")
