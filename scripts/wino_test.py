"""Winograd F(2,3)-along-x conv vs torch conv2d (fp64 on CPU) and vs the direct MFMA kernel: accuracy and time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from pesr_amd import ops, _lib
L = _lib.lib()
S = lambda: torch.cuda.current_stream().cuda_stream
def wino(x, w, b=None, skip=None, mask=None, alpha=1.0, act=0, mode=0):
    N, H, W, Cin = x.shape
    O, I = w.shape[0], w.shape[1]
    Cout = O if mode == 0 else I
    wp = torch.empty(12 * O * I, device="cuda")
    assert L.pesr_pack_conv3x3_wino(w.data_ptr(), wp.data_ptr(), O, I, mode, 0, S()) == 0
    y = torch.empty(N, H, W, Cout, device="cuda")
    p = lambda t: None if t is None else t.data_ptr()
    rc = L.pesr_conv3x3_wino(x.data_ptr(), wp.data_ptr(), p(b), p(skip), p(mask), y.data_ptr(), N, H, W, Cin, Cout, alpha, act, 0.0, 0, 0, None, 0, S())
    assert rc == 0, rc
    return y
def nhwc(t): return t.permute(0, 2, 3, 1).contiguous().cuda()
def nchw(t): return t.permute(0, 3, 1, 2).cpu()
torch.manual_seed(0)
worst = 0.0
for (N, H, W, Cin, Cout) in [(1, 6, 48, 16, 128), (2, 7, 10, 32, 128), (1, 48, 48, 256, 256), (1, 13, 96, 64, 256), (2, 5, 2, 16, 128), (1, 9, 194, 16, 128)]:
    x = torch.rand(N, Cin, H, W) - 0.5; w = (torch.rand(Cout, Cin, 3, 3) - 0.5) * 0.1; b = torch.rand(Cout) - 0.5
    skip = torch.rand(N, Cout, H, W) - 0.5; mk = torch.rand(N, Cout, H, W) - 0.5
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    y = wino(nhwc(x), w.cuda(), b.cuda())
    e1 = (nchw(y).double() - ref).abs().max().item() / ref.abs().max().item()
    yd = ops.conv3x3_fwd(nhwc(x), ops.pack_conv3x3(w.cuda(), 0), b.cuda(), Cout)
    e0 = (nchw(yd).double() - ref).abs().max().item() / ref.abs().max().item()
    ref2 = torch.relu(torch.where(mk.double() > 0, ref * 0.1, torch.zeros_like(ref)) + skip.double())
    y2 = wino(nhwc(x), w.cuda(), b.cuda(), nhwc(skip), nhwc(mk), 0.1, 1)
    e2 = (nchw(y2).double() - ref2).abs().max().item() / ref2.abs().max().item()
    # input gradient: dx = conv_transpose(dy, w) for dy [N, Cout, H, W]
    dy = torch.rand(N, Cout, H, W) - 0.5
    dref = F.conv_transpose2d(dy.double(), w.double(), padding=1)
    dx = wino(nhwc(dy), w.cuda(), mode=1) if Cin % 128 == 0 else None
    e3 = (nchw(dx).double() - dref).abs().max().item() / dref.abs().max().item() if dx is not None else float("nan")
    print(f"{N}x{H}x{W} {Cin}->{Cout}: rel err wino {e1:.2e} (direct {e0:.2e}), fused epilogue {e2:.2e}, dgrad {e3:.2e}")
    worst = max(worst, e1, e2, 0.0 if dx is None else e3)
print("worst", worst)
# timing, K1 shape
N, H, W, C = 16, 48, 48, 256
x = torch.rand(N, H, W, C, device="cuda") - 0.5; w = (torch.rand(C, C, 3, 3, device="cuda") - 0.5) * 0.1; b = torch.rand(C, device="cuda")
wpw = torch.empty(12 * C * C, device="cuda"); L.pesr_pack_conv3x3_wino(w.data_ptr(), wpw.data_ptr(), C, C, 0, 0, S())
wpd = ops.pack_conv3x3(w, 0); y = torch.empty(N, H, W, C, device="cuda")
def t(fn, it=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record()
        for _ in range(it): fn()
        e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1) / it * 1e3)
    return best
td = t(lambda: L.pesr_conv3x3_fwd(x.data_ptr(), wpd.data_ptr(), b.data_ptr(), None, None, y.data_ptr(), N, H, W, C, C, 1, 1.0, 1, 0.0, 0, None, 0, S()))
tw = t(lambda: L.pesr_conv3x3_wino(x.data_ptr(), wpw.data_ptr(), b.data_ptr(), None, None, y.data_ptr(), N, H, W, C, C, 1.0, 1, 0.0, 0, 0, None, 0, S()))
print(f"K1 shape: direct {td:.1f} us, winograd {tw:.1f} us  ({43.487e9 / tw / 1e6:.1f} algorithmic TFLOP/s)")
