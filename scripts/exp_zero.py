import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pesr_amd import ops
def timeit(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
N,H,W,C = 16,48,48,256
w = (torch.rand(C, C, 3, 3, device="cuda") - 0.5) * 0.1
b = torch.rand(C, device="cuda")
wp = ops.pack_conv3x3(w, 0)
flop = 2.0*N*H*W*C*C*9
for name, x, wpk in [("random x, random w", torch.rand(N,H,W,C,device="cuda")-0.5, wp),
                     ("zero x, random w", torch.zeros(N,H,W,C,device="cuda"), wp),
                     ("zero x, zero w", torch.zeros(N,H,W,C,device="cuda"), torch.zeros_like(wp)),
                     ("random x, random w (again)", torch.rand(N,H,W,C,device="cuda")-0.5, wp),
                     ("small ints", torch.randint(0,4,(N,H,W,C),device="cuda").float(), wp)]:
    for rep in range(2):
        t = timeit(lambda: ops.conv3x3_fwd(x, wpk, b, C, act=ops.ACT_RELU))
        print(f"{name:30s} {t*1e3:8.1f} us {flop/t/1e9:7.1f} TF/s", flush=True)
