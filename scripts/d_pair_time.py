"""What merging the Discriminator's two forwards of a phase into ONE batch-32 pass can buy (round 4, before building it): the
time of D at batch 32 against two calls at batch 16, forward only (no_grad), forward + backward with weight gradients (the D
phase) and forward + input-gradient-only backward (the G phase; at batch 32 this computes the hr half's input gradient too,
which the merged form must skip).  BatchNorm statistics at batch 32 are over all 32 samples here - the numbers are time only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from model import Discriminator

def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

def main():
    torch.manual_seed(0)
    D = Discriminator({"patch_size": 48, "spectral_norm": False}).cuda()
    x16a = torch.randint(0, 256, (16, 3, 192, 192)).float().cuda().contiguous(memory_format=torch.channels_last)
    x16b = torch.randint(0, 256, (16, 3, 192, 192)).float().cuda().contiguous(memory_format=torch.channels_last)
    x32 = torch.cat([x16a, x16b]).contiguous(memory_format=torch.channels_last)
    def fwd2():
        with torch.no_grad(): D(x16a); D(x16b)
    def fwd1():
        with torch.no_grad(): D(x32)
    def fb2():
        D.zero_grad(set_to_none=True)
        (D(x16a) - D(x16b)).sum().backward()
    def fb1():
        D.zero_grad(set_to_none=True)
        o = D(x32); (o[:16] - o[16:]).sum().backward()
    for p in D.parameters(): p.requires_grad = True
    print(f"forward only      : 2 x batch 16 {timeit(fwd2):7.3f} ms   1 x batch 32 {timeit(fwd1):7.3f} ms")
    print(f"fwd + bwd (D phase): 2 x batch 16 {timeit(fb2):7.3f} ms   1 x batch 32 {timeit(fb1):7.3f} ms")
    for p in D.parameters(): p.requires_grad = False
    xa = x16a.clone().requires_grad_(True); xc = x32.clone().requires_grad_(True)
    def gb2():
        o = D(xa)
        with torch.no_grad(): r = D(x16b)
        (o - r).sum().backward()
    def gb1():
        o = D(xc); (o[:16] - o[16:].detach()).sum().backward()
    print(f"fwd + input-gradient bwd (G phase): 2 x batch 16 (one without grad) {timeit(gb2):7.3f} ms   1 x batch 32 (full dgrad) {timeit(gb1):7.3f} ms")

if __name__ == "__main__":
    main()
