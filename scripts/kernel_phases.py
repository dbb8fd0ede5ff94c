"""Phase breakdown of the two F(4,3) kernels at the K1 shape from in-kernel real-time stamps (100 MHz counter).

Needs the diagnostic build made by scripts/build_timing.sh (both kernels with -DPESR_TIMING in exp/libtiming.so); the
product library has no stamps.
usage: bash scripts/build_timing.sh && PESR_HIP_LIB=exp/libtiming.so python scripts/kernel_phases.py
"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pesr_amd import ops, _lib

SLOTS = 8


def stamps(fn_name, wgs):
    lib = _lib.lib()
    buf = (ctypes.c_ulonglong * (wgs * SLOTS))()
    fn = getattr(lib, fn_name)
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
    rc = fn(buf, wgs * SLOTS)
    assert rc == 0, rc
    return np.frombuffer(buf, dtype=np.uint64).reshape(wgs, SLOTS).astype(np.int64)


def report(name, t, labels):
    t0 = t[:, 0].min()
    us = (t - t0) / 100.0
    print(f"== {name}: {t.shape[0]} workgroups; kernel span {us[:, len(labels) - 1].max():.1f} us (first entry -> last exit)")
    print(f"   entry skew: median {np.median(us[:, 0]):.2f} us, max {us[:, 0].max():.2f} us")
    for i in range(1, len(labels)):
        d = us[:, i] - us[:, i - 1]
        print(f"   {labels[i]:<34} median {np.median(d):7.2f} us   min {d.min():7.2f}   max {d.max():7.2f}")
    ghz = (t[:, 7] - t[:, 6]) / np.maximum(t[:, len(labels) - 1] - t[:, 0], 1) * 0.1
    print(f"   in-kernel clock (s_memtime / s_memrealtime): median {np.median(ghz):.3f} GHz, min {ghz.min():.3f}, max {ghz.max():.3f}")
    print(f"   exit: median {np.median(us[:, len(labels) - 1]):.2f} us, min {us[:, len(labels) - 1].min():.2f}, max {us[:, len(labels) - 1].max():.2f}")


def main():
    torch.manual_seed(0)
    N, H, W, C = 16, 48, 48, 256
    x = torch.rand(N, H, W, C, device="cuda") - 0.5
    w = (torch.rand(C, C, 3, 3, device="cuda") - 0.5) * 0.1
    b = torch.rand(C, device="cuda")
    dy = torch.rand(N, H, W, C, device="cuda") - 0.5
    w4 = ops.pack_conv3x3_wino4(w, 0)
    for _ in range(20):
        ops.conv3x3_fwd(x, w4, b, C, act=ops.ACT_RELU)
    torch.cuda.synchronize()
    report("conv3x3_wino4_kernel", stamps("pesr_debug_timing_wino4", 256),
           ["entry", "prologue (chunk 0 staged)", "main loop (16 chunks)", "partial exchange", "stores issued"])
    for _ in range(20):
        ops.conv3x3_wgrad(x, dy)
    torch.cuda.synchronize()
    t = stamps("pesr_debug_timing_wgrad4", 256)
    report("conv3x3_wgrad_wino4_kernel", t, ["entry", "strip start (4 rows staged)", "main loop (48 segments)", "G^T + slab store"])
    d = (t[:, 5] - t[:, 4]) / 100.0
    print(f"   mid-loop strip change: median {np.median(d):.2f} us, max {d.max():.2f}")


if __name__ == "__main__":
    main()
