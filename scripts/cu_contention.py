"""What a communication kernel holding k CUs costs the 256-workgroup body kernels: a hog kernel sits
on k CUs (scripts/cu_hog.hip, on a second stream) while the K1-shaped forward / input-gradient / weight-gradient launches are
timed with HIP events on the compute stream.  Table -> stdout (commit under profiles/).
    hipcc -O3 --offload-arch=gfx950 -fPIC -shared scripts/cu_hog.hip -o exp/libcuhog.so ; python scripts/cu_contention.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pesr_amd import ops

hog = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "exp", "libcuhog.so"))
hog.cu_hog.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda")
N, H, W, C = 16, 48, 48, 256
g = torch.Generator().manual_seed(0)
x = (torch.rand(N, H, W, C, generator=g) * 2 - 1).to(dev)
dy = (torch.rand(N, H, W, C, generator=g) * 2 - 1).to(dev)
w = ((torch.rand(C, C, 3, 3, generator=g) * 2 - 1) * 0.05).to(dev)
b = torch.zeros(C, device=dev)
wf, wd = ops.pack_conv3x3_wino4(w, 0), ops.pack_conv3x3_wino4(w, 1)
kernels = {"fwd": lambda: ops.conv3x3_fwd(x, wf, b, C, act=ops.ACT_RELU),
           "dgrad": lambda: ops.conv3x3_dgrad(dy, wd, (N, H, W, C), mask=x),
           "wgrad": lambda: ops.conv3x3_wgrad(x, dy, 1)}
# a many-workgroup kernel for comparison: VGG conv1_2 shape (16 x 192 x 192 x 64 -> 64): 1024 tiles
x2 = (torch.rand(16, 192, 192, 64, generator=g) * 2 - 1).to(dev)
w2 = ((torch.rand(64, 64, 3, 3, generator=g) * 2 - 1) * 0.05).to(dev)
w2f = ops.pack_conv3x3_wino4(w2, 0)
kernels["fwd 192x192x64 (1024 workgroups)"] = lambda: ops.conv3x3_fwd(x2, w2f, None, 64, act=ops.ACT_RELU)
side = torch.cuda.Stream()
where = torch.zeros(256, dtype=torch.int32, device=dev)
REPS = 12


def timed(fn, k, lds):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    if k:
        hog.cu_hog(k, lds, 12000, where.data_ptr(), side.cuda_stream)     # 12 ms: covers the timed launches
        torch.cuda._sleep(200000)                                          # let the hog workgroups land first
    e = [torch.cuda.Event(enable_timing=True) for _ in range(REPS + 1)]
    e[0].record()
    for i in range(REPS):
        fn()
        e[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(e[i].elapsed_time(e[i + 1]) * 1e3 for i in range(REPS))
    return ts[len(ts) // 2]


print(f"{'kernel':40s} {'hog':>22s} " + " ".join(f"k={k:<3d}   " for k in (0, 1, 4, 8, 16, 32)))
for name, fn in kernels.items():
    for lds, label in ((100 * 1024, "exclusive (100 KiB LDS)"), (1024, "light (1 KiB LDS)")):
        row = [timed(fn, k, lds) for k in (0, 1, 4, 8, 16, 32)]
        print(f"{name:40s} {label:>22s} " + " ".join(f"{t:7.1f}us" for t in row) + f"   k=8 / k=0: {row[3] / row[0]:.2f}")
xccs = sorted(set(int(v) >> 16 for v in where.cpu().tolist()[:32]))
print("XCCs the last 32 hog workgroups landed on:", xccs)
