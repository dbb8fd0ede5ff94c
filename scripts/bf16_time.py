"""Time the bf16-mode conv kernels against the fp32 F(4,3) kernels on the network's shapes (same box, interleaved).
usage: python scripts/bf16_time.py [lib.so ...]   (default: the product library)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def timed(fn, reps=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def main():
    from pesr_amd import ops
    torch.manual_seed(0)
    shapes = [(16, 48, 48, 256, 256, False), (16, 48, 48, 256, 1024, True), (16, 96, 96, 256, 1024, True), (16, 96, 96, 128, 128, False),
              (16, 24, 24, 512, 512, False), (16, 192, 192, 64, 128, False)]
    for N, H, W, Cin, Cout, ps in shapes:
        NB = 3   # rotate the activations so that no call finds its input in the caches
        xs = [torch.randn(N, H, W, Cin, device="cuda") for _ in range(NB)]
        w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
        b = torch.randn(Cout, device="cuda")
        flops = 18.0 * N * H * W * Cin * Cout
        res = []
        for name, pack in (("fp32 F(4,3)", ops.pack_conv3x3_wino4), ("bf16", ops.pack_conv3x3_bf16)):
            wp = pack(w, 0, ps)
            bb = ops.pack_bias_ps(b) if ps else b
            i = [0]
            def f():
                i[0] += 1
                ops.conv3x3_fwd(xs[i[0] % NB], wp, bb, Cout, act=ops.ACT_RELU if not ps else ops.ACT_NONE, ps_out=ps)
            med, mn = timed(f)
            res.append(f"{name}: {med:7.1f} us (min {mn:7.1f}) {flops / med / 1e6:7.1f} TF/s")
        print(f"fwd {N}x{H}x{W} {Cin}->{Cout}{' ps' if ps else ''}:  " + "  |  ".join(res), flush=True)
    # input gradient with ReLU mask + skip (a ResBlock's first conv), G-body shape
    N, H, W, C = 16, 48, 48, 256
    dys = [torch.randn(N, H, W, C, device="cuda") for _ in range(3)]
    xin = torch.randn(N, H, W, C, device="cuda")
    w = torch.randn(C, C, 3, 3, device="cuda") * 0.05
    for name, pack in (("fp32 F(4,3)", ops.pack_conv3x3_wino4), ("bf16", ops.pack_conv3x3_bf16)):
        wp = pack(w, 1)
        i = [0]
        def f():
            i[0] += 1
            ops.conv3x3_dgrad(dys[i[0] % 3], wp, (N, H, W, C), mask=xin, skip=xin)
        med, mn = timed(f)
        print(f"dgrad+mask+skip body {name}: {med:7.1f} us (min {mn:7.1f})", flush=True)


def wgrad_part():
    from pesr_amd import ops
    for N, H, W, Cin, Cout, ps in [(16, 48, 48, 256, 256, False), (16, 48, 48, 256, 1024, True), (16, 96, 96, 256, 1024, True), (16, 96, 96, 128, 128, False)]:
        xs = [torch.randn(N, H, W, Cin, device="cuda") for _ in range(3)]
        dys = [torch.randn(N, 2 * H, 2 * W, Cout // 4, device="cuda") if ps else torch.randn(N, H, W, Cout, device="cuda") for _ in range(3)]
        flops = 18.0 * N * H * W * Cin * Cout
        res = []
        i = [0]
        def f32():
            i[0] += 1
            ops.conv3x3_wgrad(xs[i[0] % 3], dys[i[0] % 3], 1, ps_in=ps, algo=ops.WGRAD_AUTO)
        def b16():
            i[0] += 1
            ops.conv3x3_wgrad_bf16(xs[i[0] % 3], dys[i[0] % 3], ps_in=ps)
        for name, f in (("fp32 F(4,3)", f32), ("bf16", b16)):
            med, mn = timed(f)
            res.append(f"{name}: {med:7.1f} us (min {mn:7.1f}) {flops / med / 1e6:7.1f} TF/s")
        print(f"wgrad {N}x{H}x{W} {Cin}->{Cout}{' ps' if ps else ''}:  " + "  |  ".join(res), flush=True)


if __name__ == "__main__":
    libs = sys.argv[1:] or [None]
    for lib in libs:
        if lib:
            os.environ["PESR_HIP_LIB"] = lib
            print("==", lib, flush=True)
        if len(libs) > 1:
            import subprocess
            subprocess.run([sys.executable, __file__], env=dict(os.environ), check=False)
        else:
            main()
            wgrad_part()
