"""Go / no-go numerics of a SPLIT-bf16 3x3 conv (SURVEY 8 f4 "bf16/split-bf16 MFMA"): every fp32 operand is
written as hi + lo with hi = bf16(v), lo = bf16(v - hi), and a product a*b is replaced by the three bf16 MFMA products
a_hi*b_hi + a_hi*b_lo + a_lo*b_hi (the lo*lo term, ~2^-18 of the product, is dropped), each exact in fp32, summed in fp32.
CPU emulation against an fp64 direct convolution, next to the fp32 direct conv, the 1-D F(4,3) kernels' arithmetic, the plain
bf16 mode and a 4-term split (with lo*lo) - the error table DESIGN.md quotes.        python scripts/split_bf16_study.py"""
import numpy as np
import torch

f32 = np.float32
BT4 = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=np.float64)
G4 = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=np.float64)
AT4 = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=np.float64)


def bf16(a):
    """round to nearest even to bfloat16, returned as float32 (what v_cvt_pk_bf16_f32 does)"""
    return torch.from_numpy(np.ascontiguousarray(a, dtype=f32)).to(torch.bfloat16).to(torch.float32).numpy()


def split(a):
    hi = bf16(a)
    lo = bf16(a.astype(f32) - hi)
    return hi, lo


def conv_taps(xs, ws, H, W, acc_dtype=f32):
    """sum over the nine taps of einsum(w[k,c], x[c,h,w]); operands fp32-valued, products exact, running sum in acc_dtype"""
    K = ws.shape[0]
    y = np.zeros((K, H, W), dtype=acc_dtype)
    for ky in range(3):
        for kx in range(3):
            y += np.einsum("kc,chw->khw", ws[:, :, ky, kx], xs[:, ky:ky + H, kx:kx + W]).astype(acc_dtype)
    return y


def run(C, K, H, W, kind, seed=0):
    rng = np.random.default_rng(seed)
    x = rng.uniform(-1, 1, (C, H + 2, W + 2))
    if kind == "relu":
        x = np.maximum(x, 0) * 2
    x[:, 0, :] = x[:, -1, :] = 0; x[:, :, 0] = x[:, :, -1] = 0
    w = rng.uniform(-1, 1, (K, C, 3, 3)) / np.sqrt(9 * C)
    y64 = conv_taps(x, w, H, W, np.float64)
    x32, w32 = x.astype(f32), w.astype(f32)
    out = {"direct fp32": conv_taps(x32, w32, H, W)}
    U = np.einsum("xj,kcyj->kcyx", G4.astype(f32), w32).astype(f32)
    y1 = np.zeros((K, H, W), dtype=f32)
    for t in range(W // 4):
        V = np.einsum("xj,chj->chx", BT4.astype(f32), x32[:, :, 4 * t:4 * t + 6]).astype(f32)
        M = np.zeros((K, H, 6), dtype=f32)
        for ky in range(3):
            M += np.einsum("kcx,chx->khx", U[:, :, ky, :], V[:, ky:ky + H, :]).astype(f32)
        y1[:, :, 4 * t:4 * t + 4] = np.einsum("oj,khj->kho", AT4.astype(f32), M).astype(f32)
    out["1-D F(4,3) fp32 (today)"] = y1
    xh, xl = split(x32)
    wh, wl = split(w32)
    out["bf16 (1 product)"] = conv_taps(xh, wh, H, W)
    # three products per chunk, accumulated into ONE fp32 accumulator like three back-to-back MFMAs would (order: small terms first)
    out["split-bf16 x3"] = conv_taps(xl, wh, H, W) + conv_taps(xh, wl, H, W) + conv_taps(xh, wh, H, W)
    out["split-bf16 x4"] = conv_taps(xl, wl, H, W) + out["split-bf16 x3"]
    # hi/lo split of the ACTIVATIONS only (weights split, activations split): same as x3; and weights-only split (x rounded once)
    out["split weights only x2"] = conv_taps(xh, wl, H, W) + conv_taps(xh, wh, H, W)
    scale = np.abs(y64).max()
    return {k: float(np.abs(v.astype(np.float64) - y64).max() / scale) for k, v in out.items()}


def main():
    K, H, W = 16, 8, 48
    cols = None
    print(f"max |error| / max |y| vs fp64; CPU emulation (operands rounded as stated, exact products, fp32 running sums), Cout {K}, {H} x {W} pixels")
    for C in (64, 256, 512):
        for kind in ("uniform", "relu"):
            r = run(C, K, H, W, kind)
            if cols is None:
                cols = list(r)
                print(f"{'Cin':>5} {'input':>8} | " + " | ".join(f"{c:>24}" for c in cols))
            print(f"{C:5d} {kind:>8} | " + " | ".join(f"{r[c]:24.2e}" for c in cols))


if __name__ == "__main__":
    main()
