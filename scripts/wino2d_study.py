"""Go / no-go numerics of a nested 2-D Winograd F(4,3) along x times F(2,3) along y for the K1 shape:
24 multiplies per 2 x 4 output pixels = 1/3 of the direct conv's 72 (the 1-D F(4,3) kernels issue 1/2).  CPU emulation in
fp32 (transforms and channel accumulation in float32, like the kernels: transforms on the VALU, products accumulated by the
fp32 MFMA) against an fp64 direct convolution.  Prints the error table that DESIGN.md quotes.   python scripts/wino2d_study.py"""
import numpy as np

f32 = np.float32
BT4 = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=np.float64)
G4 = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=np.float64)
AT4 = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=np.float64)
BT2 = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)
G2 = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=np.float64)
AT2 = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)


def mm32(a, b):
    """a @ b with both operands and the result rounded to fp32 (a small transform: per-element error of a few ulp)."""
    return (a.astype(f32) @ b.astype(f32)).astype(f32)


def run(C, K, H, W, kind, seed=0):
    rng = np.random.default_rng(seed)
    x = rng.uniform(-1, 1, (C, H + 2, W + 2))
    if kind == "relu":
        x = np.maximum(x, 0) * 2
    x[:, 0, :] = x[:, -1, :] = 0; x[:, :, 0] = x[:, :, -1] = 0          # zero padding baked in
    w = rng.uniform(-1, 1, (K, C, 3, 3)) / np.sqrt(9 * C)
    # fp64 truth
    y64 = np.zeros((K, H, W))
    for ky in range(3):
        for kx in range(3):
            y64 += np.einsum("kc,chw->khw", w[:, :, ky, kx], x[:, ky:ky + H, kx:kx + W])
    x32, w32 = x.astype(f32), w.astype(f32)
    out = {}
    # direct fp32
    yd = np.zeros((K, H, W), dtype=f32)
    for ky in range(3):
        for kx in range(3):
            yd += np.einsum("kc,chw->khw", w32[:, :, ky, kx], x32[:, ky:ky + H, kx:kx + W]).astype(f32)
    out["direct fp32"] = yd
    # 1-D F(4,3) along x: U[k,c,ky,xi] = G4 g ; V[c,row,t,xi] = BT4 d
    U = np.einsum("xj,kcyj->kcyx", G4.astype(f32), w32).astype(f32)
    y1 = np.zeros((K, H, W), dtype=f32)
    for t in range(W // 4):
        d = x32[:, :, 4 * t:4 * t + 6]                                  # [C, H+2, 6]
        V = np.einsum("xj,chj->chx", BT4.astype(f32), d).astype(f32)    # [C, H+2, 6]
        M = np.zeros((K, H, 6), dtype=f32)
        for ky in range(3):
            M += np.einsum("kcx,chx->khx", U[:, :, ky, :], V[:, ky:ky + H, :]).astype(f32)
        y1[:, :, 4 * t:4 * t + 4] = np.einsum("oj,khj->kho", AT4.astype(f32), M).astype(f32)
    out["1-D F(4,3)x"] = y1
    # nested 2-D: F(2,3) along y x F(4,3) along x
    U2 = np.einsum("ai,xj,kcij->kcax", G2.astype(f32), G4.astype(f32), w32).astype(f32)     # [K, C, 4, 6]
    y2 = np.zeros((K, H, W), dtype=f32)
    for r in range(H // 2):
        for t in range(W // 4):
            d = x32[:, 2 * r:2 * r + 4, 4 * t:4 * t + 6]                # [C, 4, 6]
            Vy = np.einsum("ai,cij->caj", BT2.astype(f32), d).astype(f32)
            V = np.einsum("xj,caj->cax", BT4.astype(f32), Vy).astype(f32)
            M = np.einsum("kcax,cax->kax", U2, V).astype(f32)          # 24 products per (k, tile), summed over c in fp32
            Yy = np.einsum("oa,kax->kox", AT2.astype(f32), M).astype(f32)
            y2[:, 2 * r:2 * r + 2, 4 * t:4 * t + 4] = np.einsum("pj,koj->kop", AT4.astype(f32), Yy).astype(f32)
    out["2-D F(2,3)y x F(4,3)x"] = y2
    # nested 2-D F(2,3) x F(2,3) (4/9 of the multiplies) for comparison
    U3 = np.einsum("ai,bj,kcij->kcab", G2.astype(f32), G2.astype(f32), w32).astype(f32)
    y3 = np.zeros((K, H, W), dtype=f32)
    for r in range(H // 2):
        for t in range(W // 2):
            d = x32[:, 2 * r:2 * r + 4, 2 * t:2 * t + 4]
            V = np.einsum("ai,bj,cij->cab", BT2.astype(f32), BT2.astype(f32), d).astype(f32)
            M = np.einsum("kcab,cab->kab", U3, V).astype(f32)
            y3[:, 2 * r:2 * r + 2, 2 * t:2 * t + 2] = np.einsum("oa,pb,kab->kop", AT2.astype(f32), AT2.astype(f32), M).astype(f32)
    out["2-D F(2,3) x F(2,3)"] = y3
    mx = np.abs(y64).max()
    return {k: float(np.abs(v.astype(np.float64) - y64).max() / mx) for k, v in out.items()}


if __name__ == "__main__":
    print("max |error| / max |y|  vs fp64, fp32 emulation (numpy float32 transforms and accumulation), Cout 16, 8 x 48 pixels")
    print(f"{'Cin':>5} {'input':>8} | " + " | ".join(f"{n:>22}" for n in ("direct fp32", "1-D F(4,3)x", "2-D F(2,3)y x F(4,3)x", "2-D F(2,3) x F(2,3)")))
    for C in (64, 256, 512):
        for kind in ("uniform", "relu"):
            worst = {}
            for seed in range(3):
                e = run(C, 16, 8, 48, kind, seed)
                for k, v in e.items():
                    worst[k] = max(worst.get(k, 0), v)
            print(f"{C:>5} {kind:>8} | " + " | ".join(f"{worst[n]:>22.2e}" for n in ("direct fp32", "1-D F(4,3)x", "2-D F(2,3)y x F(4,3)x", "2-D F(2,3) x F(2,3)")))
