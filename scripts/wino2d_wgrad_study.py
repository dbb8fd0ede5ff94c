"""Go / no-go numerics of the WEIGHT GRADIENT in a nested 2-D Winograd form, F(2,3) along y times F(4,3) along x (24 products per
2 x 4 output pixels = 2/3 of the 1-D F(4,3) kernel's 36, 1/3 of the direct form's 72), for the G-body shape: K = 16 x 48 x 48
pixels summed per weight.  CPU emulation in fp32 (transforms in float32; the products accumulated in float32 in the order the
kernels use: sequentially over the tiles of a split-K slice, the 8 slices summed afterwards) against an fp64 weight gradient.
    python scripts/wino2d_wgrad_study.py"""
import numpy as np
from wino2d_study import AT2, AT4, BT2, BT4, G2, G4, f32


def seq_accumulate(terms, nsplit=8):
    """sum over axis 0 in float32, sequentially inside each of nsplit contiguous slices, then the slices in order."""
    parts = []
    for sl in np.array_split(np.arange(terms.shape[0]), nsplit):
        acc = np.zeros(terms.shape[1:], dtype=f32)
        for i in sl:
            acc = (acc + terms[i]).astype(f32)
        parts.append(acc)
    out = np.zeros(terms.shape[1:], dtype=f32)
    for p in parts:
        out = (out + p).astype(f32)
    return out


def run(N, C, K, H, W, kind, seed=0):
    rng = np.random.default_rng(seed)
    x = rng.uniform(-1, 1, (N, C, H + 2, W + 2))
    if kind == "relu":
        x = np.maximum(x, 0) * 2
    x[:, :, 0, :] = x[:, :, -1, :] = 0; x[:, :, :, 0] = x[:, :, :, -1] = 0
    dy = rng.uniform(-1, 1, (N, K, H, W))
    dw64 = np.zeros((K, C, 3, 3))
    for ky in range(3):
        for kx in range(3):
            dw64[:, :, ky, kx] = np.einsum("nkhw,nchw->kc", dy, x[:, :, ky:ky + H, kx:kx + W])
    x32, dy32 = x.astype(f32), dy.astype(f32)
    out = {}
    # direct fp32: per tap, sequential over (n, row) blocks (one MFMA k-run = a row here)
    t = np.stack([np.stack([np.einsum("khw,chw->kc", dy32[n], x32[n, :, ky:ky + H, kx:kx + W]).astype(f32)
                            for ky in range(3) for kx in range(3)]) for n in range(N)])          # [N, 9, K, C] (rows summed by einsum)
    out["direct fp32"] = seq_accumulate(t, min(8, N)).reshape(3, 3, K, C).transpose(2, 3, 0, 1)
    # 1-D F(4,3) along x: dM[k, row, t, xi] = A4 dy ; V[c, row, t, xi] ; dU[k, c, ky, xi] = sum dM[row] V[row + ky]
    AT4f, BT4f, G4f, AT2f, BT2f, G2f = (m.astype(f32) for m in (AT4, BT4, G4, AT2, BT2, G2))
    terms = []
    for n in range(N):
        for r in range(H):
            dyr = dy32[n, :, r, :].reshape(K, W // 4, 4)
            dM = np.einsum("px,ktp->ktx", AT4f, dyr).astype(f32)                                    # [K, T, 6]
            Vs = []
            for ky in range(3):
                row = x32[n, :, r + ky, :]
                d = np.stack([row[:, 4 * t_:4 * t_ + 6] for t_ in range(W // 4)], 1)                # [C, T, 6]
                Vs.append(np.einsum("xj,ctj->ctx", BT4f, d).astype(f32))
            for t_ in range(W // 4):                 # one MFMA k-step = one x-tile
                terms.append(np.stack([np.einsum("kx,cx->kcx", dM[:, t_], Vs[ky][:, t_]).astype(f32) for ky in range(3)], 2))
    dU = seq_accumulate(np.stack(terms))                                                             # [K, C, 3, 6]
    out["1-D F(4,3)x"] = np.einsum("xj,kcyx->kcyj", G4f, dU).astype(f32)
    # nested 2-D
    terms = []
    for n in range(N):
        for r in range(H // 2):
            for t_ in range(W // 4):
                dyt = dy32[n, :, 2 * r:2 * r + 2, 4 * t_:4 * t_ + 4]                                 # [K, 2, 4]
                dMx = np.einsum("px,kop->kox", AT4f, dyt).astype(f32)
                dM = np.einsum("oa,kox->kax", AT2f, dMx).astype(f32)                                 # [K, 4, 6]
                d = x32[n, :, 2 * r:2 * r + 4, 4 * t_:4 * t_ + 6]
                Vy = np.einsum("ai,cij->caj", BT2f, d).astype(f32)
                V = np.einsum("xj,caj->cax", BT4f, Vy).astype(f32)
                terms.append(np.einsum("kax,cax->kcax", dM, V).astype(f32))
    dU2 = seq_accumulate(np.stack(terms))                                                            # [K, C, 4, 6]
    t1 = np.einsum("xj,kcax->kcaj", G4f, dU2).astype(f32)
    out["2-D F(2,3)y x F(4,3)x"] = np.einsum("ai,kcaj->kcij", G2f, t1).astype(f32)
    mx = np.abs(dw64).max()
    return {k: float(np.abs(v.astype(np.float64) - dw64).max() / mx) for k, v in out.items()}


if __name__ == "__main__":
    print("weight gradient: max |error| / max |dw| vs fp64; fp32 emulation, 8 x 8 (Cout x Cin) weights")
    names = ("direct fp32", "1-D F(4,3)x", "2-D F(2,3)y x F(4,3)x")
    print(f"{'pixels':>14} {'input':>8} | " + " | ".join(f"{n:>22}" for n in names))
    for (N, H, W) in ((2, 48, 48), (16, 48, 48)):
        for kind in ("uniform", "relu"):
            e = run(N, 8, 8, H, W, kind)
            print(f"{N:>3} x {H} x {W:<3} {kind:>8} | " + " | ".join(f"{e[n]:>22.2e}" for n in names))
