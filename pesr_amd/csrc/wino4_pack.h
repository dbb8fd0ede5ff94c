// Element e of the transformed, packed 1-D Winograd F(4,3) weights out[(ky*6 + xi)][c][n][k] (conv3x3_wino4.hip):
//   mode 0 (forward): g[kx] = w[o = n][i = 16c+k][ky][kx]
//   mode 1 (dgrad)  : g[kx] = w[o = 16c+k][i = n][2-ky][2-kx]   (the input gradient is the conv with the flipped kernel)
//   U = G g with the interpolation points 0, +-1, +-2, inf:
//       U0 = g0/4, U1 = -(g0+g1+g2)/6, U2 = -(g0-g1+g2)/6, U3 = (g0+2g1+4g2)/24, U4 = (g0-2g1+4g2)/24, U5 = g2
// No LDS swizzle: the kernel loads its B fragments straight from global memory (lane (r, g) reads the 16 bytes k = 4g..4g+3 of
// row n = r of its 16-channel block: one contiguous KiB per wave and slab).
#pragma once
// ps = 1: the conv feeds nn.PixelShuffle(2); its output channels are ordered sub-pixel-major like pack.hip does
//         (packed p = sub*C + cc  <->  original o = 4*cc + sub, C = O/4).
__device__ __forceinline__ float pesr_wino4_pack_elem(const float* __restrict__ w, int O, int I, int mode, int ps, long e) {
    const int R = mode == 0 ? I : O, Nn = mode == 0 ? O : I;
    const int k = (int)(e & 15);
    long rest = e >> 4;
    const int n = (int)(rest % Nn); rest /= Nn;
    const int c = (int)(rest % (R >> 4));
    const int t18 = (int)(rest / (R >> 4));
    const int ky = t18 / 6, xi = t18 - ky * 6;
    const int red = c * 16 + k;
    int o = mode == 0 ? n : red;
    const int i = mode == 0 ? red : n;
    if (ps) { const int C = O >> 2; const int sub = o / C, cc = o - sub * C; o = 4 * cc + sub; }
    const float* g = w + ((long)o * I + i) * 9 + (mode == 0 ? ky : 2 - ky) * 3;
    const float g0 = mode == 0 ? g[0] : g[2], g1 = g[1], g2 = mode == 0 ? g[2] : g[0];
    switch (xi) {
        case 0: return 0.25f * g0;
        case 1: return ((g0 + g2) + g1) * (-1.0f / 6.0f);
        case 2: return ((g0 + g2) - g1) * (-1.0f / 6.0f);
        case 3: return ((g0 + 4.0f * g2) + 2.0f * g1) * (1.0f / 24.0f);
        case 4: return ((g0 + 4.0f * g2) - 2.0f * g1) * (1.0f / 24.0f);
        default: return g2;
    }
}
