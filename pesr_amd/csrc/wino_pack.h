// Element e of the transformed, packed Winograd weights out[(ky*4 + xi)][c][n][k] (conv3x3_wino.hip):
//   mode 0 (forward): g[kx] = w[o = n][i = 16c+k][ky][kx]
//   mode 1 (dgrad)  : g[kx] = w[o = 16c+k][i = n][2-ky][2-kx]   (the input gradient is the conv with the flipped kernel)
//   U = [g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2]
#pragma once
// ps = 1: the conv feeds nn.PixelShuffle(2); its output channels are ordered sub-pixel-major like pack.hip does
//         (packed p = sub*C + cc  <->  original o = 4*cc + sub, C = O/4).
__device__ __forceinline__ float pesr_wino_pack_elem(const float* __restrict__ w, int O, int I, int mode, int ps, long e) {
    const int R = mode == 0 ? I : O, Nn = mode == 0 ? O : I;
    // bank swizzle of the LDS image (the slab is DMA'd verbatim): the 16-byte k-group kg of row n sits at position kg ^ ((n>>1)&3),
    // so the 16 rows a fragment read touches per k-slot fall on 8 different 16-byte slots of the 128-byte LDS window
    const int n_ = (int)((e >> 4) % (mode == 0 ? O : I));
    const int kpos = (int)(e & 15);
    const int k = ((((kpos >> 2) ^ ((n_ >> 1) & 3)) << 2) | (kpos & 3));
    long rest = e >> 4;
    const int n = (int)(rest % Nn); rest /= Nn;
    const int c = (int)(rest % (R >> 4));
    const int t12 = (int)(rest / (R >> 4));
    const int ky = t12 >> 2, xi = t12 & 3;
    const int red = c * 16 + k;
    int o = mode == 0 ? n : red;
    const int i = mode == 0 ? red : n;
    if (ps) { const int C = O >> 2; const int sub = o / C, cc = o - sub * C; o = 4 * cc + sub; }
    const float* g = w + ((long)o * I + i) * 9 + (mode == 0 ? ky : 2 - ky) * 3;
    const float g0 = mode == 0 ? g[0] : g[2], g1 = g[1], g2 = mode == 0 ? g[2] : g[0];
    if (xi == 0) return g0;
    if (xi == 1) return 0.5f * ((g0 + g1) + g2);
    if (xi == 2) return 0.5f * ((g0 - g1) + g2);
    return g2;
}
