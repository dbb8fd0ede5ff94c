// Weight packing for the MFMA conv kernels (gfx950).
//
// The nn.Module boundary keeps the reference's OIHW fp32 parameters (state_dict schema of
// reference model/pesr.py, model/basic.py).  The kernels want one contiguous [N][16] block per
// (tap, 16-channel chunk of the reduction dimension):
//   mode 0 (forward):  out[t][c][n][k] = w[o = unperm(n)][i = 16c+k][ky][kx],           t = 3ky+kx
//   mode 1 (dgrad):    out[t][c][n][k] = w[o = unperm(16c+k)][i = n][ky][kx]            (reduce over o)
// `ps` = 1 orders the output channels of a conv that feeds nn.PixelShuffle(2) sub-pixel-major:
// packed channel p = (2*si+sj)*C + cc  <->  original channel o = 4*cc + 2*si + sj, so the conv
// epilogue can write the shuffled tensor with contiguous channel runs (reference model/basic.py:56-59).
#include "common.h"
#include "launchers.h"
#include "wino_pack.h"
#include "wino4_pack.h"

// R (reduction channels) is zero-padded to a multiple of 16 and Nn ("n" channels) to 16 (if <= 16) or a multiple of 64, so
// the 3-channel RGB layers run on the same MFMA kernels.
__global__ void pack_conv3x3_kernel(const float* __restrict__ w, float* __restrict__ out, int O, int I, int mode, int ps,
                                    int R, int Nn) {
    const long total = 9L * R * Nn;
    const int C = O >> 2;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int k = (int)(e & 15);
        long rest = e >> 4;
        const int n = (int)(rest % Nn); rest /= Nn;
        const int c = (int)(rest % (R >> 4));
        const int t = (int)(rest / (R >> 4));
        const int red = c * 16 + k;
        int o = mode == 0 ? n : red;
        const int i = mode == 0 ? red : n;
        if (ps) { const int sub = o / C, cc = o - sub * C; o = 4 * cc + sub; }
        out[e] = (o < O && i < I) ? w[((long)o * I + i) * 9 + t] : 0.f;
    }
}

// Batched form: one launch packs many convs (all of a network's, right after its optimizer step).  desc[d] (8 int64):
// {src ptr, dst ptr, O, I, mode, ps, R, Nn}; blockIdx.y = d.  mode 2 / 3: the Winograd F(2,3) packing of mode 0 / 1, 4 / 5: F(4,3),
// 7 / 8: bf16, 9 / 10: split-bf16 (hi plane + lo plane).
__global__ void pack_conv3x3_batched_kernel(const long long* __restrict__ desc) {
    const long long* d = desc + (size_t)blockIdx.y * 8;
    const float* __restrict__ w = (const float*)d[0];
    float* __restrict__ out = (float*)d[1];
    const int O = (int)d[2], I = (int)d[3], mode = (int)d[4], ps = (int)d[5], R = (int)d[6], Nn = (int)d[7];
    if (mode >= 7) {   // bf16 packing (conv3x3_bf16.hip): mode 7 = forward, 8 = dgrad; out[t][c][n][k], 32-channel chunks;
        //                    9 / 10: the split-bf16 packing (conv3x3_bf16x3.hip): a hi plane (= modes 7 / 8) and a lo plane behind it
        const bool split = mode >= 9;
        const int m = split ? mode - 9 : mode - 7;
        const int Rr = m == 0 ? I : O, Nr = m == 0 ? O : I;
        __bf16* const ob = (__bf16*)out;
        const long total_b = 9L * Rr * Nr;
        for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total_b; e += (long)gridDim.x * blockDim.x) {
            const int k = (int)(e & 31);
            long rest = e >> 5;
            const int n = (int)(rest % Nr); rest /= Nr;
            const int c = (int)(rest % (Rr >> 5));
            const int t = (int)(rest / (Rr >> 5));
            const int red = c * 32 + k;
            int o = m == 0 ? n : red;
            const int i = m == 0 ? red : n;
            if (ps) { const int C = O >> 2; const int sub = o / C, cc = o - sub * C; o = 4 * cc + sub; }
            const float v = w[((long)o * I + i) * 9 + (m == 0 ? t : 8 - t)];
            const __bf16 h = (__bf16)v;
            ob[e] = h;
            if (split) ob[total_b + e] = (__bf16)(v - (float)h);
        }
        return;
    }
    if (mode >= 4) {   // Winograd F(4,3) packing (conv3x3_wino4.hip): mode 4 = forward, 5 = dgrad
        // One thread per (n, k) of a 16 x 16 tile of (output row n, reduction channel 16c + k): it reads the nine taps of its
        // (o, i) pair once (36 contiguous bytes; a tile row is 576 contiguous bytes) and writes all 18 transformed values, each
        // into a slab where the tile's 256 (n, k) entries are ONE contiguous KiB.  (One thread per output element read every
        // tap six times over and spent its time in index arithmetic: 0.8 ms per GAN step for the ~130 packings.)
        const int m = mode - 4;
        const int Rr = m == 0 ? I : O, Nr = m == 0 ? O : I;          // reduction / row extents (multiples of 16)
        const int rc = Rr >> 4, nt = Nr >> 4;
        const int n_l = threadIdx.x >> 4, k = threadIdx.x & 15;
        for (int tile = blockIdx.x; tile < rc * nt; tile += gridDim.x) {
            const int c = tile % rc, n = (tile / rc) * 16 + n_l;
            const int red = c * 16 + k;
            int o = m == 0 ? n : red;
            const int i = m == 0 ? red : n;
            if (ps) { const int C = O >> 2; const int sub = o / C, cc = o - sub * C; o = 4 * cc + sub; }
            const float* g = w + ((long)o * I + i) * 9;
            float t[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) t[q] = g[q];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int sy = m == 0 ? ky : 2 - ky;
                const float g0 = m == 0 ? t[sy * 3] : t[sy * 3 + 2], g1 = t[sy * 3 + 1], g2 = m == 0 ? t[sy * 3 + 2] : t[sy * 3];
                float* const ob = out + (((long)(ky * 6) * rc + c) * Nr + n) * 16 + k;
                const long xs = (long)rc * Nr * 16;                  // floats between consecutive xi slabs
                ob[0] = 0.25f * g0;
                ob[xs] = ((g0 + g2) + g1) * (-1.0f / 6.0f);
                ob[2 * xs] = ((g0 + g2) - g1) * (-1.0f / 6.0f);
                ob[3 * xs] = ((g0 + 4.0f * g2) + 2.0f * g1) * (1.0f / 24.0f);
                ob[4 * xs] = ((g0 + 4.0f * g2) - 2.0f * g1) * (1.0f / 24.0f);
                ob[5 * xs] = g2;
            }
        }
        return;
    }
    if (mode >= 2) {   // Winograd packing (conv3x3_wino.hip): mode 2 = forward, 3 = dgrad
        const long total_w = 12L * O * I;
        for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total_w; e += (long)gridDim.x * blockDim.x)
            out[e] = pesr_wino_pack_elem(w, O, I, mode - 2, ps, e);
        return;
    }
    const long total = 9L * R * Nn;
    const int C = O >> 2;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int k = (int)(e & 15);
        long rest = e >> 4;
        const int n = (int)(rest % Nn); rest /= Nn;
        const int c = (int)(rest % (R >> 4));
        const int t = (int)(rest / (R >> 4));
        const int red = c * 16 + k;
        int o = mode == 0 ? n : red;
        const int i = mode == 0 ? red : n;
        if (ps) { const int sub = o / C, cc = o - sub * C; o = 4 * cc + sub; }
        out[e] = (o < O && i < I) ? w[((long)o * I + i) * 9 + t] : 0.f;
    }
}

int pesr_pack_conv3x3_batched_launch(const long long* desc, int count, hipStream_t stream) {
    if (count < 1) return PESR_OK;
    hipLaunchKernelGGL(pack_conv3x3_batched_kernel, dim3(128, (unsigned)count), dim3(256), 0, stream, desc);
    return pesr_launch_status();
}

// permute a bias vector into the packed (pixel-shuffle) channel order
__global__ void pack_bias_ps_kernel(const float* __restrict__ b, float* __restrict__ out, int O) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= O) return;
    const int C = O >> 2;
    const int sub = p / C, cc = p - sub * C;
    out[p] = b[4 * cc + sub];
}

int pesr_pack_conv3x3_launch(const float* w, float* out, int O, int I, int mode, int ps, hipStream_t stream) {
    const int R = (((mode == 0 ? I : O) + 15) / 16) * 16;
    const int Nr = mode == 0 ? O : I;
    const int Nn = Nr <= 16 ? 16 : ((Nr + 63) / 64) * 64;
    if (ps && (O % 64 || I % 16)) return PESR_EINVAL;
    const long total = 9L * R * Nn;
    const int block = 256;
    const int grid = (int)((total + block - 1) / block < 4096 ? (total + block - 1) / block : 4096);
    hipLaunchKernelGGL(pack_conv3x3_kernel, dim3(grid), dim3(block), 0, stream, w, out, O, I, mode, ps, R, Nn);
    return pesr_launch_status();
}

int pesr_pack_bias_ps_launch(const float* b, float* out, int O, hipStream_t stream) {
    hipLaunchKernelGGL(pack_bias_ps_kernel, dim3((O + 255) / 256), dim3(256), 0, stream, b, out, O);
    return pesr_launch_status();
}
