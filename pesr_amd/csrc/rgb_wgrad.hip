// Weight gradients of the 3x3 convs that touch a 3-channel (RGB) tensor: 3 -> C (reference `embed`
// model/pesr.py:23, Discriminator features.0 model/pesr.py:53) and C -> 3 (Upsampler's last conv,
// model/basic.py:60).  Both are the same correlation of a C-channel tensor A with a 3-channel tensor B3:
//   corr[c][oy][ox][k] = sum_p A[p][c] * B3[p + (oy-1, ox-1)][k]          (zero outside the image)
//   mode 0 (3 -> C):  A = dy, B3 = x  : dw[c][k][ky][kx] = corr[c][ky][kx][k]
//   mode 1 (C -> 3):  A = x,  B3 = dy : dw[k][c][ky][kx] = corr[c][2-ky][2-kx][k]
// HBM-bound on reading A once (27 FMAs per loaded float).  One wave owns 64 channels and a range of
// image rows; the 27 B3 values of a pixel are wave-uniform, so they come through scalar loads from a
// zero-padded copy of B3 and feed v_fma as SGPR operands.  Per-wave partial sums are reduced by a
// second kernel in a fixed order (deterministic).
#include "common.h"
#include "launchers.h"
#include "reduce_rows.h"

// B3 [N][H][W][3] -> zero-padded [N][H+2][W+2][3]
__global__ void pad_rgb_kernel(const float* __restrict__ b3, float* __restrict__ out, int N, int H, int W) {
    const long total = (long)N * (H + 2) * (W + 2) * 3;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int k = (int)(e % 3);
        long rest = e / 3;
        const int x = (int)(rest % (W + 2)) - 1; rest /= (W + 2);
        const int y = (int)(rest % (H + 2)) - 1;
        const int n = (int)(rest / (H + 2));
        out[e] = (x >= 0 && x < W && y >= 0 && y < H) ? b3[(((long)n * H + y) * W + x) * 3 + k] : 0.f;
    }
}

// grid.x = (C/64) * nsplit blocks of ONE wave (64 threads); lane = channel
__global__ __launch_bounds__(64) void corr_rgb_kernel(const float* __restrict__ A, const float* __restrict__ b3p,
                                                      float* __restrict__ part, int NH, int H, int W, int C,
                                                      int rows_per_split) {
    const int ngroups = (C + 63) >> 6;
    const int cg = blockIdx.x % ngroups;
    const int sp = blockIdx.x / ngroups;
    const int c = cg * 64 + threadIdx.x;
    if (c >= C) return;
    const int r0 = sp * rows_per_split;
    int r1 = r0 + rows_per_split; if (r1 > NH) r1 = NH;
    float acc[27];
#pragma unroll
    for (int i = 0; i < 27; ++i) acc[i] = 0.f;
    const int WP = W + 2;
    for (int row = r0; row < r1; ++row) {
        const int n = row / H, y = row - n * H;
        const float* arow = A + (size_t)row * W * C + c;
        const float* brow = b3p + ((size_t)n * (H + 2) + y) * WP * 3;  // padded row y-1 of image n
        for (int x0 = 0; x0 < W; x0 += 4) {
            float a[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] = (x0 + u < W) ? arow[(size_t)(x0 + u) * C] : 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (x0 + u < W) {
                    const float* b = brow + (x0 + u) * 3;  // top-left neighbour (y-1, x-1) in padded coords
#pragma unroll
                    for (int oy = 0; oy < 3; ++oy)
#pragma unroll
                        for (int j = 0; j < 9; ++j) acc[oy * 9 + j] = fmaf(a[u], b[oy * WP * 3 + j], acc[oy * 9 + j]);
                }
            }
        }
    }
    float* o = part + ((size_t)sp * C + c) * 27;
#pragma unroll
    for (int i = 0; i < 27; ++i) o[i] = acc[i];
}

// MFMA form of the same correlation for C a multiple of 256: dw^T[c][n] = sum_p A[p][c] * b27[p][n] with n = oy*9 + ox*3 + k
// (27 of 32 columns) and K = pixels.  The point is the load pattern, not the flops: a lane reads 16 bytes = FOUR CONSECUTIVE
// CHANNELS of one pixel and hands component t to MFMA tile t, whose row i stands for channel 4i + t - so the 16 lanes of a
// k-slot cover 256 contiguous bytes and the four waves of a workgroup a pixel's whole 1-KiB channel vector (the VALU kernel
// above issues one 4-byte load per 27 FMAs and is bound by them).  The B operand - the 27 neighbourhood values of the pixel -
// is gathered from three zero-padded B3 rows staged in LDS.  One workgroup = 256 channels x rows_per_split image rows; wave w
// owns channels 64 w .. 64 w + 63 (4 M tiles x 2 N tiles); loads run CR_D - 1 k-steps (4 pixels each) ahead.
// CW = 4: the four waves own four 64-channel groups of the same pixels (C % 256 == 0).  CW = 1 (round 4; C % 64 == 0 - the
// 3 -> 64 layers, Discriminator features.0: 158 us on the VALU kernel above, 0.95 TB/s): the four waves own the SAME 64 channels
// and every fourth k-step each, so a wave-instruction still reads 1 KiB of contiguous memory (four pixels x 256 B); every wave
// writes its own partial.
constexpr int CR_D = 8;
template <int CW>
__global__ __launch_bounds__(256) void corr_rgb_mfma_kernel(const float* __restrict__ A, const float* __restrict__ b3p,
                                                            float* __restrict__ part, int NH, int H, int W, int C,
                                                            int rows_per_split) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [2][3][(W + 2) * 3] padded B3 rows, double buffered
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int ncol = C / (64 * CW);
    const int cc = blockIdx.x % ncol, sp = blockIdx.x / ncol;
    const int c0 = CW == 4 ? cc * 256 + wave * 64 : cc * 64;
    const int sw = CW == 4 ? 0 : wave, sm = CW == 4 ? 1 : 4;              // this wave's k-steps: sm * j + sw
    const int r0 = sp * rows_per_split;
    int r1 = r0 + rows_per_split; if (r1 > NH) r1 = NH;
    const int WP3 = (W + 2) * 3;
    const int ksteps_all = (W + 3) >> 2;
    const int ksteps = CW == 4 ? ksteps_all : (ksteps_all + 3) >> 2;       // per wave (CW = 1: steps past the row read zeros)
    f32x4 acc[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t) { acc[t][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[t][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    // B gather offsets of this lane inside a staged row triple: column n = i (N tile 0) and 16 + i (N tile 1, n < 27 only)
    const int n1 = 16 + i;
    const int boff0 = (i / 9) * WP3 + (i % 9);
    const int boff1 = n1 < 27 ? (n1 / 9) * WP3 + (n1 % 9) : -1;
    const unsigned row_bytes = (unsigned)W * C * 4;
    const unsigned a_lane = (unsigned)((c0 + 4 * i) * 4);                 // + pixel * C * 4
    // padded rows y .. y+2 of image n (= neighbours y-1 .. y+1) -> LDS.  Eight loads of a thread in flight at a time, and (in the row loop)
    // issued BEFORE the row's A loads and stored behind them: as a plain copy loop (one load, one wait, one LDS store per iteration) the
    // next row's staging stood in front of every row with seven dependent round trips (round 6).
    constexpr int SB = 8;
    const int nb3 = 3 * WP3;
    auto stage_src = [&](int row) -> const float* {
        const int n = row / H, y = row - n * H;
        return b3p + ((size_t)n * (H + 2) + y) * WP3;
    };
    auto stage_load = [&](const float* src, int base, float (&v)[SB]) {
#pragma unroll
        for (int k = 0; k < SB; ++k) { const int e = base + k * 256 + tid; v[k] = e < nb3 ? src[e] : 0.f; }
    };
    auto stage_store = [&](float* dst, int base, const float (&v)[SB]) {
#pragma unroll
        for (int k = 0; k < SB; ++k) { const int e = base + k * 256 + tid; if (e < nb3) dst[e] = v[k]; }
    };
    if (r0 < r1) {
        const float* src = stage_src(r0);
        for (int base = 0; base < nb3; base += SB * 256) { float v[SB]; stage_load(src, base, v); stage_store(lds, base, v); }
    }
    __syncthreads();
#pragma unroll 1
    for (int row = r0; row < r1; ++row) {
        const int buf = (row - r0) & 1;
        const bool more = row + 1 < r1;                                    // the next row's B3 rows: visible after the barrier at the end of this row
        const float* const nsrc = stage_src(more ? row + 1 : row);
        float* const ndst = lds + (buf ^ 1) * 3 * WP3;
        float sv[SB];
        if (more) stage_load(nsrc, 0, sv);
        const float* const bl = lds + buf * 3 * WP3;
        const unsigned long long av = (unsigned long long)(A + (size_t)row * W * C);      // provably wave-uniform base: no waterfall loops
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(av >> 32)) << 32) |
                    (unsigned)__builtin_amdgcn_readfirstlane((unsigned)av)),     // (unsigned): the builtin returns int - no sign extension
            0, row_bytes, 0x00020000);
        auto a_load = [&](int s) -> u32x4 {                                // this wave's k-step s: pixel 4 (sm s + sw) + g (beyond the row: zeros)
            const int px = 4 * (sm * s + sw) + g;
            return __builtin_amdgcn_raw_buffer_load_b128(rs, px < W ? a_lane + (unsigned)px * C * 4 : 0x80000000u, 0, 0);
        };
        u32x4 fa[CR_D];
#pragma unroll
        for (int d = 0; d < CR_D - 1; ++d) fa[d] = a_load(d);              // s >= ksteps: all lanes out of range, harmless
        if (more) {
            stage_store(ndst, 0, sv);
            for (int base = SB * 256; base < nb3; base += SB * 256) { float v[SB]; stage_load(nsrc, base, v); stage_store(ndst, base, v); }   // W > 680
        }
#pragma unroll 1
        for (int s0 = 0; s0 < ksteps; s0 += CR_D)
#pragma unroll
        for (int u = 0; u < CR_D; ++u) {
            const int s = s0 + u;
            if (s < ksteps) {
                fa[(u + CR_D - 1) % CR_D] = a_load(s + CR_D - 1);
                int px = 4 * (sm * s + sw) + g; if (px > W - 1) px = W - 1; // a pixel past the row multiplies zeros: any finite B value
                const float b0 = bl[px * 3 + boff0];
                const float b1 = boff1 >= 0 ? bl[px * 3 + boff1] : 0.f;
                const f32x4 av = __builtin_bit_cast(f32x4, fa[u]);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], b0, acc[t][0], 0, 0, 0);
                    acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], b1, acc[t][1], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    // D tile (t, h): row m = 4 g + jj is channel c0 + 4 m + t, column i is n = 16 h + i
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int n = 16 * h + i, c = c0 + 4 * (4 * g + jj) + t;
                if (n < 27) part[((size_t)(CW == 4 ? sp : sp * 4 + wave) * C + c) * 27 + n] = acc[t][h][jj];
            }
}

// second stage (fixed-order row reduce over the split partials, reduce_rows.h) and the store into the parameter's layout
__global__ __launch_bounds__(1024) void corr_rgb_final_kernel(const float* __restrict__ part, int nsplit, float* __restrict__ dw, int C,
                                                              int mode, float alpha, int accumulate) {
    __shared__ double red[16][64];
    const int e = blockIdx.x * 64 + (threadIdx.x & 63);  // e = c*27 + oy*9 + ox*3 + k
    const double s = reduce_rows_block(part, nsplit, C * 27, e, e < C * 27, red);
    if (e >= C * 27 || (threadIdx.x >> 6) != 0) return;
    const int c = e / 27, rem = e - c * 27;
    const int oy = rem / 9, ox = (rem - oy * 9) / 3, k = rem % 3;
    const float v = alpha * (float)s;
    float* d = mode == 0 ? dw + ((c * 3 + k) * 3 + oy) * 3 + ox                        // [C][3][3][3]
                         : dw + (((size_t)k * C + c) * 3 + (2 - oy)) * 3 + (2 - ox);   // [3][C][3][3]
    *d = accumulate ? *d + v : v;
}

// column sums of a 3-channel tensor [P][3] -> db[3]
__global__ __launch_bounds__(256) void colsum3_kernel(const float* __restrict__ t, float* __restrict__ part, long P) {
    float s[3] = {0.f, 0.f, 0.f};
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (long)gridDim.x * blockDim.x) {
        s[0] += t[p * 3]; s[1] += t[p * 3 + 1]; s[2] += t[p * 3 + 2];
    }
    __shared__ float red[4][3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float w = wave_sum(s[k]);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = w;
    }
    __syncthreads();
    if (threadIdx.x < 3) part[blockIdx.x * 3 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
__global__ void colsum3_final_kernel(const double* __restrict__ dsum, float* __restrict__ db, float alpha) {
    if (threadIdx.x < 3) db[threadIdx.x] = alpha * (float)dsum[threadIdx.x];
}

namespace {
struct RgbPlan { int nsplit, rows_per_split, mfma, nparts; size_t pad_bytes, part_bytes, dsum_bytes, bias_bytes, total; };
static bool rgb_plan(int N, int H, int W, int C, RgbPlan* p) {
    if (C < 1) return false;
    const int NH = N * H;
    // MFMA kernel (C % 256 == 0): one 4-wave workgroup per 256 channels and row range, two resident per CU -> ~512 workgroups
    // mfma = 4: four channel groups per workgroup; 1: one (its four waves split the pixels and write a partial each); 0: VALU kernel
    const bool fits = (size_t)W * C * 4 < ((size_t)1 << 31) && (size_t)2 * 3 * (W + 2) * 3 * sizeof(float) <= 64 * 1024;
    p->mfma = !fits ? 0 : (C % 256 == 0 ? 4 : (C % 64 == 0 ? 1 : 0));
    int want = p->mfma ? 512 / (C / (64 * p->mfma)) : 4096 / ((C + 63) / 64);
    if (want > NH) want = NH;
    if (want < 1) want = 1;
    p->rows_per_split = (NH + want - 1) / want;
    p->nsplit = (NH + p->rows_per_split - 1) / p->rows_per_split;
    p->nparts = p->mfma == 1 ? 4 * p->nsplit : p->nsplit;
    p->pad_bytes = ((size_t)N * (H + 2) * (W + 2) * 3 * sizeof(float) + 255) / 256 * 256;
    p->part_bytes = ((size_t)p->nparts * C * 27 * sizeof(float) + 255) / 256 * 256;
    p->dsum_bytes = ((size_t)C * 27 * sizeof(double) + 255) / 256 * 256;
    p->bias_bytes = (size_t)1024 * 2 * (C > 4 ? C : 4) * sizeof(float) + 2 * (size_t)(C > 4 ? C : 4) * sizeof(double) + 256;
    p->total = p->pad_bytes + p->part_bytes + p->dsum_bytes + p->bias_bytes;
    return true;
}
}  // namespace

size_t pesr_conv3x3_wgrad_rgb_ws_bytes(int N, int H, int W, int C) {
    RgbPlan p;
    return rgb_plan(N, H, W, C, &p) ? p.total : 0;
}

int pesr_conv3x3_wgrad_rgb_launch(const float* A, const float* b3, float* dw, float* db, int N, int H, int W, int C, int mode,
                                  float alpha, int accumulate, void* ws, size_t ws_bytes, hipStream_t stream) {
    RgbPlan p;
    if (!rgb_plan(N, H, W, C, &p)) return PESR_EINVAL;
    if (accumulate && db) return PESR_EINVAL;
    if (!ws || ws_bytes < p.total) return PESR_EWORKSPACE;
    float* b3p = (float*)ws;
    float* part = (float*)((char*)ws + p.pad_bytes);
    double* dsum = (double*)((char*)ws + p.pad_bytes + p.part_bytes);
    float* bpart = (float*)((char*)ws + p.pad_bytes + p.part_bytes + p.dsum_bytes);
    const long padn = (long)N * (H + 2) * (W + 2) * 3;
    hipLaunchKernelGGL(pad_rgb_kernel, dim3((unsigned)((padn + 255) / 256 < 2048 ? (padn + 255) / 256 : 2048)), dim3(256), 0, stream, b3, b3p, N, H, W);
    if (p.mfma) {
        const size_t lds = (size_t)2 * 3 * (W + 2) * 3 * sizeof(float);
        if (p.mfma == 4)
            hipLaunchKernelGGL(corr_rgb_mfma_kernel<4>, dim3((C / 256) * p.nsplit), dim3(256), lds, stream, A, (const float*)b3p, part, N * H, H, W,
                               C, p.rows_per_split);
        else
            hipLaunchKernelGGL(corr_rgb_mfma_kernel<1>, dim3((C / 64) * p.nsplit), dim3(256), lds, stream, A, (const float*)b3p, part, N * H, H, W,
                               C, p.rows_per_split);
    } else {
        hipLaunchKernelGGL(corr_rgb_kernel, dim3(((C + 63) / 64) * p.nsplit), dim3(64), 0, stream, A, (const float*)b3p, part, N * H, H, W, C, p.rows_per_split);
    }
    hipLaunchKernelGGL(corr_rgb_final_kernel, dim3((C * 27 + 63) / 64), dim3(1024), 0, stream, (const float*)part, p.nparts, dw, C, mode, alpha,
                       accumulate);
    int rc = pesr_launch_status();
    if (rc || !db) return rc;
    const long P = (long)N * H * W;
    if (mode == 0) return pesr_bias_grad_launch(A, db, P, C, W, alpha, 0, bpart, p.bias_bytes, stream);  // needs C % 4 == 0
    const int nb = 512;
    double* bsum = (double*)bpart;
    float* bp = bpart + 8;   // 32 B of sums, then the partials
    hipLaunchKernelGGL(colsum3_kernel, dim3(nb), dim3(256), 0, stream, b3, bp, P);
    rc = pesr_reduce_rows_launch(bp, bsum, nb, 3, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(colsum3_final_kernel, dim3(1), dim3(64), 0, stream, (const double*)bsum, db, alpha);
    return pesr_launch_status();
}
