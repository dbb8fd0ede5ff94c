// Weight gradient of the 3x3 conv (pad 1, stride 1|2) on the fp32-input MFMA, gfx950.
//
// Stands in for ATen convolution_backward's grad_weight for the reference `Conv`
// (reference model/basic.py:4-7), reached from loss.backward() in reference train.py:172,228,258.
//   dw[co][ci][ky][kx] = alpha * sum_{n,oy,ox} dy[n][oy][ox][co] * x[n][oy*S+ky-1][ox*S+kx-1][ci]
// GEMM view: M = Cout, N = 9*Cin, K = pixels.  The reduction runs over pixels, the slow dimension of
// both NHWC operands, so fragments are read with ds_read_b32 (16 consecutive channels per MFMA row
// group, 4 consecutive pixels as the 4 k-slots).
//
// One workgroup owns a (CO_T = 32*COW) x (CI_T = 64) x 9-tap block of dw in registers (8 waves x 36 or
// 18 accumulator tiles) and sweeps a contiguous range of output-row segments (TWO pixels each); per
// segment the 3-row input halo and the dy row segment travel global -> LDS by LDS-DMA (double buffered,
// one barrier per segment, no VGPR round trip) into a bank-interleaved image (see below).  Split-K
// partial blocks go to a workspace slab and a second kernel reduces them in a fixed order (bitwise
// reproducible, no atomics) into the OIHW parameter layout, applying alpha and the pixel-shuffle channel
// un-permutation.  The bias gradient (column sums of dy) is accumulated on the VALU from the A fragments.
// Channel counts need only be multiples of 4: tiles that overhang Cin / Cout load zeros and skip their stores.
#include <cstdlib>
#include <mutex>
#include "common.h"
#include "launchers.h"

struct WgradArgs {
    const float* x;    // [N][H][W][Cin]
    const float* dy;   // [N][OH][OW][Cout]   (or shuffled [N][2OH][2OW][Cout/4] when ps_in)
    float* slab;       // [split][9][Cout][Cin]
    int N, H, W, Cin, Cout, OH, OW;
    int segs_x;        // segments per output row group
    int row_groups;    // ceil(OH / R): a segment covers R output rows x (TWO / R) columns
    int total_segs;    // N*row_groups*segs_x
    int segs_per_split;
    int co_tiles, ci_tiles;
    int ps_in;
    float* bias_part;  // [split][Cout] partial column sums of dy (bias gradient), or null
};

__device__ __attribute__((aligned(16))) const float g_wg_zero16[4] = {0.f, 0.f, 0.f, 0.f};   // zero padding source for LDS-DMA

__device__ __forceinline__ void wg_dma16(const float* gsrc, float* lds_piece) {
    // one 1-KiB piece: lane l copies 16 B from its own gsrc to lds_piece + 16*l bytes (the LDS base is wave-uniform)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_piece, 16, 0, 0);
}

// Staging is LDS-DMA in 1-KiB pieces (one global_load_lds_dwordx4 per wave per piece), so the LDS image cannot be padded
// per pixel; instead every piece interleaves its PP pixels at 16-float granularity,
//     float offset in piece = (group * PP + slot(pixel)) * 16 + channel % 16        (group = channel / 16)
// which puts the 4 consecutive pixels that the 4 k-slots of a ds_read_b32 fragment touch on 4 different bank groups.
//   x  : piece = 4 consecutive halo pixels x 64 ci, pieces of a halo row contiguous (row padded to a multiple of 4 pixels)
//   dy : piece = 256 / CO_T pixels x CO_T co
// R: output rows per segment (narrow images: a 48-pixel segment is 2 x 24 or 4 x 12 instead of a mostly-empty 1 x 48)
template <int COW, int S, int TWO, int R>
__global__ __launch_bounds__(512) void conv3x3_wgrad_kernel(const WgradArgs a) {
    constexpr int NT = 512, NW = 8;
    constexpr int CO_T = 32 * COW;
    constexpr int CI_T = 64;
    constexpr int CW = TWO / R;                     // segment columns
    constexpr int TWX = (CW - 1) * S + 3;           // halo columns
    constexpr int HR = (R - 1) * S + 3;             // halo rows
    constexpr int XPR = (TWX + 3) / 4;              // x pieces per halo row
    constexpr int X_PIECES = HR * XPR;
    constexpr int XSTR = 256 + (S == 2 ? 32 : 0);   // floats between x pieces (stride 2: lanes 2,3 read the next piece -> other bank half)
    constexpr int PPI = 256 / CO_T;                 // dy pixels per piece: 2 (CO_T 128) or 4 (CO_T 64)
    constexpr int D_PIECES = TWO / PPI;
    constexpr int DSTR = 256 + (PPI == 2 ? 32 : 0);
    constexpr int X_FLOATS = X_PIECES * XSTR;
    constexpr int BUF_FLOATS = X_FLOATS + D_PIECES * DSTR;
    constexpr int XK = (X_PIECES + NW - 1) / NW;
    constexpr int DK = (D_PIECES + NW - 1) / NW;
    static_assert(CW % 4 == 0 && CW % PPI == 0, "a piece / a k4 step never straddles segment rows");
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int ci_tile = wave & 3, co_half = wave >> 2;

    // Workgroups b and b + 8 share an XCD: contiguous logical ranges per XCD, channel tiles fastest - the tiles of one split-K
    // slice stream the same pixels and then share an L2.
    int bid = blockIdx.x;
    if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
    const int cit = bid % a.ci_tiles;  bid /= a.ci_tiles;
    const int cot = bid % a.co_tiles;
    const int sp = bid / a.co_tiles;
    const int ci0 = cit * CI_T, co0 = cot * CO_T;

    const int seg_begin = sp * a.segs_per_split;
    int seg_end = seg_begin + a.segs_per_split;
    if (seg_end > a.total_segs) seg_end = a.total_segs;

    f32x4 acc[9][COW];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < COW; ++i) acc[t][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- DMA sources: (scalar part per piece and segment) + (one per-lane offset for all pieces) --------------------------
    // pixel slot inside a piece: identity, or (stride 2, where k-slots step by 2 pixels) the bit swap 0,2,1,3
    auto slot_of = [](int h) { return S == 1 ? h : ((h >> 1) | ((h & 1) << 1)); };
    const int x_h = slot_of((lane >> 2) & 3);       // halo pixel of this lane inside an x piece (slot_of is an involution)
    const int x_ch = ci0 + (lane >> 4) * 16 + (lane & 3) * 4;
    const int x_lane = x_ch < a.Cin ? x_h * a.Cin + x_ch : -1;                    // < 0: channel tail, always padding
    const int d_C = a.ps_in ? (a.Cout >> 2) : a.Cout;
    const int d_h = (lane >> 2) % PPI;              // dy pixel of this lane inside a piece
    const int d_ch = co0 + ((lane >> 2) / PPI) * 16 + (lane & 3) * 4;
    int d_lane = -1;
    if (d_ch < a.Cout) {
        if (a.ps_in) {   // packed channel p = sub*Cq + cc lives at shuffled pixel (2oy + sub/2, 2ox + sub%2), channel cc
            const int sub = d_ch / d_C, cc = d_ch - sub * d_C;
            d_lane = ((sub >> 1) * (2 * a.OW) + 2 * d_h + (sub & 1)) * d_C + cc;
        } else
            d_lane = d_h * a.Cout + d_ch;
    }
    const int d_rowstep = a.ps_in ? 4 * a.OW * d_C : a.OW * a.Cout;               // one output row / one output pixel of dy,
    const int d_pixstep = a.ps_in ? 2 * d_C : a.Cout;                             // in floats

    // One segment's DMA is XK + DK pieces per wave, issued as one burst at the top of the previous segment.  (Spreading
    // the pieces over the k4 steps, or staggering the two waves of a SIMD, measured the same or slower: the ~25 us the
    // staging costs per launch is memory-pipe time, not issue stalls.)  Branch-free inside a piece: an index past the
    // end re-issues the last piece (same bytes to the same place).
    struct SegCtx { const float* xseg; const float* dseg; int iy0, ix0, oy0, ox0; };
    auto seg_ctx = [&](int seg) {
        const int xs = seg % a.segs_x;
        const int rowid = seg / a.segs_x;
        const int oy0 = (rowid % a.row_groups) * R, img = rowid / a.row_groups;
        const int ox0 = xs * CW;
        SegCtx c;
        c.oy0 = oy0; c.ox0 = ox0; c.iy0 = oy0 * S - 1; c.ix0 = ox0 * S - 1;
        c.xseg = a.x + (((long)img * a.H + c.iy0) * a.W + c.ix0) * a.Cin;    // dereferenced only where valid
        c.dseg = a.dy + (long)img * a.OH * d_rowstep + (long)oy0 * d_rowstep + (long)ox0 * d_pixstep;
        return c;
    };
    auto dma_piece = [&](const SegCtx& c, int q, float* buf) {
        if (q < XK) {
            int j = wave + q * NW;
            if (X_PIECES % NW != 0 && j > X_PIECES - 1) j = X_PIECES - 1;
            const int rowj = j / XPR, col0 = (j % XPR) * 4;
            const int iy = c.iy0 + rowj, ix = c.ix0 + col0 + x_h;
            const bool ok = x_lane >= 0 && col0 + x_h < TWX && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const float* src = c.xseg + (rowj * a.W + col0) * a.Cin + x_lane;
            wg_dma16(ok ? src : g_wg_zero16, buf + j * XSTR);
        } else {
            int j = wave + (q - XK) * NW;
            if (D_PIECES % NW != 0 && j > D_PIECES - 1) j = D_PIECES - 1;
            const int py = (j * PPI) / CW, px0 = (j * PPI) % CW;
            const bool ok = d_lane >= 0 && c.oy0 + py < a.OH && c.ox0 + px0 + d_h < a.OW;
            const float* src = c.dseg + py * d_rowstep + px0 * d_pixstep + d_lane;
            wg_dma16(ok ? src : g_wg_zero16, buf + X_FLOATS + j * DSTR);
        }
    };
    constexpr int NPIECE = XK + DK;

    // ---- fragment addresses ------------------------------------------------------------------------------------------
    // A (dy): pixel 4*k4 + g, channels (co_half*COW + i)*16 + r;   B (x): halo pixel (row, hx = pxx*S + tx), channels ci_tile*16 + r
    const int a_lane = X_FLOATS + (PPI == 2 ? (g >> 1) * DSTR + (g & 1) * 16 : g * 16) + co_half * COW * PPI * 16 + r;
    int b_lane[3];
#pragma unroll
    for (int tx = 0; tx < 3; ++tx) {
        const int hx = g * S + tx;
        b_lane[tx] = (hx >> 2) * XSTR + (ci_tile * 4 + slot_of(hx & 3)) * 16 + r;
    }

    // bias gradient for free: the A fragments ARE dy, so every wave adds them up on the VALU under its MFMAs (lane (r, g)
    // covers the pixels 4*k4 + g of every segment for channel (co_half*COW + i)*16 + r); one wave per co half publishes
    float bsum[COW];
#pragma unroll
    for (int i = 0; i < COW; ++i) bsum[i] = 0.f;

    if (seg_begin < seg_end) {
        const SegCtx c0 = seg_ctx(seg_begin);
#pragma unroll
        for (int q = 0; q < NPIECE; ++q) dma_piece(c0, q, lds);
    }
    __syncthreads();

#pragma unroll 1
    for (int seg = seg_begin; seg < seg_end; ++seg) {
        const int par = (seg - seg_begin) & 1;
        const float* buf = lds + par * BUF_FLOATS;
        float* const nbuf = lds + (par ^ 1) * BUF_FLOATS;
        if (seg + 1 < seg_end) {   // lands under this segment's MFMAs
            const SegCtx cn = seg_ctx(seg + 1);
#pragma unroll
            for (int q = 0; q < NPIECE; ++q) dma_piece(cn, q, nbuf);
        }

        // fragments double-buffered across the k4 steps: the COW + 9 ds_read_b32 of step k+1 are issued before the 36
        // MFMAs of step k, so their latency hides under the matrix pipe instead of stalling in front of every MFMA group
        float av0[COW], bv0[9], av1[COW], bv1[9];
#define PESR_WG_READ(AV, BV, K4)                                                                        \
        {                                                                                              \
            const int p0_ = (K4) * 4, py_ = p0_ / CW, c0_ = (p0_ % CW) * S;                            \
            _Pragma("unroll") for (int i = 0; i < COW; ++i)                                            \
                AV[i] = buf[a_lane + (p0_ / PPI) * DSTR + i * PPI * 16];                               \
            _Pragma("unroll") for (int t = 0; t < 9; ++t)                                              \
                BV[t] = buf[b_lane[t % 3] + ((py_ * S + t / 3) * XPR + c0_ / 4) * XSTR];               \
        }
#define PESR_WG_MFMA(AV, BV)                                                                            \
        _Pragma("unroll") for (int t = 0; t < 9; ++t)                                                  \
            _Pragma("unroll") for (int i = 0; i < COW; ++i)                                            \
                acc[t][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV[i], BV[t], acc[t][i], 0, 0, 0);         \
        _Pragma("unroll") for (int i = 0; i < COW; ++i) bsum[i] += AV[i];
        PESR_WG_READ(av0, bv0, 0)
#pragma unroll
        for (int k4 = 0; k4 < TWO / 4; k4 += 2) {
            PESR_WG_READ(av1, bv1, k4 + 1)
            PESR_WG_MFMA(av0, bv0)
            if (k4 + 2 < TWO / 4) PESR_WG_READ(av0, bv0, k4 + 2)
            PESR_WG_MFMA(av1, bv1)
        }
#undef PESR_WG_READ
#undef PESR_WG_MFMA
        __syncthreads();   // retires the DMA of segment seg+1 (vmcnt) and this segment's LDS reads
    }

    if (a.bias_part && cit == 0) {   // combine the 4 k-slot lane groups through LDS (the staging buffers are free now), fixed order
        float* red = lds;
        if (ci_tile == 0) {
#pragma unroll
            for (int i = 0; i < COW; ++i) red[g * CO_T + (co_half * COW + i) * 16 + r] = bsum[i];
        }
        __syncthreads();
        if (tid < CO_T && co0 + tid < a.Cout)
            a.bias_part[(size_t)sp * a.Cout + co0 + tid] = ((red[tid] + red[CO_T + tid]) + red[2 * CO_T + tid]) + red[3 * CO_T + tid];
    }
    // slab[sp][t][co][ci]: D tile row = co (= (lane>>4)*4 + reg), col = ci (= lane&15)
    float* out = a.slab + (size_t)sp * 9 * a.Cout * a.Cin;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < COW; ++i)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int co = co0 + (co_half * COW + i) * 16 + g * 4 + jj;
                const int ci = ci0 + ci_tile * 16 + r;
                if (co < a.Cout && ci < a.Cin) out[((size_t)t * a.Cout + co) * a.Cin + ci] = acc[t][i][jj];
            }
}


// dw[o][i][t] = alpha * sum_s slab[s][t][p][i]   (p = packed channel of o when ps).
// The fused bias gradient rides along: db[o] = alpha * sum_rows bias_part[row][p] (fixed order, double) for the first
// Cout threads of the grid.
// accumulate: dw / db are added to instead of overwritten (a layer's SECOND contribution inside one backward pass).
__global__ void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int split, int Cout, int Cin,
                                    float alpha, int ps, const float* __restrict__ bias_part, int bias_rows,
                                    float* __restrict__ db, int accumulate) {
    const long total = 9L * Cout * Cin;
    const int C = Cout >> 2;
    if (bias_part) {
        const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
        if (p < Cout) {
            double s = 0.0;
            s = pesr_colsum_rows(bias_part + p, bias_rows, (size_t)Cout);
            int o = (int)p;
            if (ps) { const int sub = (int)p / C, cc = (int)p - sub * C; o = 4 * cc + sub; }
            db[o] = alpha * (float)s + (accumulate ? db[o] : 0.f);
        }
    }
    // one thread = 4 consecutive ci of one (tap, co): `split` independent 16-B loads, 8 in flight, summed in slab order
    const long total4 = total >> 2;
    const f32x4* slab4 = (const f32x4*)slab;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total4; e += (long)gridDim.x * blockDim.x) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        int k = 0;
        for (; k + 8 <= split; k += 8) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = slab4[(size_t)(k + u) * total4 + e];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; k < split; ++k) s += slab4[(size_t)k * total4 + e];
        const int ci = (int)((e * 4) % Cin);
        long rest = (e * 4) / Cin;
        const int p = (int)(rest % Cout);
        const int t = (int)(rest / Cout);
        int o = p;
        if (ps) { const int sub = p / C, cc = p - sub * C; o = 4 * cc + sub; }
        float* d = dw + ((size_t)o * Cin + ci) * 9 + t;
        if (accumulate) { d[0] += alpha * s.x; d[9] += alpha * s.y; d[18] += alpha * s.z; d[27] += alpha * s.w; }
        else { d[0] = alpha * s.x; d[9] = alpha * s.y; d[18] = alpha * s.z; d[27] = alpha * s.w; }
    }
}

// Bias gradient db[o] = alpha * sum over pixels of dy[.., o]  (ATen convolution_backward grad_bias).
// dy is viewed as a 2-D array [M][C] (C % 4 == 0).  Rows carry a class bit (row / class_div) & 1 and
// each class is summed separately: for a plain tensor class_div is huge (one class); for the gradient
// of a pixel-shuffled output the view is [N*2OH*OW][2*Cq] (two horizontally adjacent sub-pixels per
// row) with class = parity of the shuffled row, which yields the four sub-pixel sums per channel.
// Stage 1 writes per-block partials, stage 2 sums them in a fixed order (deterministic, no atomics).
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ dy, float* __restrict__ part, long M,
                                                             int C, long rows_per_block, long class_div) {
    const int C4 = C >> 2;
    const int cw = C4 < 256 ? C4 : 256;  // float4 columns handled per pass
    const int rl = 256 / cw;             // row lanes
    const int tc = threadIdx.x % cw, tr = threadIdx.x / cw;
    const long r0 = (long)blockIdx.x * rows_per_block;
    long r1 = r0 + rows_per_block; if (r1 > M) r1 = M;
    __shared__ f32x4 red[2][256];
    for (int c0 = 0; c0 < C4; c0 += cw) {
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
        if (tr < rl && c0 + tc < C4) {
            const f32x4* base = (const f32x4*)dy + c0 + tc;
            for (long rr = r0 + tr; rr < r1; rr += 4L * rl) {
                f32x4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const long q = rr + (long)u * rl;
                    v[u] = q < r1 ? base[q * C4] : (f32x4){0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const long q = rr + (long)u * rl;
                    if ((q / class_div) & 1) s1 += v[u]; else s0 += v[u];
                }
            }
        }
        red[0][threadIdx.x] = s0; red[1][threadIdx.x] = s1;
        __syncthreads();
        if (tr == 0 && c0 + tc < C4) {
            for (int k = 1; k < rl; ++k) { s0 += red[0][k * cw + tc]; s1 += red[1][k * cw + tc]; }
            f32x4* p = (f32x4*)part + (size_t)blockIdx.x * 2 * C4;
            p[c0 + tc] = s0; p[C4 + c0 + tc] = s1;
        }
        __syncthreads();
    }
}
// dsum: [2][C] column sums per row class (reduce.hip).  ps == 0: db[c] = alpha * dsum[0][c].
// ps == 1 (C = 2*Cq): db[4*cc + 2*si + sj] = alpha * dsum[si][sj*Cq + cc]
__global__ void colsum_final_kernel(const double* __restrict__ dsum, float* __restrict__ db, int C, float alpha, int ps) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = ps ? 2 * C : C;
    if (e >= total) return;
    const int si = e / C, col = e - si * C;
    int o = col;
    if (ps) { const int Cq = C >> 1; const int sj = col / Cq, cc = col - sj * Cq; o = 4 * cc + 2 * si + sj; }
    db[o] = alpha * (float)dsum[e];
}

int pesr_bias_grad_launch(const float* dy, float* db, long pixels, int Cout, int OW, float alpha, int ps_in, float* part,
                          size_t part_bytes, hipStream_t stream) {
    // pixels = N*OH*OW of the (un-shuffled) conv output; Cout its channel count
    long M; int C; long class_div;
    if (!ps_in) { M = pixels; C = Cout; class_div = (1L << 62); }
    else { M = pixels * 2; C = Cout / 2; class_div = OW; }  // [N*2OH*OW][2*Cq]
    if (C % 4) return PESR_EINVAL;
    long nb = (M + 63) / 64; if (nb > 1024) nb = 1024; if (nb < 1) nb = 1;
    long rpb = (M + nb - 1) / nb;
    nb = (M + rpb - 1) / rpb;
    // workspace: [2*C doubles of sums][nb][2][C] partials
    const size_t dsum_bytes = (size_t)2 * C * sizeof(double);
    if (part_bytes < dsum_bytes + (size_t)nb * 2 * C * sizeof(float)) return PESR_EWORKSPACE;
    double* dsum = (double*)part;
    float* pp = (float*)((char*)part + dsum_bytes);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)nb), dim3(256), 0, stream, dy, pp, M, C, rpb, class_div);
    int rc = pesr_reduce_rows_launch(pp, dsum, (int)nb, 2 * C, stream);
    if (rc) return rc;
    const int total = ps_in ? 2 * C : C;
    hipLaunchKernelGGL(colsum_final_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, (const double*)dsum, db, C, alpha, ps_in);
    return pesr_launch_status();
}

namespace {
struct WgradPlan { int cow, two, rows, row_groups, co_tiles, ci_tiles, segs_x, total_segs, split, segs_per_split; size_t slab_bytes, total_bytes; int colsum_blocks; };

static bool wgrad_plan(int N, int H, int W, int Cin, int Cout, int stride, WgradPlan* p) {
    if (Cin % 4 || Cout % 4 || (stride != 1 && stride != 2)) return false;
    const int OH = (H - 1) / stride + 1, OW = (W - 1) / stride + 1;
    p->cow = (Cout % 128 == 0) ? 4 : 2;
    p->two = stride == 1 ? 48 : 24;
    p->co_tiles = (Cout + 32 * p->cow - 1) / (32 * p->cow);
    p->ci_tiles = (Cin + 63) / 64;
    p->rows = 1;
    if (stride == 1) { if (OW <= 12) p->rows = 4; else if (OW <= 24) p->rows = 2; }
    else if (OW <= 12) p->rows = 2;
    const int cw = p->two / p->rows;
    p->segs_x = (OW + cw - 1) / cw;
    p->row_groups = (OH + p->rows - 1) / p->rows;
    p->total_segs = N * p->row_groups * p->segs_x;
    const int out_tiles = p->co_tiles * p->ci_tiles;
    int split = (256 + out_tiles - 1) / out_tiles;     // aim at >= 256 workgroups
    if (split > p->total_segs) split = p->total_segs;
    if (split < 1) split = 1;
    p->segs_per_split = (p->total_segs + split - 1) / split;
    p->split = (p->total_segs + p->segs_per_split - 1) / p->segs_per_split;
    p->slab_bytes = ((size_t)p->split * 9 * Cout * Cin * sizeof(float) + 255) / 256 * 256;
    p->colsum_blocks = 2048;
    const size_t part_bytes = (size_t)p->colsum_blocks * 2 * Cout * sizeof(float) + 2 * (size_t)Cout * sizeof(double) + 256;
    const size_t fused_bytes = (size_t)Cout * sizeof(double) + (size_t)p->split * p->ci_tiles * Cout * sizeof(float) + 1024;
    p->total_bytes = p->slab_bytes + (part_bytes > fused_bytes ? part_bytes : fused_bytes);
    return true;
}

template <int COW, int S, int TWO, int R>
static int launch_wgrad(const WgradArgs& a, int split, hipStream_t stream) {
    constexpr int TWX = (TWO / R - 1) * S + 3;
    constexpr int HR = (R - 1) * S + 3;
    constexpr int X_FLOATS = HR * ((TWX + 3) / 4) * (256 + (S == 2 ? 32 : 0));
    constexpr int PPI = 256 / (32 * COW);
    constexpr int D_FLOATS = (TWO / PPI) * (256 + (PPI == 2 ? 32 : 0));
    constexpr size_t lds = 2 * (size_t)(X_FLOATS + D_FLOATS) * sizeof(float);
    static_assert(lds <= 160 * 1024, "wgrad LDS budget");
    auto kern = conv3x3_wgrad_kernel<COW, S, TWO, R>;
    static PesrDeviceOnce attr_once;
    attr_once([&] {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    const int grid = split * a.co_tiles * a.ci_tiles;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, a);
    return pesr_launch_status();
}
}  // namespace

// The same reduction with coalesced output (round 3): one workgroup = one output channel x up to 256 input channels, NINE waves -
// wave t sums tap t's partial rows (lane = 4 consecutive ci: one KiB per slab and wave, `split` loads in slab order, 8 in flight) and
// parks its 4 sums in LDS at [ci][t]; then the 576 threads write the block's 9 * CIB floats of dw[o][ci0 ..][0 .. 8] - ONE contiguous
// run of the OIHW tensor - as 16-byte stores.  (The kernel above writes four scattered 4-byte values per thread: 8.4 us per G-body
// layer for 21 MB, 2.5 TB/s.)  Same summation order, bit-identical results.
__global__ __launch_bounds__(576) void wgrad_reduce_rows_kernel(const float* __restrict__ slab, float* __restrict__ dw, int split, int Cout,
                                                                int Cin, int CIB, float alpha, int ps, const float* __restrict__ bias_part,
                                                                int bias_rows, float* __restrict__ db, int accumulate) {
    __shared__ __attribute__((aligned(16))) float ob[256 * 9];
    const int tid = threadIdx.x, lane = tid & 63, t = tid >> 6;            // wave = tap
    const int blocks_ci = Cin / CIB;
    const int p = blockIdx.x / blocks_ci, ci0 = (blockIdx.x - p * blocks_ci) * CIB;
    const int C = Cout >> 2;
    int o = p;
    if (ps) { const int sub = p / C, cc = p - sub * C; o = 4 * cc + sub; }
    if (bias_part && ci0 == 0 && tid == 0) {
        const double sb = pesr_colsum_rows(bias_part + p, bias_rows, (size_t)Cout);
        db[o] = alpha * (float)sb + (accumulate ? db[o] : 0.f);
    }
    if (lane * 4 < CIB) {
        const size_t total4 = (size_t)9 * Cout * Cin / 4;
        const f32x4* src = (const f32x4*)slab + (((size_t)t * Cout + p) * Cin + ci0) / 4 + lane;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        int k = 0;
        for (; k + 8 <= split; k += 8) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(k + u) * total4];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; k < split; ++k) s += src[(size_t)k * total4];
        float* q = ob + (lane * 4) * 9 + t;
        q[0] = alpha * s.x; q[9] = alpha * s.y; q[18] = alpha * s.z; q[27] = alpha * s.w;
    }
    __syncthreads();
    const int n4 = CIB * 9 / 4;                                            // 16-byte pieces of the block's contiguous output run
    f32x4* dst = (f32x4*)(dw + ((size_t)o * Cin + ci0) * 9);
    for (int j = tid; j < n4; j += 576) {
        f32x4 v = ((const f32x4*)ob)[j];
        if (accumulate) v += dst[j];
        dst[j] = v;
    }
}

int pesr_wgrad_reduce_launch(const float* slab, float* dw, int split, int Cout, int Cin, float alpha, int ps, const float* bias_part,
                             int bias_rows, float* db, int accumulate, hipStream_t stream) {
    if (Cin % 4 == 0 && (Cin % 256 == 0 || Cin <= 256)) {      // coalesced-output form: blocks of CIB = min(Cin, 256) input channels
        const int CIB = Cin < 256 ? Cin : 256;
        hipLaunchKernelGGL(wgrad_reduce_rows_kernel, dim3((unsigned)(Cout * (Cin / CIB))), dim3(576), 0, stream, slab, dw, split, Cout, Cin, CIB,
                           alpha, ps, bias_part, bias_rows, db, accumulate);
        return pesr_launch_status();
    }
    const long total = 9L * Cout * Cin;
    const int rgrid = (int)((total / 4 + 255) / 256 < 2048 ? (total / 4 + 255) / 256 : 2048);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(rgrid), dim3(256), 0, stream, slab, dw, split, Cout, Cin, alpha, ps, bias_part, bias_rows, db,
                       accumulate);
    return pesr_launch_status();
}

// algo: PESR_WGRAD_AUTO (0) = the Winograd F(4,3) form where it applies, else F(2,3), else the direct kernel;
// PESR_WGRAD_DIRECT (1) = the direct kernel everywhere; PESR_WGRAD_WINO23 (2) = F(2,3) where it applies, else direct (A/B runs,
// parity cross-checks).  An explicit argument: the library keeps no hidden state.
size_t pesr_conv3x3_wgrad_ws_bytes(int N, int H, int W, int Cin, int Cout, int stride, int algo) {
    WgradPlan p;
    if (!wgrad_plan(N, H, W, Cin, Cout, stride, &p)) return 0;
    size_t need = p.total_bytes;
    if (stride == 1 && (algo == 0 || algo >= 2)) {
        const size_t ww = pesr_conv3x3_wgrad_wino_ws_bytes(N, H, W, Cin, Cout);
        if (ww > need) need = ww;
    }
    if (stride == 1 && (algo == 0 || algo >= 3)) {
        const size_t w4 = pesr_conv3x3_wgrad_wino4_ws_bytes(N, H, W, Cin, Cout);
        if (w4 > need) need = w4;
    }
    return need;
}

int pesr_conv3x3_wgrad_launch(const float* x, const float* dy, float* dw, float* db, int N, int H, int W, int Cin, int Cout,
                              int stride, float alpha, int ps_in, int algo, int accumulate, void* ws, size_t ws_bytes, hipStream_t stream) {
    WgradPlan p;
    if (algo < 0 || algo > 5) return PESR_EINVAL;
    if (!wgrad_plan(N, H, W, Cin, Cout, stride, &p)) return PESR_EINVAL;
    if (ws_bytes < p.total_bytes || !ws) return PESR_EWORKSPACE;
    if (ps_in && (stride != 1 || Cout % 16)) return PESR_EINVAL;
    if (stride == 1 && (algo == 0 || algo >= 3)) {   // Winograd F(4,3) where it applies (width % 4 == 0 and >= 48 - the 32x32x2 kernel also 24 / 16 / 12 / 8 -, 64-multiple channels)
        // auto: the 32x32x2-MFMA form with the transform nested in y (F(2,3)y x F(4,3)x, 1/3 of the direct form's multiplies), staging on
        // producer waves (round 5); PESR_WGRAD_WINO4_12W (5): round 4's 12-wave kernel of the same transform; PESR_WGRAD_WINO4_1D (4):
        // round 3's 1-D F(4,3) transform on that kernel; PESR_WGRAD_WINO4_16X16 (3): round 2's 16x16x4 form of the 1-D transform -
        // all three kept as cross-checks
        const int rc = pesr_conv3x3_wgrad_wino4_launch(x, dy, dw, db, N, H, W, Cin, Cout, alpha, ps_in, accumulate, algo == 3 ? 0 : (algo == 4 ? 1 : (algo == 5 ? 2 : 3)), ws,
                                                       ws_bytes, stream);
        if (rc != PESR_EINVAL && rc != PESR_EWORKSPACE) return rc;
    }
    // (the F(2,3) form's own reduce kernel has no accumulate mode: such a call goes to the direct kernel)
    if (stride == 1 && (algo == 0 || algo >= 2) && !accumulate) {   // Winograd F(2,3) where it applies (even width >= 48, 64-multiple channels)
        const int rc = pesr_conv3x3_wgrad_wino_launch(x, dy, dw, db, N, H, W, Cin, Cout, alpha, ps_in, ws, ws_bytes, stream);
        if (rc != PESR_EINVAL && rc != PESR_EWORKSPACE) return rc;
    }
    WgradArgs a{};
    a.x = x; a.dy = dy; a.slab = (float*)ws;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.OH = (H - 1) / stride + 1; a.OW = (W - 1) / stride + 1;
    a.segs_x = p.segs_x; a.row_groups = p.row_groups; a.total_segs = p.total_segs; a.segs_per_split = p.segs_per_split;
    a.co_tiles = p.co_tiles; a.ci_tiles = p.ci_tiles; a.ps_in = ps_in;
    // bias gradient fused into the wgrad kernel: partials [split][Cout] (+ Cout doubles) behind the slab
    const size_t bias_rows = (size_t)p.split;
    const size_t bias_need = (size_t)Cout * sizeof(double) + bias_rows * Cout * sizeof(float) + 256;
    const bool fuse_bias = db != nullptr && ws_bytes - p.slab_bytes >= bias_need;
    if (accumulate && db && !fuse_bias) return PESR_EWORKSPACE;      // the stand-alone bias-gradient kernels overwrite
    a.bias_part = fuse_bias ? (float*)((char*)ws + p.slab_bytes + (((size_t)Cout * sizeof(double) + 255) / 256) * 256) : nullptr;
    int rc;
#define PESR_WG(S_, TWO_, R_) (p.cow == 4 ? launch_wgrad<4, S_, TWO_, R_>(a, p.split, stream) : launch_wgrad<2, S_, TWO_, R_>(a, p.split, stream))
    if (stride == 1) rc = p.rows == 4 ? PESR_WG(1, 48, 4) : (p.rows == 2 ? PESR_WG(1, 48, 2) : PESR_WG(1, 48, 1));
    else rc = p.rows == 2 ? PESR_WG(2, 24, 2) : PESR_WG(2, 24, 1);
#undef PESR_WG
    if (rc) return rc;
    rc = pesr_wgrad_reduce_launch((const float*)ws, dw, p.split, Cout, Cin, alpha, ps_in,
                                  fuse_bias ? (const float*)a.bias_part : (const float*)nullptr, (int)bias_rows, db, accumulate, stream);
    if (rc || !db || fuse_bias) return rc;
    float* part = (float*)((char*)ws + p.slab_bytes);
    return pesr_bias_grad_launch(dy, db, (long)N * a.OH * a.OW, Cout, a.OW, alpha, ps_in, part, ws_bytes - p.slab_bytes, stream);
}
