// Weight gradient of the stride-1 3x3 conv with the transposed 1-D Winograd F(2,3) along x, fp32-input MFMA, gfx950.
//
// Same contract as conv3x3_wgrad.hip (ATen convolution_backward's grad_weight for the reference `Conv`,
// model/basic.py:4-7) for even widths >= 48 and channel counts that are multiples of 64, with 2/3 of the multiplies.
// It is the adjoint of conv3x3_wino.hip: with V = [d0-d2, d1+d2, d2-d1, d1-d3] of the input columns of an x-tile (pixel
// pair) and dM = [g0, g0+g1, g0-g1, -g1] of the pair's two output gradients,
//     dU_xi[ky][co][ci] = sum over rows, x-tiles of dM_xi[row][t][co] * V_xi[row + ky - 1][t][ci]          (12 products)
//     dw[..][ky][0] = dU0 + (dU1+dU2)/2,   dw[..][ky][1] = (dU1-dU2)/2,   dw[..][ky][2] = (dU1+dU2)/2 + dU3
// i.e. 12 MFMA accumulator sets over K = x-tiles instead of 9 taps over K = pixels.
//
// One workgroup owns a 64(co) x 64(ci) x 12 block of dU in registers (8 waves x 2 x 12 accumulator tiles = 96 VGPRs) and
// sweeps a range of segments (one output row x 24 x-tiles).  Operands are staged global -> registers -> LDS with the
// transforms applied on the way (no transform pass): V rows live in a 4-slot ring - a segment needs rows r-1, r, r+1 and
// the next one only adds row r+2 - and dM is double buffered; one barrier per segment.  The LDS image interleaves the four
// x-tiles of a k-step at 16-float granularity, so the ds_read_b32 fragments are bank-conflict free without padding.
// Split-K partial blocks go to a workspace slab; wgrad_wino_reduce_kernel sums them in a fixed order, applies the output
// transform, alpha and the PixelShuffle channel un-permutation, and writes OIHW.  The bias gradient is accumulated on the
// VALU from the dM fragments (dy0 + dy1 = dM0 - dM3).
#include <mutex>
#include "common.h"
#include "launchers.h"

struct WgWinoArgs {
    const float* x;    // [N][H][W][Cin]
    const float* dy;   // [N][H][W][Cout]   (or shuffled [N][2H][2W][Cout/4] when ps_in)
    float* slab;       // [split][12][Cout][Cin]
    int N, H, W, Cin, Cout;
    int segs_x;        // 24-x-tile strips per image row
    int total_segs;    // N * segs_x * H, ordered (image, strip, row): consecutive segments walk down a strip
    int segs_per_split;
    int co_tiles, ci_tiles;
    int ps_in;
    float* bias_part;  // [split][Cout] partial column sums of dy, or null
};

constexpr int WW_NT = 512, WW_TXT = 24, WW_PLANE = (WW_TXT / 4) * 256, WW_SLOT = 4 * WW_PLANE;   // floats

__global__ __launch_bounds__(WW_NT) void conv3x3_wgrad_wino_kernel(const WgWinoArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const vring = lds;                         // [4 slots][4 xi][6 blocks][256]
    float* const dmbuf = lds + 4 * WW_SLOT;           // [2][4 xi][6 blocks][256]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int ci_tile = wave & 3, co_half = wave >> 2;

    int bid = blockIdx.x;
    const int cit = bid % a.ci_tiles;  bid /= a.ci_tiles;
    const int cot = bid % a.co_tiles;
    const int sp = bid / a.co_tiles;
    const int ci0 = cit * 64, co0 = cot * 64;

    const int seg_begin = sp * a.segs_per_split;
    int seg_end = seg_begin + a.segs_per_split;
    if (seg_end > a.total_segs) seg_end = a.total_segs;

    f32x4 acc[12][2];
#pragma unroll
    for (int t = 0; t < 12; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[t][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum[2] = {0.f, 0.f};

    // ---- staging roles: threads 0..383 own one (x-tile, 4-channel group) item of a V row, threads 128..511 one of dM ----
    const bool v_thr = tid < 384, d_thr = tid >= 128;
    const int vt = (v_thr ? tid : 0) >> 4, vc4 = tid & 15;
    const int dt = (d_thr ? tid - 128 : 0) >> 4, dc4 = tid & 15;
    // position of an item inside a plane: block of 4 x-tiles, 16-float groups interleaved over the 4 x-tiles
    const int v_pos = (vt >> 2) * 256 + (((vc4 >> 2) * 4 + (vt & 3)) * 16) + (vc4 & 3) * 4;
    const int d_pos = (dt >> 2) * 256 + (((dc4 >> 2) * 4 + (dt & 3)) * 16) + (dc4 & 3) * 4;
    const bool v_ch_ok = ci0 + vc4 * 4 < a.Cin, d_ch_ok = co0 + dc4 * 4 < a.Cout;
    const int d_C = a.ps_in ? (a.Cout >> 2) : a.Cout;
    int d_choff = 0;                                  // channel part of a dy address (floats)
    {
        const int pch = co0 + dc4 * 4;
        if (a.ps_in) { const int sub = pch / d_C, cc = pch - sub * d_C; d_choff = ((sub >> 1) * (2 * a.W) + (sub & 1)) * d_C + cc; }
        else d_choff = pch;
    }
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // Loads put RAW values into registers (an out-of-range element reads a dummy in-range address); zeroing, transform and
    // ds_write happen in store_*() behind the MFMA block.  (With the select next to the load, hipcc waits for the load
    // right there - vmcnt(0) in front of the MFMAs - and every segment pays the full global-memory latency.)
    f32x4 vx[4], dd[2];
    unsigned vmask = 0, dmask = 0;
    // per strip, a thread's column offsets (floats from the start of an image row) and their validity are fixed; a segment
    // only moves the wave-uniform row pointers
    int v_off[4], d_off[2];
    unsigned v_cols = 0, d_cols = 0;
    auto set_strip = [&](int xs) {
        v_cols = 0; d_cols = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ix = xs * 48 + 2 * vt - 1 + j;
            const bool ok = v_thr && v_ch_ok && ix >= 0 && ix < a.W;
            v_off[j] = ok ? ix * a.Cin + ci0 + vc4 * 4 : 0;
            v_cols |= ok ? (1u << j) : 0u;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ox = xs * 48 + 2 * dt + j;
            const bool ok = d_thr && d_ch_ok && ox < a.W;
            d_off[j] = ok ? (a.ps_in ? 2 * ox * d_C : ox * a.Cout) + d_choff : 0;
            d_cols |= ok ? (1u << j) : 0u;
        }
    };
    auto load_vrow = [&](int img, int xs, int iy) {   // the four input columns of this thread's x-tile, row iy
        const bool row_ok = iy >= 0 && iy < a.H;       // wave-uniform
        const float* const rowp = a.x + ((size_t)img * a.H + (row_ok ? iy : 0)) * a.W * a.Cin;
        vmask = row_ok ? v_cols : 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) vx[j] = *(const f32x4*)(rowp + v_off[j]);
    };
    auto store_vrow = [&](int slot) {
        if (v_thr) {
            const f32x4 d0 = (vmask & 1u) ? vx[0] : zero4, d1 = (vmask & 2u) ? vx[1] : zero4, d2 = (vmask & 4u) ? vx[2] : zero4,
                        d3 = (vmask & 8u) ? vx[3] : zero4;
            float* p = vring + slot * WW_SLOT + v_pos;
            *(f32x4*)(p) = d0 - d2;
            *(f32x4*)(p + WW_PLANE) = d1 + d2;
            *(f32x4*)(p + 2 * WW_PLANE) = d2 - d1;
            *(f32x4*)(p + 3 * WW_PLANE) = d1 - d3;
        }
    };
    auto load_dm = [&](int img, int xs, int oy) {     // the two output-gradient pixels of this thread's x-tile, row oy
        const float* const rowp = a.ps_in ? a.dy + ((size_t)img * (2 * a.H) + 2 * oy) * (2 * a.W) * d_C
                                          : a.dy + ((size_t)img * a.H + oy) * a.W * a.Cout;
        dmask = d_cols;
#pragma unroll
        for (int j = 0; j < 2; ++j) dd[j] = *(const f32x4*)(rowp + d_off[j]);
    };
    auto store_dm = [&](int buf) {
        if (d_thr) {
            const f32x4 g0 = (dmask & 1u) ? dd[0] : zero4, g1 = (dmask & 2u) ? dd[1] : zero4;
            float* p = dmbuf + buf * WW_SLOT + d_pos;
            *(f32x4*)(p) = g0;
            *(f32x4*)(p + WW_PLANE) = g0 + g1;
            *(f32x4*)(p + 2 * WW_PLANE) = g0 - g1;
            *(f32x4*)(p + 3 * WW_PLANE) = zero4 - g1;
        }
    };
    auto seg_coords = [&](int seg, int& img, int& xs, int& row) {
        const int strip = seg / a.H;
        row = seg - strip * a.H;
        img = strip / a.segs_x;
        xs = strip - img * a.segs_x;
    };

    // ---- fragment addresses (floats): lane (r, g) reads x-tile 4k + g, channel 16*tile + r ------------------------------
    const int b_lane = (ci_tile * 4 + g) * 16 + r;
    const int a_lane = (co_half * 2 * 4 + g) * 16 + r;     // second m-tile: + 64

    if (seg_begin >= seg_end) return;                       // (never: the planner hands every workgroup at least one segment)
    int img, xs, row;
    seg_coords(seg_begin, img, xs, row);
    set_strip(xs);
    // prologue: rows row-1, row, row+1 -> slots 0, 1, 2; dM(row) -> buffer 0
#pragma unroll 1
    for (int k = 0; k < 3; ++k) { load_vrow(img, xs, row - 1 + k); store_vrow(k); }
    load_dm(img, xs, row); store_dm(0);
    __syncthreads();
    int base = 0;                                           // ring slot of the segment's top halo row

#pragma unroll 1
    for (int seg = seg_begin; seg < seg_end; ++seg) {
        const int par = (seg - seg_begin) & 1;
        const bool more = seg + 1 < seg_end;
        const bool cont = more && row + 1 < a.H;            // the next segment is the next row of the same strip
        if (cont) { load_vrow(img, xs, row + 2); load_dm(img, xs, row + 1); }

        const float* const db = dmbuf + par * WW_SLOT + a_lane;
        const float* const vb0 = vring + ((base + 0) & 3) * WW_SLOT + b_lane;
        const float* const vb1 = vring + ((base + 1) & 3) * WW_SLOT + b_lane;
        const float* const vb2 = vring + ((base + 2) & 3) * WW_SLOT + b_lane;
        float av0[8], bv0[12], av1[8], bv1[12];
#define WW_READ(AV, BV, K4)                                                                              \
        {                                                                                                \
            _Pragma("unroll") for (int xi = 0; xi < 4; ++xi) {                                           \
                AV[xi * 2 + 0] = db[xi * WW_PLANE + (K4) * 256];                                         \
                AV[xi * 2 + 1] = db[xi * WW_PLANE + (K4) * 256 + 64];                                    \
                BV[0 + xi] = vb0[xi * WW_PLANE + (K4) * 256];                                            \
                BV[4 + xi] = vb1[xi * WW_PLANE + (K4) * 256];                                            \
                BV[8 + xi] = vb2[xi * WW_PLANE + (K4) * 256];                                            \
            }                                                                                            \
        }
#define WW_MFMA(AV, BV)                                                                                  \
        _Pragma("unroll") for (int ky = 0; ky < 3; ++ky)                                                 \
            _Pragma("unroll") for (int xi = 0; xi < 4; ++xi)                                             \
                _Pragma("unroll") for (int i = 0; i < 2; ++i)                                            \
                    acc[ky * 4 + xi][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV[xi * 2 + i], BV[ky * 4 + xi], acc[ky * 4 + xi][i], 0, 0, 0); \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) bsum[i] += AV[i] - AV[6 + i];
        WW_READ(av0, bv0, 0)
#pragma unroll
        for (int k4 = 0; k4 < WW_TXT / 4; k4 += 2) {
            WW_READ(av1, bv1, k4 + 1)
            WW_MFMA(av0, bv0)
            if (k4 + 2 < WW_TXT / 4) WW_READ(av0, bv0, k4 + 2)
            if (k4 == 4) {
                // The staging stores go to LDS that nobody reads in this segment (the free ring slot, the other dM buffer),
                // so they need not wait for the end of the MFMA block: placed here - two thirds in, the loads have landed -
                // their ds_writes run under the last MFMAs, and only the barrier is left at the end of the segment.
                __builtin_amdgcn_sched_barrier(0);
                if (cont) { store_vrow((base + 3) & 3); store_dm(par ^ 1); }
                __builtin_amdgcn_sched_barrier(0);
            }
            WW_MFMA(av1, bv1)
        }
#undef WW_READ
#undef WW_MFMA
        __builtin_amdgcn_sched_barrier(0);
        if (cont) {                                         // row + 2 went to the slot of row - 1, which the next segment drops
            __syncthreads();
            base = (base + 1) & 3; ++row;
        } else if (more) {                                  // new strip / image: its three halo rows are staged from scratch
            __syncthreads();
            seg_coords(seg + 1, img, xs, row);
            set_strip(xs);
#pragma unroll 1
            for (int k = 0; k < 3; ++k) { load_vrow(img, xs, row - 1 + k); store_vrow(k); }
            load_dm(img, xs, row); store_dm(par ^ 1);
            __syncthreads();
            base = 0;
        }
    }
    __syncthreads();

    if (a.bias_part && cit == 0) {   // combine the 4 k-slot lane groups through LDS (the staging buffers are free now), fixed order
        float* red = lds;
        if (ci_tile == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i) red[g * 64 + (co_half * 2 + i) * 16 + r] = bsum[i];
        }
        __syncthreads();
        if (tid < 64 && co0 + tid < a.Cout)
            a.bias_part[(size_t)sp * a.Cout + co0 + tid] = ((red[tid] + red[64 + tid]) + red[128 + tid]) + red[192 + tid];
    }
    // slab[sp][ky*4+xi][co][ci]: D tile row = co (= (lane>>4)*4 + reg), col = ci (= lane&15)
    float* out = a.slab + (size_t)sp * 12 * a.Cout * a.Cin;
#pragma unroll
    for (int t = 0; t < 12; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int co = co0 + (co_half * 2 + i) * 16 + g * 4 + jj;
                const int ci = ci0 + ci_tile * 16 + r;
                if (co < a.Cout && ci < a.Cin) out[((size_t)t * a.Cout + co) * a.Cin + ci] = acc[t][i][jj];
            }
}

// dw[o][i][ky][kx] from the summed dU[ky*4+xi][p][i] (p = packed channel of o when ps); db as in wgrad_reduce_kernel.
__global__ void wgrad_wino_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int split, int Cout, int Cin,
                                         float alpha, int ps, const float* __restrict__ bias_part, int bias_rows,
                                         float* __restrict__ db) {
    const int C = Cout >> 2;
    if (bias_part) {
        const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
        if (p < Cout) {
            double s = 0.0;
            s = pesr_colsum_rows(bias_part + p, bias_rows, (size_t)Cout);
            int o = (int)p;
            if (ps) { const int sub = (int)p / C, cc = (int)p - sub * C; o = 4 * cc + sub; }
            db[o] = alpha * (float)s;
        }
    }
    // one thread = 4 consecutive ci of one (ky, co): 4 xi x split 16-byte loads, summed in slab order per xi
    const long plane4 = (long)Cout * Cin / 4;              // float4 units of one (ky, xi) plane
    const long total4 = 12 * plane4;
    const f32x4* slab4 = (const f32x4*)slab;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < 3 * plane4; e += (long)gridDim.x * blockDim.x) {
        const int ky = (int)(e / plane4);
        const long q = e - ky * plane4;                    // (co, ci4)
        f32x4 u[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        const f32x4* src = slab4 + (size_t)(ky * 4) * plane4 + q;
        int k = 0;
        for (; k + 4 <= split; k += 4) {      // 16 independent 16-byte loads in flight; each xi still sums its splits in slab order
            f32x4 v[4][4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int xi = 0; xi < 4; ++xi) v[kk][xi] = src[(size_t)(k + kk) * total4 + (size_t)xi * plane4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int xi = 0; xi < 4; ++xi) u[xi] += v[kk][xi];
        }
        for (; k < split; ++k)
#pragma unroll
            for (int xi = 0; xi < 4; ++xi) u[xi] += src[(size_t)k * total4 + (size_t)xi * plane4];
        const f32x4 h = 0.5f * (u[1] + u[2]);
        const f32x4 w0 = alpha * (u[0] + h), w1 = alpha * (0.5f * (u[1] - u[2])), w2 = alpha * (h + u[3]);
        const int ci = (int)((q * 4) % Cin);
        const int p = (int)((q * 4) / Cin);
        int o = p;
        if (ps) { const int sub = p / C, cc = p - sub * C; o = 4 * cc + sub; }
        float* d = dw + ((size_t)o * Cin + ci) * 9 + ky * 3;
        d[0] = w0.x; d[1] = w1.x; d[2] = w2.x;
        d[9] = w0.y; d[10] = w1.y; d[11] = w2.y;
        d[18] = w0.z; d[19] = w1.z; d[20] = w2.z;
        d[27] = w0.w; d[28] = w1.w; d[29] = w2.w;
    }
}

namespace {
struct WwPlan { int co_tiles, ci_tiles, segs_x, total_segs, split, segs_per_split; size_t slab_bytes, total_bytes; };

static bool ww_plan(int N, int H, int W, int Cin, int Cout, WwPlan* p) {
    if (W % 2 || W < 48 || Cin % 64 || Cout % 64 || N < 1 || H < 1) return false;
    p->co_tiles = Cout / 64; p->ci_tiles = Cin / 64;
    p->segs_x = (W / 2 + WW_TXT - 1) / WW_TXT;
    // a ragged last strip wastes MFMAs on zeros: accept up to ~1/8
    if ((long)p->segs_x * WW_TXT * 8 > (long)(W / 2) * 9) return false;
    p->total_segs = N * p->segs_x * H;
    const int tiles = p->co_tiles * p->ci_tiles;
    int split = (256 + tiles - 1) / tiles;
    if (split > p->total_segs) split = p->total_segs;
    if (split < 1) split = 1;
    p->segs_per_split = (p->total_segs + split - 1) / split;
    // whole strips per workgroup where possible: a strip change re-stages three halo rows synchronously
    if (p->segs_per_split > H) p->segs_per_split = (p->segs_per_split + H - 1) / H * H;
    p->split = (p->total_segs + p->segs_per_split - 1) / p->segs_per_split;
    if ((long)tiles * p->split < 8) return false;          // a handful of workgroups: leave it to the direct kernel
    p->slab_bytes = ((size_t)p->split * 12 * Cout * Cin * sizeof(float) + 255) / 256 * 256;
    p->total_bytes = p->slab_bytes + (size_t)Cout * sizeof(double) + (size_t)p->split * Cout * sizeof(float) + 1024;
    return true;
}
}  // namespace

size_t pesr_conv3x3_wgrad_wino_ws_bytes(int N, int H, int W, int Cin, int Cout) {
    WwPlan p;
    return ww_plan(N, H, W, Cin, Cout, &p) ? p.total_bytes : 0;
}

// returns PESR_EINVAL when the shape is not covered (the caller then uses the direct kernel)
int pesr_conv3x3_wgrad_wino_launch(const float* x, const float* dy, float* dw, float* db, int N, int H, int W, int Cin, int Cout,
                                   float alpha, int ps_in, void* ws, size_t ws_bytes, hipStream_t stream) {
    WwPlan p;
    if (!ww_plan(N, H, W, Cin, Cout, &p)) return PESR_EINVAL;
    if (!ws || ws_bytes < p.total_bytes) return PESR_EWORKSPACE;
    if (ps_in && Cout % 256) return PESR_EINVAL;
    WgWinoArgs a{};
    a.x = x; a.dy = dy; a.slab = (float*)ws;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.segs_x = p.segs_x; a.total_segs = p.total_segs; a.segs_per_split = p.segs_per_split;
    a.co_tiles = p.co_tiles; a.ci_tiles = p.ci_tiles; a.ps_in = ps_in;
    a.bias_part = db ? (float*)((char*)ws + p.slab_bytes + (((size_t)Cout * sizeof(double) + 255) / 256) * 256) : nullptr;
    constexpr size_t lds = (size_t)(6 * WW_SLOT) * sizeof(float);
    static_assert(lds <= 160 * 1024, "wgrad-wino LDS budget");
    static PesrDeviceOnce attr_once;
    attr_once([&] {
        (void)hipFuncSetAttribute((const void*)conv3x3_wgrad_wino_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    const int grid = p.split * p.co_tiles * p.ci_tiles;
    hipLaunchKernelGGL(conv3x3_wgrad_wino_kernel, dim3(grid), dim3(WW_NT), lds, stream, a);
    const long units = 3L * Cout * Cin / 4;
    const int rgrid = (int)((units + 255) / 256 < 2048 ? (units + 255) / 256 : 2048);
    hipLaunchKernelGGL(wgrad_wino_reduce_kernel, dim3(rgrid), dim3(256), 0, stream, (const float*)ws, dw, p.split, Cout, Cin, alpha, ps_in,
                       (const float*)a.bias_part, p.split, db);
    return pesr_launch_status();
}
