// Weight gradient of the stride-1 3x3 conv with the transposed 1-D Winograd F(4,3) along x, fp32-input MFMA, gfx950.
//
// Same contract as conv3x3_wgrad.hip (ATen convolution_backward's grad_weight for the reference `Conv`,
// model/basic.py:4-7) for widths that are multiples of 4 (>= 48) and channel counts that are multiples of 64, with HALF of the
// direct kernel's multiplies.  It is the adjoint of conv3x3_wino4.hip: with V = B^T d of the six input columns of an x-tile
// (four output pixels) and dM = A dy of the tile's four output gradients,
//     dM = [dy0, dy0+dy1+dy2+dy3, dy0-dy1+dy2-dy3, dy0+2dy1+4dy2+8dy3, dy0-2dy1+4dy2-8dy3, dy3]
//     dU_xi[ky][co][ci] = sum over rows, x-tiles of dM_xi[row][t][co] * V_xi[row + ky - 1][t][ci]          (18 products)
//     dw[..][ky][0] = dU0/4 - (dU1+dU2)/6 + (dU3+dU4)/24
//     dw[..][ky][1] =        - (dU1-dU2)/6 + (dU3-dU4)/12
//     dw[..][ky][2] =        - (dU1+dU2)/6 + (dU3+dU4)/6 + dU5
// i.e. 18 MFMA accumulator sets over K = x-tiles (pixels / 4) instead of 9 taps over K = pixels.  Measured against fp64 the
// error is ~1.5e-6 .. 2e-6 of the gradient's maximum (direct / F(2,3): 0.5 .. 1.4e-6).
//
// One workgroup owns a 64(co) x 64(ci) x 18 block of dU in registers (8 waves x 2 x 18 accumulator tiles = 144 VGPRs) and
// sweeps a range of segments (one output row x 12 x-tiles = 48 pixels).  Operands are staged global -> registers -> LDS with
// the transforms applied on the way: V rows live in a 4-slot ring - a segment needs rows r-1, r, r+1 and the next one only
// adds row r+2 - and dM is double buffered; one barrier per segment.  The LDS image interleaves the four x-tiles of a k-step
// at 16-float granularity, so the ds_read_b32 fragments are bank-conflict free without padding.  The G^T output transform
// happens in registers before the partial block leaves, so the split-K slab has the direct kernel's [split][9][Cout][Cin]
// layout and its fixed-order reduce kernel (alpha, PixelShuffle channel un-permutation, OIHW store, bias) is shared.
// The bias gradient is accumulated on the VALU from the dM_1 fragments (dy0+dy1+dy2+dy3).
#include "common.h"
#include "launchers.h"

struct Wg4Args {
    const float* x;    // [N][H][W][Cin]
    const float* dy;   // [N][H][W][Cout]   (or shuffled [N][2H][2W][Cout/4] when ps_in)
    float* slab;       // [split][9][Cout][Cin]
    int N, H, W, Cin, Cout;
    int segs_x;        // 12-x-tile strips per image row
    int total_segs;    // N * segs_x * H, ordered (image, strip, row): consecutive segments walk down a strip
    int segs_per_split;
    int co_tiles, ci_tiles;
    int ps_in;
    float* bias_part;  // [split][Cout] partial column sums of dy, or null
};

constexpr int G4_NT = 512, G4_TXT = 12, G4_PLANE = (G4_TXT / 4) * 256, G4_SLOT = 6 * G4_PLANE;   // floats

__global__ __launch_bounds__(G4_NT) void conv3x3_wgrad_wino4_kernel(const Wg4Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const vring = lds;                         // [4 slots][6 xi][3 blocks][256]
    float* const dmbuf = lds + 4 * G4_SLOT;           // [2][6 xi][3 blocks][256]

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

    f32x4 acc[18][2];
#pragma unroll
    for (int t = 0; t < 18; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[t][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum[2] = {0.f, 0.f};

    // ---- staging roles (wave-uniform): waves 0..2 own one (x-tile, 4-channel group) item of a V row each thread, waves 4..6
    //      one of dM; both kinds go through the SAME six staging registers -----------------------------------------------------
    const bool v_role = wave < 3, d_role = wave >= 4 && wave < 7;
    const int it = v_role ? tid : (d_role ? tid - 256 : 0);
    const int st = it >> 4, sc4 = it & 15;            // x-tile of the strip, 4-channel group
    // position of an item inside a plane: block of 4 x-tiles, 16-float groups interleaved over the 4 x-tiles
    const int s_pos = (st >> 2) * 256 + (((sc4 >> 2) * 4 + (st & 3)) * 16) + (sc4 & 3) * 4;
    const int d_C = a.ps_in ? (a.Cout >> 2) : a.Cout;
    int s_choff;                                      // channel part of this thread's addresses (floats)
    if (v_role) s_choff = ci0 + sc4 * 4;
    else {
        const int pch = co0 + sc4 * 4;
        if (a.ps_in) { const int sub = pch / d_C, cc = pch - sub * d_C; s_choff = ((sub >> 1) * (2 * a.W) + (sub & 1)) * d_C + cc; }
        else s_choff = pch;
    }
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // Loads put RAW values into registers (an out-of-range element reads a dummy in-range address); zeroing, transform and
    // ds_write happen in stage_store() behind most of the MFMA block.
    f32x4 sx[6];
    unsigned s_cols = 0, s_mask = 0;
    int s_off[6];
    auto set_strip = [&](int xs) {   // per strip, a thread's column offsets (floats from the start of a row) and their validity
        s_cols = 0;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            bool ok; int off;
            if (v_role) {
                const int ix = xs * 48 + 4 * st - 1 + j;
                ok = ix >= 0 && ix < a.W;
                off = ix * a.Cin;
            } else {
                const int ox = xs * 48 + 4 * st + j;
                ok = d_role && j < 4 && ox < a.W;
                off = a.ps_in ? 2 * ox * d_C : ox * a.Cout;
            }
            s_off[j] = ok ? off + s_choff : 0;
            s_cols |= ok ? (1u << j) : 0u;
        }
    };
    auto stage_load = [&](int img, int row_v, int row_d) {   // V row row_v (may be out of the image) / dM row row_d
        if (v_role) {
            const bool row_ok = row_v >= 0 && row_v < a.H;
            const float* const rowp = a.x + ((size_t)img * a.H + (row_ok ? row_v : 0)) * a.W * a.Cin;
            s_mask = row_ok ? s_cols : 0u;
#pragma unroll
            for (int j = 0; j < 6; ++j) sx[j] = *(const f32x4*)(rowp + s_off[j]);
        } else if (d_role) {
            const float* const rowp = a.ps_in ? a.dy + ((size_t)img * (2 * a.H) + 2 * row_d) * (2 * a.W) * d_C
                                              : a.dy + ((size_t)img * a.H + row_d) * a.W * a.Cout;
            s_mask = s_cols;
#pragma unroll
            for (int j = 0; j < 4; ++j) sx[j] = *(const f32x4*)(rowp + s_off[j]);
        }
    };
    auto stage_store = [&](int vslot, int dbuf) {
        if (v_role) {
            const f32x4 d0 = (s_mask & 1u) ? sx[0] : zero4, d1 = (s_mask & 2u) ? sx[1] : zero4, d2 = (s_mask & 4u) ? sx[2] : zero4,
                        d3 = (s_mask & 8u) ? sx[3] : zero4, d4 = (s_mask & 16u) ? sx[4] : zero4, d5 = (s_mask & 32u) ? sx[5] : zero4;
            const f32x4 t1 = d4 - 4.0f * d2, t2 = d3 - 4.0f * d1, t3 = d4 - d2, t4 = d3 - d1;
            float* p = vring + vslot * G4_SLOT + s_pos;
            *(f32x4*)(p) = 4.0f * d0 + (d4 - 5.0f * d2);
            *(f32x4*)(p + G4_PLANE) = t1 + t2;
            *(f32x4*)(p + 2 * G4_PLANE) = t1 - t2;
            *(f32x4*)(p + 3 * G4_PLANE) = t3 + 2.0f * t4;
            *(f32x4*)(p + 4 * G4_PLANE) = t3 - 2.0f * t4;
            *(f32x4*)(p + 5 * G4_PLANE) = 4.0f * d1 + (d5 - 5.0f * d3);
        } else if (d_role) {
            const f32x4 g0 = (s_mask & 1u) ? sx[0] : zero4, g1 = (s_mask & 2u) ? sx[1] : zero4, g2 = (s_mask & 4u) ? sx[2] : zero4,
                        g3 = (s_mask & 8u) ? sx[3] : zero4;
            const f32x4 e02 = g0 + g2, e13 = g1 + g3, f02 = g0 + 4.0f * g2, f13 = g1 + 4.0f * g3;
            float* p = dmbuf + dbuf * G4_SLOT + s_pos;
            *(f32x4*)(p) = g0;
            *(f32x4*)(p + G4_PLANE) = e02 + e13;
            *(f32x4*)(p + 2 * G4_PLANE) = e02 - e13;
            *(f32x4*)(p + 3 * G4_PLANE) = f02 + 2.0f * f13;
            *(f32x4*)(p + 4 * G4_PLANE) = f02 - 2.0f * f13;
            *(f32x4*)(p + 5 * G4_PLANE) = g3;
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
    for (int k = 0; k < 3; ++k) { stage_load(img, row - 1 + k, row); stage_store(k, 0); }
    __syncthreads();
    int base = 0;                                           // ring slot of the segment's top halo row

#pragma unroll 1
    for (int seg = seg_begin; seg < seg_end; ++seg) {
        const int par = (seg - seg_begin) & 1;
        const bool more = seg + 1 < seg_end;
        const bool cont = more && row + 1 < a.H;            // the next segment is the next row of the same strip
        if (cont) stage_load(img, row + 2, row + 1);

        const float* const db = dmbuf + par * G4_SLOT + a_lane;
        const float* const vb0 = vring + ((base + 0) & 3) * G4_SLOT + b_lane;
        const float* const vb1 = vring + ((base + 1) & 3) * G4_SLOT + b_lane;
        const float* const vb2 = vring + ((base + 2) & 3) * G4_SLOT + b_lane;
        // Fragments are refreshed IN PLACE: right after the two MFMAs of an (ky, xi) pair have been issued, the B register of
        // that xi is re-read for the next ky group (10 MFMAs ahead of its next use) and, in the last ky group of a k-step, so
        // are the pair's two A registers for the next k-step - 18 fragment VGPRs instead of 36 with explicit double buffers
        // (next to 144 accumulators that is the difference between no spills and 54).
        float av[12], bv[6];
        const float* const vbs[3] = {vb0, vb1, vb2};
#pragma unroll
        for (int xi = 0; xi < 6; ++xi) {
            av[xi * 2 + 0] = db[xi * G4_PLANE];
            av[xi * 2 + 1] = db[xi * G4_PLANE + 64];
            bv[xi] = vb0[xi * G4_PLANE];
        }
#pragma unroll
        for (int k4 = 0; k4 < G4_TXT / 4; ++k4) {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const bool last_ky = ky == 2, last_k4 = k4 + 1 == G4_TXT / 4;
                if (last_ky) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) bsum[i] += av[2 + i];          // dM_1 = dy0 + dy1 + dy2 + dy3 (before it is refreshed)
                }
#pragma unroll
                for (int xi = 0; xi < 6; ++xi) {
                    __builtin_amdgcn_sched_barrier(0);
                    acc[ky * 6 + xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[xi * 2 + 0], bv[xi], acc[ky * 6 + xi][0], 0, 0, 0);
                    acc[ky * 6 + xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[xi * 2 + 1], bv[xi], acc[ky * 6 + xi][1], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (!last_ky) bv[xi] = vbs[ky + 1][xi * G4_PLANE + k4 * 256];
                    else if (!last_k4) {
                        bv[xi] = vb0[xi * G4_PLANE + (k4 + 1) * 256];
                        av[xi * 2 + 0] = db[xi * G4_PLANE + (k4 + 1) * 256];
                        av[xi * 2 + 1] = db[xi * G4_PLANE + (k4 + 1) * 256 + 64];
                    }
                }
                if (k4 == 1 && ky == 1) {
                    // The staging stores go to LDS that nobody reads in this segment (the free ring slot, the other dM buffer):
                    // placed here, past the middle, the loads have landed and the ds_writes run under the remaining MFMAs.
                    __builtin_amdgcn_sched_barrier(0);
                    if (cont) stage_store((base + 3) & 3, par ^ 1);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (cont) {                                         // row + 2 went to the slot of row - 1, which the next segment drops
            __syncthreads();
            base = (base + 1) & 3; ++row;
        } else if (more) {                                  // new strip / image: its three halo rows are staged from scratch
            __syncthreads();
            seg_coords(seg + 1, img, xs, row);
            set_strip(xs);
#pragma unroll 1
            for (int k = 0; k < 3; ++k) { stage_load(img, row - 1 + k, row); stage_store(k, par ^ 1); }
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
    // G^T in registers, then slab[sp][ky*3+kx][co][ci]: D tile row = co (= (lane>>4)*4 + reg), col = ci (= lane&15)
    float* out = a.slab + (size_t)sp * 9 * a.Cout * a.Cin;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const f32x4 u0 = acc[ky * 6 + 0][i], u1 = acc[ky * 6 + 1][i], u2 = acc[ky * 6 + 2][i], u3 = acc[ky * 6 + 3][i],
                        u4 = acc[ky * 6 + 4][i], u5 = acc[ky * 6 + 5][i];
            const f32x4 s12 = u1 + u2, d12 = u1 - u2, s34 = u3 + u4, d34 = u3 - u4;
            const f32x4 w0 = (0.25f * u0 - (1.0f / 6.0f) * s12) + (1.0f / 24.0f) * s34;
            const f32x4 w1 = (1.0f / 12.0f) * d34 - (1.0f / 6.0f) * d12;
            const f32x4 w2 = ((1.0f / 6.0f) * s34 - (1.0f / 6.0f) * s12) + u5;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int co = co0 + (co_half * 2 + i) * 16 + g * 4 + jj;
                const int ci = ci0 + ci_tile * 16 + r;
                const size_t o = ((size_t)(ky * 3) * a.Cout + co) * a.Cin + ci;
                const size_t tap = (size_t)a.Cout * a.Cin;
                out[o] = w0[jj]; out[o + tap] = w1[jj]; out[o + 2 * tap] = w2[jj];
            }
        }
}

namespace {
struct Wg4Plan { int co_tiles, ci_tiles, segs_x, total_segs, split, segs_per_split; size_t slab_bytes, total_bytes; };

static bool wg4_plan(int N, int H, int W, int Cin, int Cout, Wg4Plan* p) {
    if (W % 4 || W < 48 || Cin % 64 || Cout % 64 || N < 1 || H < 1) return false;
    p->co_tiles = Cout / 64; p->ci_tiles = Cin / 64;
    p->segs_x = (W / 4 + G4_TXT - 1) / G4_TXT;
    // a ragged last strip wastes MFMAs on zeros: accept up to ~1/8
    if ((long)p->segs_x * G4_TXT * 8 > (long)(W / 4) * 9) return false;
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
    p->slab_bytes = ((size_t)p->split * 9 * Cout * Cin * sizeof(float) + 255) / 256 * 256;
    p->total_bytes = p->slab_bytes + (size_t)Cout * sizeof(double) + (size_t)p->split * Cout * sizeof(float) + 1024;
    return true;
}
}  // namespace

size_t pesr_conv3x3_wgrad_wino4_ws_bytes(int N, int H, int W, int Cin, int Cout) {
    Wg4Plan p;
    return wg4_plan(N, H, W, Cin, Cout, &p) ? p.total_bytes : 0;
}

// returns PESR_EINVAL when the shape is not covered (the caller then tries the F(2,3) form / the direct kernel)
int pesr_conv3x3_wgrad_wino4_launch(const float* x, const float* dy, float* dw, float* db, int N, int H, int W, int Cin, int Cout,
                                    float alpha, int ps_in, void* ws, size_t ws_bytes, hipStream_t stream) {
    Wg4Plan p;
    if (!wg4_plan(N, H, W, Cin, Cout, &p)) return PESR_EINVAL;
    if (!ws || ws_bytes < p.total_bytes) return PESR_EWORKSPACE;
    if (ps_in && Cout % 256) return PESR_EINVAL;
    Wg4Args a{};
    a.x = x; a.dy = dy; a.slab = (float*)ws;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.segs_x = p.segs_x; a.total_segs = p.total_segs; a.segs_per_split = p.segs_per_split;
    a.co_tiles = p.co_tiles; a.ci_tiles = p.ci_tiles; a.ps_in = ps_in;
    a.bias_part = db ? (float*)((char*)ws + p.slab_bytes + (((size_t)Cout * sizeof(double) + 255) / 256) * 256) : nullptr;
    constexpr size_t lds = (size_t)(6 * G4_SLOT) * sizeof(float);
    static_assert(lds <= 160 * 1024, "wgrad-wino4 LDS budget");
    static bool attr_set = false;   // benign race: idempotent
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv3x3_wgrad_wino4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const int grid = p.split * p.co_tiles * p.ci_tiles;
    hipLaunchKernelGGL(conv3x3_wgrad_wino4_kernel, dim3(grid), dim3(G4_NT), lds, stream, a);
    int rc = pesr_launch_status();
    if (rc) return rc;
    // the partial blocks already are dw in tap order: the direct kernel's fixed-order reduce finishes the job
    return pesr_wgrad_reduce_launch((const float*)ws, dw, p.split, Cout, Cin, alpha, ps_in, (const float*)a.bias_part, p.split, db, stream);
}
