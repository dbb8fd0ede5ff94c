// Weight gradient of the 3x3 stride-1 conv on the bf16 MFMA (v_mfma_f32_16x16x32_bf16), gfx950: the OPTIONAL reduced-precision
// mode, companion of conv3x3_bf16.hip.
//
// Stands in for ATen convolution_backward's grad_weight / grad_bias of the reference `Conv` (model/basic.py:4-7) the way the bf16
// mode defines it:   dw[co][ci][ky][kx] = alpha * sum_{n,y,x} bf16(dy[n][y][x][co]) * bf16(x[n][y+ky-1][x+kx-1][ci])   (fp32 sums),
// db[co] = alpha * sum dy (fp32, NOT rounded: added up on the vector unit from the staged values).
// GEMM view: rows = ci, columns = co (per tap), K = pixels - the slow dimension of both NHWC operands.  Both fragments therefore
// come out of LDS through ds_read_b64_tr_b16 (cdna_hip_programming.md T10): the images stay [pixel][channel] (channel-contiguous,
// as they arrive), a tap is a shift of the pixel index, and one transposed read hands lane i channel c0 + i of four pixels.
//
// One workgroup owns a 64 (ci) x 128 (co) x 9-tap block of dw in registers (8 waves = 4 ci tiles x 2 co halves, 36 accumulator
// tiles each) and sweeps a contiguous range of segments of 2 rows x 48 columns.  A K-step is the 32 pixels
// {row h, column 16 s + 4 g + q : h < 2, g < 4, q < 4} (k = 8 g + 4 h + q): the two transposed reads of a fragment take one row
// each, and a step / a tap only adds a wave-uniform constant to the lane's address.  Per segment the dy rows and the 4-row x halo
// go global fp32 -> registers -> v_cvt_pk_bf16_f32 -> LDS (double buffered, one barrier per segment), a third per K-step.
// The 32-byte channel groups of a pixel are XOR-swizzled by the pixel index (dy: 256-byte rows, key px & 7; x: 128-byte rows,
// key (px >> 1) & 3, row pitch 64 pixels so that a row shift keeps the key): the 8 consecutive pixels a half-wave's transposed
// read touches then cover all 64 banks once, at every tap offset.
// Split-K partial blocks go to a workspace slab; the direct kernel's fixed-order reduce (conv3x3_wgrad.hip) finishes them into the
// OIHW parameter layout (alpha, PixelShuffle un-permutation, accumulate).
#include <mutex>
#include "common.h"
#include "launchers.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

struct WgB16Args {
    const float* x;    // [N][H][W][Cin]
    const float* dy;   // [N][H][W][Cout]   (or shuffled [N][2H][2W][Cout/4] when ps_in)
    float* slab;       // [split][9][Cout][Cin]
    float* bias_part;  // [split][Cout] partial column sums of dy, or null
    int N, H, W, Cin, Cout;
    int segs_x;        // W / 48
    int row_groups;    // ceil(H / 2)
    int total_segs;    // N * row_groups * segs_x
    int segs_per_split;
    int co_tiles, ci_tiles;
    int ps_in;
};

constexpr int WB_TARGET_WGS = 256;      // workgroups (tiles x split-K) aimed at: one round; 384 / 512 measured 25-35 % slower (slab traffic)
namespace {
constexpr int WB_CW = 48, WB_XP = 64;                    // segment columns; x halo row pitch in LDS (pixels: 2 staging passes of 32 column slots)
constexpr int WB_XBYTES = 4 * WB_XP * 128;               // x halo: 4 rows x 64 pixels (50 used) x 64 ci bf16
constexpr int WB_DBYTES = 2 * WB_CW * 256;               // dy: 96 pixels x 128 co bf16
constexpr int WB_BUF = WB_XBYTES + WB_DBYTES;            // 57,344 bytes per buffer
}  // namespace

__device__ __forceinline__ bf16x8 wb_frag(const char* lo, const char* hi) {
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)lo);
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)hi);
    const s16x8 v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

__global__ __launch_bounds__(512) void conv3x3_wgrad_bf16_kernel(const WgB16Args a) {
    constexpr int NT = 512;
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int ci_tile = wave & 3, co_half = wave >> 2;

    // Workgroups b and b + 8 share an XCD (round-robin dispatch): give every XCD a contiguous range of logical blocks, tile index
    // fastest, so that the co x ci tiles of one split-K slice - which read the SAME pixels - share an L2 (without it every XCD
    // fetched every pixel: 225 MB of HBM / MALL reads per G-body launch for 75 MB of operands, profiles/r03_bf16_pmc_summary.csv).
    int bid = blockIdx.x;
    if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
    const int cit = bid % a.ci_tiles;  bid /= a.ci_tiles;
    const int cot = bid % a.co_tiles;
    const int sp = bid / a.co_tiles;
    const int ci0 = cit * 64, co0 = cot * 128;
    const int seg_begin = sp * a.segs_per_split;
    int seg_end = seg_begin + a.segs_per_split;
    if (seg_end > a.total_segs) seg_end = a.total_segs;

    // ---- transposed-read addresses: lane (g, q, p) supplies pixel (row h, column 4 g + q [+ kx]), channels 4 p .. 4 p + 3 of its tile
    int x_off[3][2], d_off[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int P = h * WB_XP + 4 * g + q + kx;
            x_off[kx][h] = P * 128 + ((ci_tile ^ ((P >> 1) & 3)) * 32) + p * 8;
        }
        const int px = h * WB_CW + 4 * g + q;
#pragma unroll
        for (int i = 0; i < 4; ++i) d_off[h][i] = WB_XBYTES + px * 256 + (((co_half * 4 + i) ^ (px & 7)) * 32) + p * 8;
    }

    // ---- staging items: affine in the item index, so a thread keeps ONE source offset and ONE LDS offset per operand ---------------------
    // dy: thread = (pixel slot tid >> 5 of 16, channel group cg = tid & 31); item i: row i / 3, column slot + 16 (i % 3)
    // x : thread = (column slot tid >> 4 of 32, channel group tid & 15); item i: halo row i >> 1, column slot + 32 (i & 1) (< 50)
    // Out-of-image rows fall outside the per-image buffer descriptor and read zeros; out-of-image COLUMNS would wrap into the
    // neighbouring row, so those items (and the column slots past the halo) get the marker offset 0xC0000000, which stays out of
    // range whatever segment offset is added (images are < 1 GB).  The range check is on the VGPR offset, so everything is in it.
    const int d_C = a.ps_in ? (a.Cout >> 2) : a.Cout;
    const int d_rowstep = a.ps_in ? 4 * a.W * d_C : a.W * a.Cout;      // one output row / one output pixel of dy, in floats
    const int d_pixstep = a.ps_in ? 2 * d_C : a.Cout;
    int d_src0, d_dst0;
    {
        const int cg = tid & 31, px = tid >> 5;
        const int ch = co0 + cg * 4;
        int ch_off = ch;
        if (a.ps_in) {   // packed channel p = sub*Cq + cc lives at shuffled pixel (2y + sub/2, 2x + sub%2), channel cc
            const int sub = ch / d_C, cc = ch - sub * d_C;
            ch_off = ((sub >> 1) * (2 * a.W) + (sub & 1)) * d_C + cc;
        }
        d_src0 = (px * d_pixstep + ch_off) * 4;
        d_dst0 = WB_XBYTES + px * 256 + (((cg >> 2) ^ (px & 7)) * 32) + (cg & 3) * 8;
    }
    const int xslot = tid >> 4, cgx = tid & 15;
    const unsigned X_OOB = 0xC0000000u;
    const unsigned x_srcA = (unsigned)(((xslot - 1) * a.Cin + ci0 + cgx * 4) * 4);                               // halo column xslot
    const unsigned x_srcB = xslot + 32 < 50 ? (unsigned)(((xslot + 31) * a.Cin + ci0 + cgx * 4) * 4) : X_OOB;   // halo column xslot + 32
    const int x_dst0 = xslot * 128 + (((cgx >> 2) ^ ((xslot >> 1) & 3)) * 32) + (cgx & 3) * 8;
    const bool is_left = xslot == 0, is_right = xslot == 17;

    f32x4 acc[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[t][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};

    // ---- per segment: image descriptors + the segment's byte offset inside the image ------------------------------------------------
    const size_t x_img_bytes = (size_t)a.H * a.W * a.Cin * 4, d_img_bytes = (size_t)a.H * a.W * a.Cout * 4;
    const int x_rowbytes = a.W * a.Cin * 4;
    struct Seg { __amdgpu_buffer_rsrc_t xr, dr; int xo, dofs; unsigned va, vb; };
    auto seg_ctx = [&](int seg) -> Seg {
        const int xs = seg % a.segs_x;
        const int rowid = seg / a.segs_x;
        const int oy0 = (rowid % a.row_groups) * 2, img = rowid / a.row_groups;
        const int ox0 = xs * WB_CW;
        Seg s;
        s.xr = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a.x + (size_t)img * x_img_bytes), 0, (unsigned)x_img_bytes, 0x00020000);
        s.dr = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a.dy + (size_t)img * d_img_bytes), 0, (unsigned)d_img_bytes, 0x00020000);
        s.xo = ((oy0 - 1) * a.W + ox0) * a.Cin * 4;         // halo row 0; rows above / below the image fall outside the descriptor
        s.dofs = (oy0 * d_rowstep + ox0 * d_pixstep) * 4;
        s.va = (ox0 == 0 && is_left) ? X_OOB : x_srcA;      // the column left / right of the image
        s.vb = (ox0 + WB_CW == a.W && is_right) ? X_OOB : x_srcB;
        return s;
    };

    // A third of a segment's staging per K-step: batch b = dy items {2b, 2b+1} and x items {3b.. } (3 + 3 + 2).
    u32x4 sv[5];
    auto batch_load = [&](const Seg& s, const int b) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = 2 * b + k;
            sv[k] = __builtin_amdgcn_raw_buffer_load_b128(s.dr, (unsigned)(d_src0 + s.dofs + ((i / 3) * d_rowstep + (i % 3) * 16 * d_pixstep) * 4), 0, 0);
        }
        const int x0 = 3 * b, nx = b == 2 ? 2 : 3;
#pragma unroll
        for (int k = 0; k < nx; ++k) {
            const int i = x0 + k;
            sv[2 + k] = __builtin_amdgcn_raw_buffer_load_b128(s.xr, ((i & 1) ? s.vb : s.va) + (unsigned)(s.xo + (i >> 1) * x_rowbytes), 0, 0);
        }
    };
    auto cvt_store = [&](char* dst, const u32x4 v) {
        unsigned lo, hi;
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lo) : "v"(v.x), "v"(v.y));
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(v.z), "v"(v.w));
        *(u32x2*)dst = (u32x2){lo, hi};
    };
    auto batch_store = [&](char* buf, const int b, const bool count_bias) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = 2 * b + k;
            if (count_bias) bsum += __builtin_bit_cast(f32x4, sv[k]);
            cvt_store(buf + d_dst0 + i * 16 * 256, sv[k]);
        }
        const int x0 = 3 * b, nx = b == 2 ? 2 : 3;
#pragma unroll
        for (int k = 0; k < nx; ++k) {
            const int i = x0 + k;
            cvt_store(buf + x_dst0 + ((i >> 1) * WB_XP + (i & 1) * 32) * 128, sv[2 + k]);
        }
    };

    if (seg_begin < seg_end) {
        const Seg s0 = seg_ctx(seg_begin);
#pragma unroll
        for (int b = 0; b < 3; ++b) { batch_load(s0, b); batch_store(lds, b, true); }
    }
    __syncthreads();

#pragma unroll 1
    for (int seg = seg_begin; seg < seg_end; ++seg) {
        const int par = (seg - seg_begin) & 1;
        const char* const buf = lds + par * WB_BUF;
        char* const nbuf = lds + (par ^ 1) * WB_BUF;
        // the last segment stages itself again into the other buffer (nobody reads it): no branch in the loop
        const bool more = seg + 1 < seg_end;
        const Seg sn = seg_ctx(more ? seg + 1 : seg);
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            batch_load(sn, s);
            bf16x8 fd[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fd[i] = wb_frag(buf + d_off[0][i] + s * 16 * 256, buf + d_off[1][i] + s * 16 * 256);
            bf16x8 fx[2];
            fx[0] = wb_frag(buf + x_off[0][0] + s * 16 * 128, buf + x_off[0][1] + s * 16 * 128);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                if (t + 1 < 9) {
                    const int ky = (t + 1) / 3, kx = (t + 1) % 3;
                    fx[(t + 1) & 1] = wb_frag(buf + x_off[kx][0] + ky * WB_XP * 128 + s * 16 * 128, buf + x_off[kx][1] + ky * WB_XP * 128 + s * 16 * 128);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[t][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fx[t & 1], fd[i], acc[t][i], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            batch_store(nbuf, s, more);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }

    // ---- bias gradient partial: the 16 threads that share a channel group meet in LDS (the staging buffers are free now) -----------
    if (a.bias_part && cit == 0) {
        f32x4* red = (f32x4*)lds;
        red[tid] = bsum;                                                 // [tid >> 5][cg]
        __syncthreads();
        if (tid < 32) {
            f32x4 s = red[tid];
#pragma unroll
            for (int k = 1; k < 16; ++k) s += red[k * 32 + tid];
            *(f32x4*)(a.bias_part + (size_t)sp * a.Cout + co0 + tid * 4) = s;
        }
    }
    // ---- slab[sp][t][co][ci]: lane (r, g) holds ci = 4 g .. 4 g + 3 (rows of the x operand) of co = r ---------------------------------
    float* out = a.slab + (size_t)sp * 9 * a.Cout * a.Cin;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int co = co0 + (co_half * 4 + i) * 16 + r;
            const int ci = ci0 + ci_tile * 16 + g * 4;
            *(f32x4*)(out + ((size_t)t * a.Cout + co) * a.Cin + ci) = acc[t][i];
        }
}

namespace {
struct WgB16Plan { int segs_x, row_groups, total_segs, co_tiles, ci_tiles, split, segs_per_split; size_t slab_bytes, total_bytes; };

static bool wgb16_plan(int N, int H, int W, int Cin, int Cout, WgB16Plan* p) {
    if (N < 1 || H < 1 || W < WB_CW || W % WB_CW || Cin % 64 || Cout % 128) return false;
    if ((size_t)H * W * Cin * 4 >= ((size_t)1 << 30) || (size_t)H * W * Cout * 4 >= ((size_t)1 << 30)) return false;   // (the marker offset)
    p->segs_x = W / WB_CW; p->row_groups = (H + 1) / 2; p->total_segs = N * p->row_groups * p->segs_x;
    p->co_tiles = Cout / 128; p->ci_tiles = Cin / 64;
    const int tiles = p->co_tiles * p->ci_tiles;
    // two rounds of 256 workgroups at most, at least 4 segments per split (each slice pays a prologue, a slab and its reduce)
    int split = (WB_TARGET_WGS + tiles - 1) / tiles;
    if (split > p->total_segs / 4) split = p->total_segs / 4;
    if (split < 1) split = 1;
    p->segs_per_split = (p->total_segs + split - 1) / split;
    p->split = (p->total_segs + p->segs_per_split - 1) / p->segs_per_split;
    p->slab_bytes = (size_t)p->split * 9 * Cout * Cin * sizeof(float);
    p->total_bytes = p->slab_bytes + (size_t)p->split * Cout * sizeof(float) + 256;
    return true;
}
}  // namespace

size_t pesr_conv3x3_wgrad_bf16_ws_bytes(int N, int H, int W, int Cin, int Cout) {
    WgB16Plan p;
    return wgb16_plan(N, H, W, Cin, Cout, &p) ? p.total_bytes : 0;
}

int pesr_conv3x3_wgrad_bf16_launch(const float* x, const float* dy, float* dw, float* db, int N, int H, int W, int Cin, int Cout,
                                   float alpha, int ps_in, int accumulate, void* ws, size_t ws_bytes, hipStream_t stream) {
    WgB16Plan p;
    if (!wgb16_plan(N, H, W, Cin, Cout, &p)) return PESR_EINVAL;
    if (!ws || ws_bytes < p.total_bytes) return PESR_EWORKSPACE;
    if (ps_in && Cout % 512) return PESR_EINVAL;            // a 128-channel co tile must stay inside one sub-pixel plane
    WgB16Args a{};
    a.x = x; a.dy = dy; a.slab = (float*)ws;
    a.bias_part = db ? (float*)((char*)ws + p.slab_bytes) : nullptr;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.segs_x = p.segs_x; a.row_groups = p.row_groups; a.total_segs = p.total_segs; a.segs_per_split = p.segs_per_split;
    a.co_tiles = p.co_tiles; a.ci_tiles = p.ci_tiles; a.ps_in = ps_in;
    static PesrDeviceOnce attr_once;
    attr_once([&] {
        (void)hipFuncSetAttribute((const void*)conv3x3_wgrad_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    hipLaunchKernelGGL(conv3x3_wgrad_bf16_kernel, dim3((unsigned)(p.split * p.co_tiles * p.ci_tiles)), dim3(512), 2 * WB_BUF, stream, a);
    const int rc = pesr_launch_status();
    if (rc) return rc;
    return pesr_wgrad_reduce_launch((const float*)ws, dw, p.split, Cout, Cin, alpha, ps_in, db ? (const float*)a.bias_part : nullptr, p.split, db,
                                    accumulate, stream);
}
