// FORKED FROM pesr_amd/csrc/conv3x3_wgrad_bf16.hip as of commit e354bfa (2026-10-03); drift since then: python scripts/diag/check_drift.py
// EXPERIMENT, not the product kernel (scripts/README.md): the bf16 weight gradient with its staging by LDS-DMA of the raw fp32 segment + an
// in-LDS conversion pass, transposed reads as inline asm with hand-counted lgkmcnt.  Correct (tests/test_bf16_gpu.py passes with it), 207 VGPRs, and
// 3-6 % SLOWER than the register-staged product kernel (94 vs 91 us G body, 758 vs 715 us upsample.2): the main loop already runs at ~70 % of the
// clock-limited MFMA-only rate, the staging was not its limit.  Build: scripts/build_variant.sh wbdma conv3x3_wgrad_bf16_ldsdma.hip
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
// go global fp32 -> LDS (LDS-DMA, raw, a whole segment ahead) -> v_cvt_pk_bf16_f32 -> the bf16 image (a conversion pass between
// two barriers after the segment's MFMAs).
// The 32-byte channel groups of a pixel are XOR-swizzled by the pixel index (dy: 256-byte rows, key px & 7; x: 128-byte rows,
// key (px >> 1) & 3, row pitch 64 pixels so that a row shift keeps the key): the 8 consecutive pixels a half-wave's transposed
// read touches then cover all 64 banks once, at every tap offset.
// Split-K partial blocks go to a workspace slab; the direct kernel's fixed-order reduce (conv3x3_wgrad.hip) finishes them into the
// OIHW parameter layout (alpha, PixelShuffle un-permutation, accumulate).
#include <mutex>
#include "common.h"
#include "launchers.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __attribute__((aligned(16))) const float g_wb_zero16[4] = {0.f, 0.f, 0.f, 0.f};   // what an out-of-image item copies

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

#ifndef WB_TARGET_WGS
#define WB_TARGET_WGS 256      // workgroups (tiles x split-K) aimed at: one round; 384 / 512 measured 25-35 % slower (slab traffic)
#endif
namespace {
constexpr int WB_CW = 48, WB_XP = 64;                    // segment columns; x halo row pitch in LDS (pixels: 2 staging passes of 32 column slots)
constexpr int WB_XBYTES = 4 * WB_XP * 128;               // x halo: 4 rows x 64 pixels (50 used) x 64 ci bf16
constexpr int WB_DBYTES = 2 * WB_CW * 256;               // dy: 96 pixels x 128 co bf16
constexpr int WB_BUF = WB_XBYTES + WB_DBYTES;            // 57,344 bytes: the bf16 image
constexpr int WB_RAW = (6 * 8 + 4 * 13) * 1024;          // 102,400 bytes: the raw fp32 segment (LDS-DMA target)
}  // namespace

// The transposed reads are inline asm: behind the builtin (a read of "any" LDS to hipcc) the compiler drains the LDS-DMA queue -
// s_waitcnt vmcnt(0) - in front of the first fragment read of every segment, i.e. before the MFMAs the DMA is meant to run under
// (cdna_hip_programming.md, "three .s-level traps", (a)).  hipcc does not count asm reads, so they are waited for by hand:
// WB_WAIT(n, regs...) = s_waitcnt lgkmcnt(n) that names the fragments it covers as in / out operands, which keeps every MFMA that
// uses them behind it.  LDS operations complete in order, so "at most n outstanding" covers everything older than the last n reads
// (scalar loads in flight can only make the wait stricter).
#define WB_TR(DST, ADDR, IMM) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "i"(IMM))
struct WbFrag { u32x2 lo, hi; };
__device__ __forceinline__ bf16x8 wb_pack(const WbFrag& f) {
    const u32x4 v = {f.lo.x, f.lo.y, f.hi.x, f.hi.y};
    return __builtin_bit_cast(bf16x8, v);
}

__global__ __launch_bounds__(512) void conv3x3_wgrad_bf16_kernel(const WgB16Args a) {
    constexpr int NT = 512;
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int ci_tile = wave & 3, co_half = wave >> 2;

    int bid = blockIdx.x;
    const int cit = bid % a.ci_tiles;  bid /= a.ci_tiles;
    const int cot = bid % a.co_tiles;
    const int sp = bid / a.co_tiles;
    const int ci0 = cit * 64, co0 = cot * 128;
    const int seg_begin = sp * a.segs_per_split;
    int seg_end = seg_begin + a.segs_per_split;
    if (seg_end > a.total_segs) seg_end = a.total_segs;

    // ---- transposed-read addresses: lane (g, q, p) supplies pixel (row h, column 4 g + q [+ kx]), channels 4 p .. 4 p + 3 of its tile
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;      // LDS address of the bf16 image
    unsigned x_a[3][2], d_a[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int P = h * WB_XP + 4 * g + q + kx;
            x_a[kx][h] = lds0 + P * 128 + ((ci_tile ^ ((P >> 1) & 3)) * 32) + p * 8;
        }
        const int px = h * WB_CW + 4 * g + q;
#pragma unroll
        for (int i = 0; i < 4; ++i) d_a[h][i] = lds0 + WB_XBYTES + px * 256 + (((co_half * 4 + i) ^ (px & 7)) * 32) + p * 8;
    }

    // ---- staging items: affine in the item index, so a thread keeps ONE source offset and ONE LDS offset per operand ---------------------
    // dy: thread = (pixel slot tid >> 5 of 16, channel group cg = tid & 31); item i: row i / 3, column slot + 16 (i % 3)
    // x : thread = (column slot tid >> 4 of 32, channel group tid & 15); item i: halo row i >> 1, column slot + 32 (i & 1) (< 50)
    // Out-of-image rows / columns and the column slots past the halo are not dereferenced: such an item copies 16 zero bytes.
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
    const int x_srcA = ((xslot - 1) * a.Cin + ci0 + cgx * 4) * 4;        // halo column xslot
    const int x_srcB = ((xslot + 31) * a.Cin + ci0 + cgx * 4) * 4;       // halo column xslot + 32 (exists for xslot < 18)
    const int x_dst0 = xslot * 128 + (((cgx >> 2) ^ ((xslot >> 1) & 3)) * 32) + (cgx & 3) * 8;
    const bool is_left = xslot == 0, is_right = xslot == 17;

    f32x4 acc[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[t][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};

    // ---- per segment: image base pointers, the segment's byte offset inside the image, which rows / edge columns exist ------------------
    const size_t x_img_bytes = (size_t)a.H * a.W * a.Cin * 4, d_img_bytes = (size_t)a.H * a.W * a.Cout * 4;
    const int x_rowbytes = a.W * a.Cin * 4;
    struct Seg { const char* xp; const char* dp; int oy0; bool a_ok, b_ok; };
    auto seg_ctx = [&](int seg) -> Seg {
        const int xs = seg % a.segs_x;
        const int rowid = seg / a.segs_x;
        const int oy0 = (rowid % a.row_groups) * 2, img = rowid / a.row_groups;
        const int ox0 = xs * WB_CW;
        Seg s;
        s.oy0 = oy0;
        s.xp = (const char*)a.x + (size_t)img * x_img_bytes + (long)((oy0 - 1) * a.W + ox0) * a.Cin * 4;     // halo row 0 (dereferenced only where it exists)
        s.dp = (const char*)a.dy + (size_t)img * d_img_bytes + (long)(oy0 * d_rowstep + ox0 * d_pixstep) * 4;
        s.a_ok = !(ox0 == 0 && is_left);                    // the column left / right of the image does not exist
        s.b_ok = xslot + 32 < 50 && !(ox0 + WB_CW == a.W && is_right);
        return s;
    };

    // ---- staging: the raw fp32 segment goes global -> LDS by LDS-DMA (buffer_load ... lds: no VGPRs in flight, so the whole next
    // segment - 100 KB per CU - streams in under this segment's MFMAs; with register staging Little's law wanted 52 VGPRs per
    // thread), lane-linear: item i of thread tid lands at raw[i][tid] (one KiB per wave and item).  After the MFMAs every thread
    // converts ITS OWN items (v_cvt_pk_bf16_f32) into the swizzled bf16 image.  One bf16 image + one raw buffer = 159,744 bytes.
    char* const raw_d = lds + WB_BUF;                       // dy: 6 items x 8 waves x 1 KiB
    char* const raw_x = raw_d + 6 * 8 * 1024;               // x: 4 halo rows x (8 waves of pass A + 5 waves of pass B) x 1 KiB
    const bool pass_b = wave < 5;                           // halo columns 32 .. 49 (.. 51: two unused slots) belong to threads 0 .. 319
    // (global_load_lds, not buffer_load ... lds: behind the buffer form hipcc drains the queue - s_waitcnt vmcnt(0) - in front of the
    // next LDS read, i.e. before the MFMAs the DMA is meant to run under.  So out-of-image items read 16 zero bytes instead.)
    auto lds_dma = [&](const char* src, const bool ok, char* dst_wave) {
        const char* const p = ok ? src : (const char*)g_wb_zero16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p, (__attribute__((address_space(3))) void*)dst_wave, 16, 0, 0);
    };
    auto dma_segment = [&](const Seg& s) {
#pragma unroll
        for (int i = 0; i < 6; ++i)
            lds_dma(s.dp + d_src0 + ((i / 3) * d_rowstep + (i % 3) * 16 * d_pixstep) * 4, s.oy0 + i / 3 < a.H, raw_d + (i * 8 + wave) * 1024);
#pragma unroll
        for (int rw = 0; rw < 4; ++rw) {
            const bool row_ok = (unsigned)(s.oy0 - 1 + rw) < (unsigned)a.H;
            lds_dma(s.xp + rw * x_rowbytes + x_srcA, row_ok && s.a_ok, raw_x + (rw * 13 + wave) * 1024);
            if (pass_b) lds_dma(s.xp + rw * x_rowbytes + x_srcB, row_ok && s.b_ok, raw_x + (rw * 13 + 8 + wave) * 1024);
        }
    };
    auto cvt_store = [&](char* dst, const f32x4 v) {
        unsigned lo, hi;
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lo) : "v"(v.x), "v"(v.y));
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(v.z), "v"(v.w));
        *(u32x2*)dst = (u32x2){lo, hi};
    };
    auto convert_segment = [&](const bool count_bias) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const f32x4 v = *(const f32x4*)(raw_d + (i * 8 + wave) * 1024 + lane * 16);
            if (count_bias) bsum += v;
            cvt_store(lds + d_dst0 + i * 16 * 256, v);
            if (i & 1) __builtin_amdgcn_sched_barrier(0);   // two items in flight at a time: the accumulators leave no room for more
        }
#pragma unroll
        for (int rw = 0; rw < 4; ++rw) {
            cvt_store(lds + x_dst0 + (rw * WB_XP) * 128, *(const f32x4*)(raw_x + (rw * 13 + wave) * 1024 + lane * 16));
            if (pass_b) cvt_store(lds + x_dst0 + (rw * WB_XP + 32) * 128, *(const f32x4*)(raw_x + (rw * 13 + 8 + wave) * 1024 + lane * 16));
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    if (seg_begin < seg_end) {
        dma_segment(seg_ctx(seg_begin));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        convert_segment(true);
    }
    __syncthreads();

#pragma unroll 1
    for (int seg = seg_begin; seg < seg_end; ++seg) {
        // the last segment fetches itself again (nobody converts it): no branch around the DMA issue
        const bool more = seg + 1 < seg_end;
        dma_segment(seg_ctx(more ? seg + 1 : seg));
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            WbFrag fd[4], fx[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) { WB_TR(fd[i].lo, d_a[0][i], s * 16 * 256); WB_TR(fd[i].hi, d_a[1][i], s * 16 * 256); }
            WB_TR(fx[0].lo, x_a[0][0], s * 16 * 128); WB_TR(fx[0].hi, x_a[0][1], s * 16 * 128);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                if (t + 1 < 9) {
                    const int ky = (t + 1) / 3, kx = (t + 1) % 3;
                    WB_TR(fx[(t + 1) & 1].lo, x_a[kx][0], ky * WB_XP * 128 + s * 16 * 128);
                    WB_TR(fx[(t + 1) & 1].hi, x_a[kx][1], ky * WB_XP * 128 + s * 16 * 128);
                }
                // everything but the two reads just issued has landed: this tap's x fragment (and, at t == 0, the dy fragments)
                if (t == 0)
                    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fx[0].lo), "+v"(fx[0].hi), "+v"(fd[0].lo), "+v"(fd[0].hi), "+v"(fd[1].lo), "+v"(fd[1].hi),
                                 "+v"(fd[2].lo), "+v"(fd[2].hi), "+v"(fd[3].lo), "+v"(fd[3].hi));
                else if (t < 8)
                    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fx[t & 1].lo), "+v"(fx[t & 1].hi));
                else
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fx[t & 1].lo), "+v"(fx[t & 1].hi));
                const bf16x8 xf = wb_pack(fx[t & 1]);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[t][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf, wb_pack(fd[i]), acc[t][i], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's DMA of the next segment has landed ...
        __syncthreads();                                     // ... everyone's has, and everyone is done with the bf16 image
        if (more) convert_segment(true);
        __syncthreads();                                     // the next image is complete
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
    if ((size_t)H * W * Cin * 4 >= ((size_t)1 << 30) || (size_t)H * W * Cout * 4 >= ((size_t)1 << 30)) return false;   // 32-bit offsets inside an image
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
    static std::once_flag attr_once;
    std::call_once(attr_once, [&] {
        (void)hipFuncSetAttribute((const void*)conv3x3_wgrad_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    hipLaunchKernelGGL(conv3x3_wgrad_bf16_kernel, dim3((unsigned)(p.split * p.co_tiles * p.ci_tiles)), dim3(512), WB_BUF + WB_RAW, stream, a);
    const int rc = pesr_launch_status();
    if (rc) return rc;
    return pesr_wgrad_reduce_launch((const float*)ws, dw, p.split, Cout, Cin, alpha, ps_in, db ? (const float*)a.bias_part : nullptr, p.split, db,
                                    accumulate, stream);
}
