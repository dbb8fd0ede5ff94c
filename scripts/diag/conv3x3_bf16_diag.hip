// FORKED FROM pesr_amd/csrc/conv3x3_bf16.hip as of commit e354bfa (2026-10-03); drift since then: python scripts/diag/check_drift.py
// DIAGNOSTIC copy of pesr_amd/csrc/conv3x3_bf16.hip with its timing-experiment switches (scripts/README.md): -DB16_FAKE_W / -DB16_FAKE_X (weights / halo re-read
// from one hot KiB: WRONG results), -DB16_ABL_NOREAD / _NOMFMA / _NOW (no LDS fragment reads / MFMAs / weight loads: WRONG results), -DB16_FXD=2|3, -DB16_PRIO=1,
// -DB16_STAGE_T=n.  Built only by scripts/build_variant.sh <name> conv3x3_bf16_diag.hip ... into exp/; profiles/r03_bf16_kernel_times.txt has what they measured.
// 3x3 stride-1 convolution on the bf16 MFMA (v_mfma_f32_16x16x32_bf16), gfx950: the OPTIONAL reduced-precision mode.
//
// Same contract and fused epilogue as conv3x3_wino4.hip (reference nn.Conv2d(k=3, padding=1), model/basic.py:4-7, forward and -
// with mode-1 packed weights - input gradient): y = act(alpha * (conv + bias) [masked] + skip), fused PixelShuffle store / fused
// pixel-unshuffle load.  Activations stay fp32 in HBM; BOTH operands of every product are rounded to bf16 (round to nearest
// even) on their way into the matrix pipe and the products are accumulated in fp32.  That is a different arithmetic from the
// reference's fp32 conv (relative error of a product 2^-8 instead of 2^-24): it has its own oracle (oracle/ops.py
// conv3x3_bf16: the same rounding, float64 accumulation) and its own tolerance, and is never selected by default.
//
// One workgroup = 144 output pixels (TR rows x TW columns) x BN = 128 * NTW output channels, 8 waves; wave w owns the 16 * NTW
// channels (w * NTW + j) * 16 .. for all nine 16-pixel m-tiles: 9 * NTW accumulator tiles.  Per 32-channel chunk:
//   * B operand (pixels): the tile's (TR + 2) x (TW + 2) input halo as bf16, [pixel][32 ch] = 64 B per pixel, double buffered in
//     LDS.  Global fp32 -> registers (issued at the top of the previous chunk) -> v_cvt_pk_bf16_f32 -> ds_write_b64 two thirds
//     into the previous chunk's MFMA stream; ONE barrier per chunk.  All nine taps read the same image at shifted pixel offsets
//     (a wave-uniform byte offset per tap); a pixel takes 96 bytes of LDS (64 of data, 32 of padding), the stride at which the
//     ds_read_b128 of 16 consecutive pixels is bank-conflict free at EVERY start offset with plain linear addressing (lane groups
//     of MI355X_MICROARCH.md, LDS table; 64 / 80 / 112 / 144 bytes are 2-way).
//   * A operand (weights): the wave's [16 ch][32 k] bf16 slabs straight from global memory into registers (1 KiB per wave and
//     slab, nothing to share through LDS), a whole chunk (9 taps) ahead.
// The weights are the MFMA's A operand, so a lane ends up with FOUR CONSECUTIVE output channels of one pixel: 16-byte stores,
// 16-byte skip / mask loads.
#include <mutex>
#include "common.h"
#include "launchers.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

struct Bf16Args {
    const float* x;            // [N][H][W][Cin]
    const char* wp;            // packed bf16 weights [9][Cin/32][Cout][32]
    const float* bias;         // [Cout] or null
    const float* skip;         // [N][H][W][Cout] or null
    const float* mask;         // [N][H][W][Cout] or null : result zeroed where mask <= 0
    float* y;                  // [N][H][W][Cout]
    int N, H, W, Cin, Cout;
    int TR, TW;                // tile: TR rows x TW pixels (TR * TW == 144)
    int tiles_x, tiles_y, n_tiles;
    int HT, WT;                // halo rows / columns
    float alpha, slope;
    int act;
    int ps;                    // 1: output stored pixel-shuffled (r = 2): packed channel (2*si+sj)*C + c -> y[n][2oy+si][2ox+sj][c], C = Cout/4
    int ps_in;                 // 1: x is a pixel-shuffled tensor [N][2H][2W][Cin/4] read as its sub-pixel-major [N][H][W][Cin] view
};

#ifndef B16_STAGE_T
#define B16_STAGE_T 6
#endif
#ifndef B16_FXD
#define B16_FXD 3
#endif
#ifndef B16_PRIO
#define B16_PRIO 0
#endif
constexpr int B16_MG = 9;      // m-tiles of 16 pixels per workgroup
constexpr int B16_PX = 96;     // LDS bytes per halo pixel: 64 of data + 32 of padding

// WTC: the halo width TW + 2 as a compile-time constant (the tile shapes of the networks' layers), or 0 = a.WT at run time.  With it
// a tap's LDS offset is an immediate of the ds_read, and the fragment reads need no address arithmetic inside the loop: with two
// waves per SIMD on 16-cycle MFMAs every other vector instruction competes with the matrix pipe for issue slots.
template <int NTW, int WTC>
__global__ __launch_bounds__(512) void conv3x3_bf16_kernel(const Bf16Args a) {
    constexpr int NT = 512, NU = 4, BN = 128 * NTW;
    const int WT = WTC ? WTC : a.WT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int v_bytes = (a.HT * WT + 1) * B16_PX;                       // + the dump pixel

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;

    // blockIdx -> (pixel tile, n-tile).  Workgroups b and b + 8 share an XCD: give every XCD a contiguous range of logical tiles,
    // n-tile fastest, so the workgroups that read the same pixels share an L2.
    int b = blockIdx.x;
    if ((gridDim.x & 7) == 0) b = (b & 7) * (gridDim.x >> 3) + (b >> 3);
    int bid = b;
    const int nt = bid % a.n_tiles;  bid /= a.n_tiles;
    const int tx = bid % a.tiles_x;  bid /= a.tiles_x;
    const int ty = bid % a.tiles_y;
    const int img = bid / a.tiles_y;
    const int gy0 = ty * a.TR, gx0 = tx * a.TW;
    const int n0 = nt * BN;
    const int C32 = a.Cin >> 5;

    // ---- pixel-operand fragment offsets: lane (r, g) reads k-group g of pixel 16 i + r, shifted by the tap's column kx ---------
    int a_off[B16_MG];
#pragma unroll
    for (int i = 0; i < B16_MG; ++i) {
        const int m = i * 16 + r;
        const int trow = m / a.TW, tcol = m - trow * a.TW;
        a_off[i] = (trow * WT + tcol) * B16_PX + g * 16;
    }

    // ---- weight operand: this lane's 16 bytes of slab (tap, chunk), n-tile j.  A buffer load: the slab's offset is a SCALAR offset
    // and j a 1-KiB immediate, so a weight fetch costs the vector unit nothing but its issue slot (the packed weights are < 4 GB).
    const int slab_bytes = a.Cout * 64;
    const unsigned b_lane = (unsigned)(((n0 + wave * NTW * 16 + r) * 32 + g * 8) * 2);
    const __amdgpu_buffer_rsrc_t w_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.wp, 0, (unsigned)((size_t)9 * a.Cin * a.Cout * 2), 0x00020000);
    auto ldw = [&](int t, int j, int cc) -> bf16x8 {
#ifdef B16_FAKE_W   // timing experiment only (wrong results): every wave re-reads ONE KiB of weights - is the L2 -> CU weight stream the limit?
        const int so = 0 * (t + cc);
#else
        const int so = (t * C32 + cc) * slab_bytes;
#endif
        return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, b_lane + j * 1024, so, 0));
    };

    // ---- staging items: (halo pixel, 4-channel group q of 8) ---------------------------------------------------------------------
    const float* const x_img = a.x + (size_t)img * a.H * a.W * a.Cin;
    const int n_items = a.HT * WT * 8;
    const int Cq = a.Cin >> 2;
    unsigned st_off[NU];
    int st_dst[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int it = tid + u * NT;
        const int q = it & 7, px = it >> 3;
        const int hrow = px / WT, hcol = px - hrow * WT;
        const int iy = gy0 - 1 + hrow, ix = gx0 - 1 + hcol;
        const bool ok = it < n_items && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        const int pix = a.ps_in ? ((2 * iy) * (2 * a.W) + 2 * ix) * Cq : (iy * a.W + ix) * a.Cin;
        // out-of-image pixels are fetched beyond the buffer descriptor's range: the load returns zeros (images are < 2 GB)
        st_off[u] = ok ? (unsigned)((pix + q * 4) * 4) : 0x80000000u;
        st_dst[u] = (it < n_items ? px : a.HT * WT) * B16_PX + q * 8;     // items past the halo land in a dump pixel behind it
    }
    const __amdgpu_buffer_rsrc_t x_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)x_img, 0, (unsigned)((size_t)a.H * a.W * a.Cin * 4), 0x00020000);
    auto chunk_off = [&](int cc) -> int {                  // channel part of an input address (bytes)
        int coff = cc * 32;
        if (a.ps_in) {   // chunk = channels [32cc, 32cc+32) of sub-pixel `sub`: one pixel of the shuffled tensor
            const int sub = coff / Cq, cc0 = coff - sub * Cq;
            coff = ((sub >> 1) * (2 * a.W) + (sub & 1)) * Cq + cc0;
        }
        return coff * 4;
    };
    u32x4 sx[NU];
    auto stage_load = [&](int cc) {
        const int so = __builtin_amdgcn_readfirstlane(chunk_off(cc));
#pragma unroll
#ifdef B16_FAKE_X   // timing experiment only: the halo is fetched once (chunk 0's bytes every time, L1 / L2 hits)
        for (int u = 0; u < NU; ++u) sx[u] = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, st_off[u], so & 0, 0);
#else
        for (int u = 0; u < NU; ++u) sx[u] = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, st_off[u], so, 0);
#endif
    };
    auto stage_store = [&](char* vdst) {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            // (volatile: left to itself hipcc converts right behind the loads, i.e. waits for them at the top of the chunk)
            unsigned lo, hi;
            asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lo) : "v"(sx[u].x), "v"(sx[u].y));
            asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(sx[u].z), "v"(sx[u].w));
            *(u32x2*)(vdst + st_dst[u]) = (u32x2){lo, hi};
        }
    };

    f32x4 acc[NTW][B16_MG];
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int i = 0; i < B16_MG; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    bf16x8 fw[9][NTW];         // the wave's weight fragments, one chunk ahead
    bf16x8 fx[B16_FXD][3];     // pixel fragments: groups of three m-tiles in a ring of B16_FXD register sets

#define B16_READ_X(FX, VB, T, GRP)                                                                       \
    {                                                                                                    \
        const int to_ = (((T) / 3) * WT + (T) % 3) * B16_PX;                                             \
        _Pragma("unroll") for (int i = 0; i < 3; ++i)                                                    \
            FX[i] = *(const bf16x8*)((VB) + a_cur[(GRP) * 3 + i] + to_);                                 \
    }
#define B16_MFMA(FX, T, GRP)                                                                             \
    _Pragma("unroll") for (int i = 0; i < 3; ++i)                                                        \
        _Pragma("unroll") for (int j = 0; j < NTW; ++j)                                                  \
            acc[j][(GRP) * 3 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[T][j], FX[i], acc[j][(GRP) * 3 + i], 0, 0, 0);

    // ---- prologue: chunk 0 staged synchronously, its nine weight slabs -----------------------------------------------------------
    // (the loads are pinned in the loop's issue order: the waits the compiler counts for the loop's first uses cover both ways in)
    stage_load(0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int j = 0; j < NTW; ++j) fw[t][j] = ldw(t, j, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    stage_store(smem);
    __syncthreads();

#pragma unroll 1
    for (int c = 0; c < C32; ++c) {
        const int cur_off = (c & 1) * v_bytes;
        int a_cur[B16_MG];                                 // this chunk's image folded into the per-lane offsets: 9 adds per chunk, not 81
#pragma unroll
        for (int i = 0; i < B16_MG; ++i) a_cur[i] = a_off[i] + cur_off;
        char* const vnext = smem + ((c & 1) ^ 1) * v_bytes;
        // The last chunk "prefetches" itself again (loads and LDS stores nobody consumes): with no branch in the loop the
        // compiler counts the outstanding loads exactly instead of draining the queue (vmcnt(0)) at every use.
        const int cn = c + 1 < C32 ? c + 1 : c;
        stage_load(cn);                                    // lands while this chunk computes
        // fragment groups G = 3 t + grp (27 per chunk) go through a ring of B16_FXD register sets: the reads of group G + B16_FXD - 1
        // are issued in front of the MFMAs of group G, i.e. B16_FXD - 1 groups (6 MFMAs each) ahead of their use
#pragma unroll
        for (int G0 = 0; G0 < B16_FXD - 1; ++G0) B16_READ_X(fx[G0], smem, G0 / 3, G0 % 3)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
#pragma unroll
            for (int grp = 0; grp < 3; ++grp) {
                const int G = t * 3 + grp, Gn = G + B16_FXD - 1;
#ifndef B16_ABL_NOREAD
                if (Gn < 27) B16_READ_X(fx[Gn % B16_FXD], smem, Gn / 3, Gn % 3)
#endif
                __builtin_amdgcn_sched_barrier(0);         // keep the prefetch ahead of the MFMA group
#if B16_PRIO
                __builtin_amdgcn_s_setprio(1);
#endif
#ifndef B16_ABL_NOMFMA
                B16_MFMA(fx[G % B16_FXD], t, grp)
#else
                _Pragma("unroll") for (int i = 0; i < 3; ++i) asm volatile("" :: "v"(fx[G % B16_FXD][i]));
#endif
#if B16_PRIO
                __builtin_amdgcn_s_setprio(0);
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
            // this tap's weights are consumed: fetch the same tap of the next chunk into their registers
#ifndef B16_ABL_NOW
#pragma unroll
            for (int j = 0; j < NTW; ++j) fw[t][j] = ldw(t, j, cn);
#endif
            if (t == B16_STAGE_T) { stage_store(vnext); __builtin_amdgcn_sched_barrier(0); }
        }
        __syncthreads();                                   // the next image is complete and visible; everyone is done with this one
    }
#undef B16_READ_X
#undef B16_MFMA

    // ---- epilogue: lane (r, g) holds channels co .. co + 3 of pixel 16 i + r -----------------------------------------------------
    const size_t img_out = (size_t)img * a.H * a.W;
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        const int co = n0 + (wave * NTW + j) * 16 + g * 4;
        f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
        if (a.bias) bias4 = *(const f32x4*)(a.bias + co);
#pragma unroll
        for (int ib = 0; ib < B16_MG; ib += 3) {
            f32x4 mkv[3], skv[3];
            size_t idx[3];
            bool ok[3];
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                const int m = (ib + e) * 16 + r;
                const int trow = m / a.TW, tcol = m - trow * a.TW;
                const int oy = gy0 + trow, ox = gx0 + tcol;
                ok[e] = oy < a.H && ox < a.W;
                if (a.ps) {   // packed channel co = (2*si+sj)*C + c  ->  out[n][2*oy+si][2*ox+sj][c]
                    const int C = a.Cout >> 2;
                    const int sub = co / C, cc = co - sub * C;
                    idx[e] = (((size_t)img * (2 * a.H) + 2 * oy + (sub >> 1)) * (2 * a.W) + 2 * ox + (sub & 1)) * C + cc;
                } else {
                    idx[e] = (img_out + (size_t)oy * a.W + ox) * a.Cout + co;
                }
                if (!ok[e]) idx[e] = 0;
                if (a.mask) mkv[e] = *(const f32x4*)(a.mask + idx[e]);
                if (a.skip) skv[e] = *(const f32x4*)(a.skip + idx[e]);
            }
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                if (!ok[e]) continue;
                f32x4 o = acc[j][ib + e];
                if (a.bias) o += bias4;
                o *= a.alpha;
                if (a.mask) {
                    const f32x4 mk = mkv[e];
                    o.x = mk.x > 0.f ? o.x : 0.f; o.y = mk.y > 0.f ? o.y : 0.f; o.z = mk.z > 0.f ? o.z : 0.f; o.w = mk.w > 0.f ? o.w : 0.f;
                }
                if (a.skip) o += skv[e];
                if (a.act == PESR_ACT_RELU) {
                    o.x = o.x > 0.f ? o.x : 0.f; o.y = o.y > 0.f ? o.y : 0.f; o.z = o.z > 0.f ? o.z : 0.f; o.w = o.w > 0.f ? o.w : 0.f;
                } else if (a.act == PESR_ACT_LRELU) {
                    o.x = o.x > 0.f ? o.x : o.x * a.slope; o.y = o.y > 0.f ? o.y : o.y * a.slope;
                    o.z = o.z > 0.f ? o.z : o.z * a.slope; o.w = o.w > 0.f ? o.w : o.w * a.slope;
                }
                *(f32x4*)(a.y + idx[e]) = o;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// weight packing: OIHW fp32 -> [9][R/32][Nn][32] bf16 (round to nearest even)
//   mode 0 (forward): out[t][c][n][k] = w[o = unperm(n)][i = 32c + k][t]
//   mode 1 (dgrad)  : out[t][c][n][k] = w[o = unperm(32c + k)][i = n][8 - t]   (the input gradient is the conv with the flipped kernel)
// ps = 1: the conv feeds nn.PixelShuffle(2); its output channels are ordered sub-pixel-major like pack.hip does.
__global__ void pack_bf16_kernel(const float* __restrict__ w, __bf16* __restrict__ out, int O, int I, int mode, int ps) {
    const int R = mode == 0 ? I : O, Nn = mode == 0 ? O : I;
    const long total = 9L * R * Nn;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int k = (int)(e & 31);
        long rest = e >> 5;
        const int n = (int)(rest % Nn); rest /= Nn;
        const int c = (int)(rest % (R >> 5));
        const int t = (int)(rest / (R >> 5));
        const int red = c * 32 + k;
        int o = mode == 0 ? n : red;
        const int i = mode == 0 ? red : n;
        if (ps) { const int C = O >> 2; const int sub = o / C, cc = o - sub * C; o = 4 * cc + sub; }
        out[e] = (__bf16)w[((long)o * I + i) * 9 + (mode == 0 ? t : 8 - t)];
    }
}

int pesr_pack_conv3x3_bf16_launch(const float* w, void* out, int O, int I, int mode, int ps, hipStream_t stream) {
    if (O % 32 || I % 32 || (mode != 0 && mode != 1) || (ps && O % 128)) return PESR_EINVAL;
    const long total = 9L * O * I;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_bf16_kernel, dim3(grid), dim3(256), 0, stream, w, (__bf16*)out, O, I, mode, ps);
    return pesr_launch_status();
}

namespace {
struct B16Plan { int TR, TW, HT, WT, tiles_x, tiles_y, n_tiles, ntw; long tiles; size_t lds; int score; };

// Tile shape TR x TW == 144 pixels with the least out-of-image area whose halo fits the four staging items per thread.
static bool b16_plan(int N, int H, int W, int Cin, int Cout, B16Plan* p, int min_wgs = 128) {
    if (N < 1 || H < 1 || W < 1 || Cin % 32 || Cin < 32 || Cout % 128) return false;
    if ((size_t)H * W * Cin * 4 >= ((size_t)1 << 31)) return false;   // one image per buffer descriptor, offsets below 2^31
    long best = -1;
    for (int TW = 1; TW <= 144; ++TW) {
        if (144 % TW) continue;
        const int TR = 144 / TW, HT = TR + 2, WT = TW + 2;
        if (HT * WT * 8 > 4 * 512) continue;
        const long cover = (long)pesr_cdiv(H, TR) * TR * pesr_cdiv(W, TW) * TW;
        // least waste first; then m-tiles that stay inside one row (conflict-free fragment reads); then the smallest halo
        const long score = cover * 8192 + (TW % 16 ? 4096 : 0) + (long)HT * WT;
        if (best < 0 || score < best) { best = score; p->TR = TR; p->TW = TW; }
    }
    if (best < 0) return false;
    p->HT = p->TR + 2; p->WT = p->TW + 2;
    p->ntw = Cout % 256 == 0 ? 2 : 1;
    p->tiles_y = pesr_cdiv(H, p->TR); p->tiles_x = pesr_cdiv(W, p->TW); p->n_tiles = Cout / (128 * p->ntw);
    p->tiles = (long)N * p->tiles_y * p->tiles_x * p->n_tiles;
    p->lds = (size_t)2 * (p->HT * p->WT + 1) * B16_PX;
    const double cover_eff = (double)H * W / ((double)p->tiles_y * p->TR * p->tiles_x * p->TW);
    p->score = p->tiles >= min_wgs ? (int)(1000.0 * cover_eff) : 0;
    return true;
}
}  // namespace

// per-mille of tile area inside the image (0: unsupported shape, or fewer than min_wgs workgroups: not worth leaving the fp32 kernels)
int pesr_conv3x3_bf16_score_impl(int N, int H, int W, int Cin, int Cout, int min_wgs) {
    B16Plan p;
    if (!b16_plan(N, H, W, Cin, Cout, &p, min_wgs)) return 0;
    return p.score;
}

int pesr_conv3x3_bf16_launch(const float* x, const void* wp, const float* bias, const float* skip, const float* mask, float* y,
                             int N, int H, int W, int Cin, int Cout, float alpha, int act, float slope, int ps, int ps_in,
                             hipStream_t stream) {
    B16Plan p;
    if (!b16_plan(N, H, W, Cin, Cout, &p)) return PESR_EINVAL;
    if (ps && (Cout % (4 * 128 * p.ntw) || skip || mask)) return PESR_EINVAL;   // an n-tile must stay inside one sub-pixel plane
    if (ps_in && Cin % 128) return PESR_EINVAL;                                 // a 32-channel chunk must stay inside one sub-pixel
    Bf16Args a{};
    a.x = x; a.wp = (const char*)wp; a.bias = bias; a.skip = skip; a.mask = mask; a.y = y;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.alpha = alpha; a.slope = slope; a.act = act; a.ps = ps; a.ps_in = ps_in;
    a.TR = p.TR; a.TW = p.TW; a.HT = p.HT; a.WT = p.WT;
    a.tiles_x = p.tiles_x; a.tiles_y = p.tiles_y; a.n_tiles = p.n_tiles;
#define B16_LAUNCH(NTW_, WT_) hipLaunchKernelGGL((conv3x3_bf16_kernel<NTW_, WT_>), dim3((unsigned)p.tiles), dim3(512), p.lds, stream, a)
#define B16_BY_WT(NTW_)                                      \
    switch (p.WT) {                                          \
        case 50: B16_LAUNCH(NTW_, 50); break;                \
        case 26: B16_LAUNCH(NTW_, 26); break;                \
        case 14: B16_LAUNCH(NTW_, 14); break;                \
        default: B16_LAUNCH(NTW_, 0); break;                 \
    }
    if (p.ntw == 2) { B16_BY_WT(2) } else { B16_BY_WT(1) }
#undef B16_BY_WT
#undef B16_LAUNCH
    return pesr_launch_status();
}
