// 3x3 stride-1 convolution on the bf16 MFMA with SPLIT operands (v_mfma_f32_16x16x32_bf16 x 3), gfx950: the OPTIONAL
// "split-bf16" mode (SURVEY 8 f4 "bf16/split-bf16 MFMA"; never the default, never the headline row).
//
// Same contract and fused epilogue as conv3x3_wino4.hip / conv3x3_bf16.hip (reference nn.Conv2d(k=3, padding=1), model/basic.py:4-7,
// forward and - with mode-1 packed weights - input gradient).  Every fp32 operand v is written as hi + lo with hi = bf16(v) and
// lo = bf16(v - hi) (v - hi is exact in fp32), and a product a*b is replaced by THREE bf16 products, each exact in fp32, summed in
// the fp32 accumulator smallest terms first:
//     a*b ~ a_hi*b_lo + a_lo*b_hi + a_hi*b_hi          (a_lo*b_lo, ~2^-18 of the product, is dropped)
// hi + lo carries 16 significand bits, so the representation error of an operand is 2^-17 relative: measured against an fp64 conv
// the result is 3.6 .. 4.7e-6 of the output maximum at 64 .. 512 input channels (profiles/r04_split_bf16_numerics.txt; the fp32
// F(4,3) kernels: 0.8 .. 2.2e-6; the plain bf16 mode: 2e-3) - inside the 1e-5 kernel-level bound and the 1e-4 gradient tolerance the
// fp32 kernels are held to, which is why this mode is checked against the reference's fp32 oracle itself, with the fp32 tolerances.
//
// Structure = conv3x3_bf16.hip's: one workgroup = 144 output pixels x BN = 128 * NTW output channels, 8 waves, wave w owns channels
// (w * NTW + j) * 16 .. for all nine 16-pixel m-tiles.  Per 32-channel chunk:
//   * pixels: the (TR + 2) x (TW + 2) halo as TWO bf16 images (hi plane, lo plane), [pixel][32 ch] in 96 bytes per pixel (64 of data +
//     32 of padding: the stride at which ds_read_b128 of 16 consecutive pixels is conflict-free at every start offset), double
//     buffered: global fp32 -> registers -> hi / residual / lo on the VALU -> two ds_write_b64, two thirds into the previous chunk;
//     ONE barrier per chunk;
//   * weights: the wave's [16 ch][32 k] slabs of the hi and the lo packing straight from global memory into registers, in a ring
//     THREE TAPS deep (a whole chunk ahead, as the one-product kernel keeps them, would be 144 registers here);
//   * 3 MFMAs per (tap, m-tile, n-tile): 16 cycles each, i.e. 3/16 of the direct fp32 MFMA's time and 3/8 of the F(4,3) kernel's.
#include "common.h"
#include "launchers.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

struct B3Args {
    const float* x;            // [N][H][W][Cin]
    const char* wp;            // packed bf16 weights [2 (hi, lo)][9][Cin/32][Cout][32]
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
    int ps;                    // 1: output stored pixel-shuffled (r = 2)
    int ps_in;                 // 1: x is a pixel-shuffled tensor read as its sub-pixel-major view
};

constexpr int B3_MG = 9;       // m-tiles of 16 pixels per workgroup
constexpr int B3_PX = 96;      // LDS bytes per halo pixel and plane
constexpr int B3_WD = 3;       // weight ring depth (taps)
constexpr int B3_STAGE_T = 6;  // the staged halo is converted and stored after this tap

template <int NTW, int WTC>
__global__ __launch_bounds__(512) void conv3x3_bf16x3_kernel(const B3Args a) {
    constexpr int NT = 512, NU = 4, BN = 128 * NTW;
    const int WT = WTC ? WTC : a.WT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int p_bytes = (a.HT * WT + 1) * B3_PX;                        // one plane (+ the dump pixel)
    const int v_bytes = 2 * p_bytes;                                    // hi plane, lo plane

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;

    int b = blockIdx.x;
    if ((gridDim.x & 7) == 0) b = (b & 7) * (gridDim.x >> 3) + (b >> 3);     // n-tiles of a pixel tile on one XCD / L2
    int bid = b;
    const int nt = bid % a.n_tiles;  bid /= a.n_tiles;
    const int tx = bid % a.tiles_x;  bid /= a.tiles_x;
    const int ty = bid % a.tiles_y;
    const int img = bid / a.tiles_y;
    const int gy0 = ty * a.TR, gx0 = tx * a.TW;
    const int n0 = nt * BN;
    const int C32 = a.Cin >> 5;

    int a_off[B3_MG];          // lane (r, g) reads k-group g of pixel 16 i + r
#pragma unroll
    for (int i = 0; i < B3_MG; ++i) {
        const int m = i * 16 + r;
        const int trow = m / a.TW, tcol = m - trow * a.TW;
        a_off[i] = (trow * WT + tcol) * B3_PX + g * 16;
    }

    // weights: this lane's 16 bytes of slab (plane, tap, chunk), n-tile j: scalar slab offset + 1-KiB immediate per n-tile
    const int slab_bytes = a.Cout * 64;
    const unsigned b_lane = (unsigned)(((n0 + wave * NTW * 16 + r) * 32 + g * 8) * 2);
    const __amdgpu_buffer_rsrc_t w_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.wp, 0, (unsigned)((size_t)2 * 9 * a.Cin * a.Cout * 2), 0x00020000);
    auto ldw = [&](int plane, int t, int j, int cc) -> bf16x8 {
        const int so = ((plane * 9 + t) * C32 + cc) * slab_bytes;
        return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, b_lane + j * 1024, so, 0));
    };

    // staging items: (halo pixel, 4-channel group q of 8)
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
        st_off[u] = ok ? (unsigned)((pix + q * 4) * 4) : 0x80000000u;      // beyond the descriptor: the load returns zeros
        st_dst[u] = (it < n_items ? px : a.HT * WT) * B3_PX + q * 8;        // items past the halo land in the dump pixel
    }
    const __amdgpu_buffer_rsrc_t x_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)x_img, 0, (unsigned)((size_t)a.H * a.W * a.Cin * 4), 0x00020000);
    auto chunk_off = [&](int cc) -> int {
        int coff = cc * 32;
        if (a.ps_in) {
            const int sub = coff / Cq, cc0 = coff - sub * Cq;
            coff = ((sub >> 1) * (2 * a.W) + (sub & 1)) * Cq + cc0;
        }
        return coff * 4;
    };
    u32x4 sx[NU];
    auto stage_load = [&](int cc) {
        const int so = __builtin_amdgcn_readfirstlane(chunk_off(cc));
#pragma unroll
        for (int u = 0; u < NU; ++u) sx[u] = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, st_off[u], so, 0);
    };
    auto stage_store = [&](char* vdst) {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            // hi = bf16(v) (round to nearest even); lo = bf16(v - hi), the subtraction exact in fp32.  (volatile: left to itself
            // hipcc converts right behind the loads, i.e. waits for them at the top of the chunk)
            const f32x4 v = __builtin_bit_cast(f32x4, sx[u]);
            unsigned h01, h23, l01, l23;
            asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(h01) : "v"(v.x), "v"(v.y));
            asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(h23) : "v"(v.z), "v"(v.w));
            const float r0 = v.x - __builtin_bit_cast(float, h01 << 16), r1 = v.y - __builtin_bit_cast(float, h01 & 0xffff0000u);
            const float r2 = v.z - __builtin_bit_cast(float, h23 << 16), r3 = v.w - __builtin_bit_cast(float, h23 & 0xffff0000u);
            asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(l01) : "v"(r0), "v"(r1));
            asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(l23) : "v"(r2), "v"(r3));
            *(u32x2*)(vdst + st_dst[u]) = (u32x2){h01, h23};
            *(u32x2*)(vdst + p_bytes + st_dst[u]) = (u32x2){l01, l23};
        }
    };

    f32x4 acc[NTW][B3_MG];
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int i = 0; i < B3_MG; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    bf16x8 fwh[B3_WD][NTW], fwl[B3_WD][NTW];   // weight fragments (hi, lo): a ring of B3_WD taps
    bf16x8 fxh[2][3], fxl[2][3];               // pixel fragments (hi, lo): groups of three m-tiles, one group ahead

#define B3_READ_X(SET, VB, T, GRP)                                                                       \
    {                                                                                                    \
        const int to_ = (((T) / 3) * WT + (T) % 3) * B3_PX;                                              \
        _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                                  \
            fxh[SET][i] = *(const bf16x8*)((VB) + a_cur[(GRP) * 3 + i] + to_);                           \
            fxl[SET][i] = *(const bf16x8*)((VB) + p_bytes + a_cur[(GRP) * 3 + i] + to_);                 \
        }                                                                                                \
    }
#define B3_MFMA(SET, SLOT, GRP)                                                                          \
    _Pragma("unroll") for (int i = 0; i < 3; ++i)                                                        \
        _Pragma("unroll") for (int j = 0; j < NTW; ++j) {                                                \
            f32x4 c_ = acc[j][(GRP) * 3 + i];                                                            \
            c_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fwh[SLOT][j], fxl[SET][i], c_, 0, 0, 0);        \
            c_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fwl[SLOT][j], fxh[SET][i], c_, 0, 0, 0);        \
            c_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fwh[SLOT][j], fxh[SET][i], c_, 0, 0, 0);        \
            acc[j][(GRP) * 3 + i] = c_;                                                                  \
        }

    // ---- prologue: chunk 0 staged synchronously, the first B3_WD taps' weights -------------------------------------------------
    stage_load(0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < B3_WD; ++t) {
#pragma unroll
        for (int j = 0; j < NTW; ++j) { fwh[t][j] = ldw(0, t, j, 0); fwl[t][j] = ldw(1, t, j, 0); }
        __builtin_amdgcn_sched_barrier(0);
    }
    stage_store(smem);
    __syncthreads();

#pragma unroll 1
    for (int c = 0; c < C32; ++c) {
        const int cur_off = (c & 1) * v_bytes;
        int a_cur[B3_MG];
#pragma unroll
        for (int i = 0; i < B3_MG; ++i) a_cur[i] = a_off[i] + cur_off;
        char* const vnext = smem + ((c & 1) ^ 1) * v_bytes;
        // (no branch in the loop: the last chunk prefetches itself again, so that hipcc counts the outstanding loads exactly)
        const int cn = c + 1 < C32 ? c + 1 : c;
        stage_load(cn);
        B3_READ_X(0, smem, 0, 0)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
#pragma unroll
            for (int grp = 0; grp < 3; ++grp) {
                const int G = t * 3 + grp, Gn = G + 1;
                if (Gn < 27) B3_READ_X(Gn & 1, smem, Gn / 3, Gn % 3)
                __builtin_amdgcn_sched_barrier(0);         // keep the prefetch ahead of the MFMA group
                B3_MFMA(G & 1, t % B3_WD, grp)
                __builtin_amdgcn_sched_barrier(0);
            }
            // this tap's weights are consumed: fetch tap t + B3_WD (of this chunk, or of the next) into their registers
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                const int tt = t + B3_WD;
                fwh[t % B3_WD][j] = ldw(0, tt < 9 ? tt : tt - 9, j, tt < 9 ? c : cn);
                fwl[t % B3_WD][j] = ldw(1, tt < 9 ? tt : tt - 9, j, tt < 9 ? c : cn);
            }
            if (t == B3_STAGE_T) { stage_store(vnext); __builtin_amdgcn_sched_barrier(0); }
        }
        __syncthreads();
    }
#undef B3_READ_X
#undef B3_MFMA

    // ---- epilogue: lane (r, g) holds channels co .. co + 3 of pixel 16 i + r -----------------------------------------------------
    const size_t img_out = (size_t)img * a.H * a.W;
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        const int co = n0 + (wave * NTW + j) * 16 + g * 4;
        f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
        if (a.bias) bias4 = *(const f32x4*)(a.bias + co);
#pragma unroll
        for (int ib = 0; ib < B3_MG; ib += 3) {
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
// weight packing: OIHW fp32 -> [2 (hi, lo)][9][R/32][Nn][32] bf16; hi = bf16(w), lo = bf16(w - hi)
//   mode 0 (forward): out[p][t][c][n][k] from w[o = unperm(n)][i = 32c + k][t];  mode 1 (dgrad): w[o = unperm(32c + k)][i = n][8 - t]
__device__ __forceinline__ void b3_split(float v, __bf16* hi, __bf16* lo) {
    const __bf16 h = (__bf16)v;
    *hi = h;
    *lo = (__bf16)(v - (float)h);
}

__global__ void pack_bf16x3_kernel(const float* __restrict__ w, __bf16* __restrict__ out, int O, int I, int mode, int ps) {
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
        b3_split(w[((long)o * I + i) * 9 + (mode == 0 ? t : 8 - t)], out + e, out + total + e);
    }
}

int pesr_pack_conv3x3_bf16x3_launch(const float* w, void* out, int O, int I, int mode, int ps, hipStream_t stream) {
    if (O % 32 || I % 32 || (mode != 0 && mode != 1) || (ps && O % 128)) return PESR_EINVAL;
    const long total = 9L * O * I;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_bf16x3_kernel, dim3(grid), dim3(256), 0, stream, w, (__bf16*)out, O, I, mode, ps);
    return pesr_launch_status();
}

namespace {
struct B3Plan { int TR, TW, HT, WT, tiles_x, tiles_y, n_tiles, ntw, bn; long tiles; size_t lds; int score; };

// Tile shape TR x TW == 144 pixels with the least out-of-image area whose halo fits the four staging items per thread.
static bool b3_plan(int N, int H, int W, int Cin, int Cout, B3Plan* p, int min_wgs = 128) {
    if (N < 1 || H < 1 || W < 1 || Cin % 32 || Cin < 32 || Cout % 128) return false;
    if ((size_t)H * W * Cin * 4 >= ((size_t)1 << 31)) return false;   // one image per buffer descriptor, offsets below 2^31
    long best = -1;
    for (int TW = 1; TW <= 144; ++TW) {
        if (144 % TW) continue;
        const int TR = 144 / TW, HT = TR + 2, WT = TW + 2;
        if (HT * WT * 8 > 2048) continue;
        const long cover = (long)pesr_cdiv(H, TR) * TR * pesr_cdiv(W, TW) * TW;
        const long score = cover * 8192 + (TW % 16 ? 4096 : 0) + (long)HT * WT;
        if (best < 0 || score < best) { best = score; p->TR = TR; p->TW = TW; }
    }
    if (best < 0) return false;
    p->HT = p->TR + 2; p->WT = p->TW + 2;
    p->tiles_y = pesr_cdiv(H, p->TR); p->tiles_x = pesr_cdiv(W, p->TW);
    // 256 output channels per workgroup unless that leaves fewer than 128 workgroups: then 128 each
    p->ntw = (Cout % 256 == 0 && (long)N * p->tiles_y * p->tiles_x * (Cout / 256) >= 128) ? 2 : 1;
    p->bn = 128 * p->ntw;
    p->n_tiles = Cout / p->bn;
    p->tiles = (long)N * p->tiles_y * p->tiles_x * p->n_tiles;
    p->lds = (size_t)4 * (p->HT * p->WT + 1) * B3_PX;          // two buffers x (hi, lo)
    const double cover_eff = (double)H * W / ((double)p->tiles_y * p->TR * p->tiles_x * p->TW);
    p->score = (p->tiles >= min_wgs && p->lds <= 160 * 1024) ? (int)(1000.0 * cover_eff) : 0;
    return true;
}
}  // namespace

// per-mille of tile area inside the image (0: unsupported shape, or fewer than min_wgs workgroups)
int pesr_conv3x3_bf16x3_score_impl(int N, int H, int W, int Cin, int Cout, int min_wgs) {
    B3Plan p;
    if (!b3_plan(N, H, W, Cin, Cout, &p, min_wgs)) return 0;
    return p.score;
}

int pesr_conv3x3_bf16x3_launch(const float* x, const void* wp, const float* bias, const float* skip, const float* mask, float* y,
                               int N, int H, int W, int Cin, int Cout, float alpha, int act, float slope, int ps, int ps_in,
                               hipStream_t stream) {
    B3Plan p;
    if (!b3_plan(N, H, W, Cin, Cout, &p, 1) || p.lds > 160 * 1024) return PESR_EINVAL;
    if (ps && (Cout % (4 * p.bn) || skip || mask)) return PESR_EINVAL;          // an n-tile must stay inside one sub-pixel plane
    if (ps_in && Cin % 128) return PESR_EINVAL;                                 // a 32-channel chunk must stay inside one sub-pixel
    B3Args a{};
    a.x = x; a.wp = (const char*)wp; a.bias = bias; a.skip = skip; a.mask = mask; a.y = y;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.alpha = alpha; a.slope = slope; a.act = act; a.ps = ps; a.ps_in = ps_in;
    a.TR = p.TR; a.TW = p.TW; a.HT = p.HT; a.WT = p.WT;
    a.tiles_x = p.tiles_x; a.tiles_y = p.tiles_y; a.n_tiles = p.n_tiles;
#define B3_LAUNCH(NTW_, WT_)                                                                                              \
    {                                                                                                                     \
        auto kern = conv3x3_bf16x3_kernel<NTW_, WT_>;                                                                     \
        static PesrDeviceOnce once;                                                                                       \
        once([&] { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }); \
        hipLaunchKernelGGL(kern, dim3((unsigned)p.tiles), dim3(512), p.lds, stream, a);                                   \
    }
#define B3_BY_WT(NTW_)                                   \
    switch (p.WT) {                                      \
        case 50: B3_LAUNCH(NTW_, 50) break;              \
        case 26: B3_LAUNCH(NTW_, 26) break;              \
        case 14: B3_LAUNCH(NTW_, 14) break;              \
        default: B3_LAUNCH(NTW_, 0) break;               \
    }
    if (p.ntw == 2) { B3_BY_WT(2) } else { B3_BY_WT(1) }
#undef B3_BY_WT
#undef B3_LAUNCH
    return pesr_launch_status();
}
