// 3x3 stride-1 convolution with a 1-D Winograd F(2,3) transform along x, on the fp32-input MFMA, gfx950.
//
// Same contract as conv3x3_mfma.hip (reference nn.Conv2d(k=3, padding=1), model/basic.py:4-7, forward and - with
// dgrad-transformed weights - input gradient) for even widths and Cout % 128 == 0, with 2/3 of the multiplies:
// for an output pixel pair (2t, 2t+1) of a row and the input columns d0..d3 = x[2t-1 .. 2t+2],
//     V  = [d0 - d2, d1 + d2, d2 - d1, d1 - d3]                        (input transform, per row and channel)
//     U  = [g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2]                        (weight transform along kx, done by pack.hip)
//     M_xi = sum_{ky, ci} V_xi[row + ky - 1][ci] * U_xi[ky][ci]         (4 accumulators instead of 2 outputs x 3 taps)
//     y[2t] = M0 + M1 + M2,   y[2t+1] = M1 - M2 - M3.
// So a 16-channel chunk has 3 (ky) x 4 (xi) = 12 weight slabs of [Cout][16] instead of 9 taps, and the GEMM rows are
// x-tiles (pixel pairs): 12 slab-MFMAs per 2 pixels instead of 18.
//
// One workgroup = 144 x-tiles (TR rows x TXT tiles, 288 output pixels) x 128 output channels, 8 waves, each wave owning
// 16 channels x 4 xi x 9 m-tiles of accumulators (144 VGPRs).  Per chunk the raw input halo arrives by LDS-DMA (one chunk
// ahead), a cooperative pass turns it into V ([row][xi][x-tile][16ch], 64 B per entry, 16-byte k-groups XOR-swizzled by
// (x-tile >> 1) & 3: the LDS serves a ds_read_b128 eight lanes (128 B = 32 banks) per clock, and with this key eight
// consecutive x-tiles hit eight different 16-byte slots - SQ_LDS_BANK_CONFLICT drops from 40.7 M to 15.5 M per launch; the
// weights carry the same swizzle from pack time) and
// the 12 slabs stream through an 8-slot LDS-DMA ring that each wave fences for itself (it reads only the 1 KiB of a slab it
// DMA'd); fragments are read one 3-m-tile group ahead.  Layers with too few tiles split the Cin chunks over workgroups
// (raw partial sums + the direct kernel's fixed-order finish kernel).  The output transform happens in registers before the tile leaves
// through LDS as coalesced 16-byte stores with the usual fused epilogue (bias, scale, ReLU mask, skip, activation).
#include <mutex>
#include "common.h"
#include "launchers.h"
#include "wino_pack.h"

__device__ __attribute__((aligned(16))) const float g_wino_zero16[4] = {0.f, 0.f, 0.f, 0.f};

__device__ __forceinline__ void wino_dma16(const float* gsrc, char* lds_piece) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_piece, 16, 0, 0);
}

struct WinoArgs {
    const float* x;     // [N][H][W][Cin]
    const float* wp;    // packed, transformed weights [3*4][Cin/16][Cout][16]
    const float* bias;  // [Cout] or null
    const float* skip;  // [N][H][W][Cout] or null
    const float* mask;  // [N][H][W][Cout] or null : result zeroed where mask <= 0
    float* y;           // [N][H][W][Cout]
    int N, H, W, Cin, Cout;
    int TR, TXT;        // tile: TR output rows x TXT x-tiles (TR * TXT == 144)
    int tiles_x, tiles_y, n_tiles;
    int HT, WT;         // raw halo: TR + 2 rows x 2*TXT + 2 columns
    float alpha, slope;
    int act;
    int ps;             // 1: output stored pixel-shuffled (r = 2): packed channel (2*si+sj)*C + c -> y[n][2oy+si][2ox+sj][c], C = Cout/4
    int ps_in;          // 1: x is a pixel-shuffled tensor [N][2H][2W][Cin/4] read as its sub-pixel-major [N][H][W][Cin] view
    int ksplit;         // > 1: the Cin chunks are split over ksplit workgroups per tile; raw partial sums go to slab[ks][...]
    int chunks_per_split;
    float* slab;
};

constexpr int WINO_NT = 512, WINO_BN = 128, WINO_MG = 9, WINO_RING = 8;
constexpr int WINO_SLAB = WINO_BN * 64;   // bytes of one weight slab [128][16]
constexpr int WINO_HL = 4;                // raw-halo DMA pieces per wave (<= 32 pieces = 512 pixels)

__global__ __launch_bounds__(WINO_NT) void conv3x3_wino_kernel(const WinoArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int halo_pix = a.HT * a.WT;
    const int raw_bytes = ((halo_pix * 64 + 1023) / 1024) * 1024;
    const int v_bytes = a.HT * 4 * a.TXT * 64;
    char* const raw = smem;
    char* const vbuf = smem + raw_bytes;
    char* const ring = vbuf + v_bytes;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;

    const int tiles_total = a.n_tiles * a.tiles_x * a.tiles_y * a.N;
    const int ks = blockIdx.x / tiles_total;              // split-K slice (0 when ksplit == 1)
    int bid = blockIdx.x - ks * tiles_total;
    const int nt = bid % a.n_tiles;  bid /= a.n_tiles;
    const int tx = bid % a.tiles_x;  bid /= a.tiles_x;
    const int ty = bid % a.tiles_y;
    const int img = bid / a.tiles_y;
    const int gy0 = ty * a.TR, gt0 = tx * a.TXT;          // first output row / first x-tile of the tile
    const int n0 = nt * WINO_BN;
    const int C16T = a.Cin >> 4;                           // chunks in the packed weights
    const int CB = ks * a.chunks_per_split;                // this workgroup's chunk range [CB, CB + C16)
    const int C16 = (C16T - CB) < a.chunks_per_split ? (C16T - CB) : a.chunks_per_split;

    // per-lane LDS offsets: A = V[row][xi][x-tile][16ch] (row and xi added per slab), B = slab[channel][16]
    int a_off[WINO_MG];
#pragma unroll
    for (int i = 0; i < WINO_MG; ++i) {
        const int m = i * 16 + r;
        const int trow = m / a.TXT, txt = m - trow * a.TXT;
        a_off[i] = (trow * 4 * a.TXT + txt) * 64 + (g ^ ((txt >> 1) & 3)) * 16;   // k-group g of x-tile txt sits at g ^ ((txt>>1)&3)
    }
    const int b_off = (wave * 16 + r) * 64 + (g ^ ((r >> 1) & 3)) * 16;                  // same swizzle, baked into the packed weights
    const int xi_stride = a.TXT * 64;                      // bytes between the xi planes of a V row

    // raw halo: per-lane global offsets of this wave's DMA pieces (piece = 16 pixels x 64 B)
    const float* const xi_img = a.x + (size_t)img * a.H * a.W * a.Cin;
    int h_src[WINO_HL];
#pragma unroll
    for (int k = 0; k < WINO_HL; ++k) {
        const int e = (wave + k * 8) * 64 + lane;          // float4 unit inside the halo image
        const int hp = e >> 2, q = e & 3;
        int off = -2;
        if (hp < halo_pix) {
            const int hy = hp / a.WT, hx = hp - hy * a.WT;
            const int iy = gy0 - 1 + hy, ix = 2 * gt0 - 1 + hx;
            if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)
                off = a.ps_in ? ((2 * iy) * (2 * a.W) + 2 * ix) * (a.Cin >> 2) + q * 4 : (iy * a.W + ix) * a.Cin + q * 4;
            else
                off = -1;
        }
        h_src[k] = off;
    }
    auto dma_raw = [&](int c) {
        int coff = (CB + c) * 16;
        if (a.ps_in) {   // chunk c = channels [16c, 16c+16) of sub-pixel `sub`: one pixel of the shuffled tensor
            const int C = a.Cin >> 2;
            const int sub = coff / C, cc0 = coff - sub * C;
            coff = ((sub >> 1) * (2 * a.W) + (sub & 1)) * C + cc0;
        }
#pragma unroll
        for (int k = 0; k < WINO_HL; ++k) {
            if (h_src[k] != -2) {
                const float* src = h_src[k] >= 0 ? xi_img + h_src[k] + coff : g_wino_zero16;
                wino_dma16(src, raw + (wave + k * 8) * 1024);
            }
        }
    };
    // weight slab s = (c*3 + ky)*4 + xi of the kernel's own order; packed as [ky*4 + xi][c][Cout][16]: one 1-KiB piece per wave
    const float* const wn = a.wp + (size_t)n0 * 16 + wave * 256 + lane * 4;
    const size_t slab_stride = (size_t)a.Cout * 16;
    const int nslab = C16 * 12;
    // DMA cursor: slab sd = (chunk cd, tap td) lives at wn + woff; past the last slab it stays there (harmless re-fetches)
    int sd = 0, cd = 0, td = 0;
    size_t woff = (size_t)CB * slab_stride;
    auto dma_next = [&]() {
        wino_dma16(wn + woff, ring + (sd & (WINO_RING - 1)) * WINO_SLAB + wave * 1024);
        ++sd;
        if (sd < nslab) {
            if (++td == 12) { td = 0; ++cd; woff = (size_t)(CB + cd) * slab_stride; }
            else woff += (size_t)C16T * slab_stride;
        }
    };
    // raw -> V: item = (halo row, x-tile, 4-channel group)
    // (at most 2 items per thread: HT*WT <= 512 pixels; both items' reads are issued before the first write, and the item
    //  coordinates - the same for every chunk - are computed once)
    const int tr_items = a.HT * a.TXT * 4;
    int tr_src[2], tr_dst[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int it = tid + u * WINO_NT;
        const int q = it & 3, rest = it >> 2;
        const int hrow = rest / a.TXT, txt = rest - hrow * a.TXT;
        tr_src[u] = it < tr_items ? (hrow * a.WT + 2 * txt) * 64 + q * 16 : -1;
        tr_dst[u] = (hrow * 4 * a.TXT + txt) * 64 + (q ^ ((txt >> 1) & 3)) * 16;
    }
    auto transform = [&]() {
        f32x4 d[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u)
            if (tr_src[u] >= 0) {
                const char* src = raw + tr_src[u];
                d[u][0] = *(const f32x4*)(src); d[u][1] = *(const f32x4*)(src + 64);
                d[u][2] = *(const f32x4*)(src + 128); d[u][3] = *(const f32x4*)(src + 192);
            }
#pragma unroll
        for (int u = 0; u < 2; ++u)
            if (tr_src[u] >= 0) {
                char* dst = vbuf + tr_dst[u];
                *(f32x4*)(dst) = d[u][0] - d[u][2];
                *(f32x4*)(dst + xi_stride) = d[u][1] + d[u][2];
                *(f32x4*)(dst + 2 * xi_stride) = d[u][2] - d[u][1];
                *(f32x4*)(dst + 3 * xi_stride) = d[u][1] - d[u][3];
            }
    };

    f32x4 acc[4][WINO_MG];
#pragma unroll
    for (int x4 = 0; x4 < 4; ++x4)
#pragma unroll
        for (int i = 0; i < WINO_MG; ++i) acc[x4][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // Fragments: the 9 m-tiles of a slab are walked in 3 groups of 3; while a group's 12 MFMAs run (k-step outer, m-tile
    // inner: an accumulator is revisited every 3rd MFMA) the next group's 3 A fragments are read into the other half of
    // fa[], and the next slab's B fragment into the other fb[] - 24 + 8 fragment VGPRs next to the 144 accumulators.
    // The weight ring needs no barrier: a wave reads only the 16 channels (1 KiB of every slab) that it DMA'd itself, so
    // s_waitcnt vmcnt(0) once per ky is enough; barriers only fence the shared raw / V buffers, twice per chunk.
    f32x4 fa[2][3], fb[2];
#define WINO_READ_A(FA, KY, XI, GRP)                                                                     \
    {                                                                                                    \
        const char* const vb_ = vbuf + ((KY) * 4 + (XI)) * xi_stride;                                    \
        _Pragma("unroll") for (int i = 0; i < 3; ++i) FA[i] = *(const f32x4*)(vb_ + a_off[(GRP) * 3 + i]); \
    }
#define WINO_READ_B(FB, S_) FB = *(const f32x4*)(ring + ((S_) & (WINO_RING - 1)) * WINO_SLAB + b_off);
#define WINO_MFMA(FA, FB, XI, GRP)                                                                       \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk)                                                     \
        _Pragma("unroll") for (int i = 0; i < 3; ++i)                                                    \
            acc[XI][(GRP) * 3 + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(FA[i][kk], FB[kk], acc[XI][(GRP) * 3 + i], 0, 0, 0);

    // ---- prologue: chunk 0's halo, slabs 0..4 ------------------------------------------------------------------------------
    dma_raw(0);
    dma_next(); dma_next(); dma_next(); dma_next(); dma_next();
    __syncthreads();
    transform();
    __syncthreads();
    if (C16 > 1) dma_raw(1);
    WINO_READ_B(fb[0], 0)
    int s = 0;                                             // first slab of the current ky step
#pragma unroll 1
    for (int c = 0; c < C16; ++c) {
        WINO_READ_A(fa[0], 0, 0, 0)                        // V of this chunk exists only now; the slab's B fragment is already in fb[0]
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0): this wave's pieces of slabs s+1 .. s+4 (issued a step ago) are in LDS
            dma_next(); dma_next(); dma_next(); dma_next(); // slabs s+5 .. s+8 -> the slots of s-3 .. s (read by this wave already)
#pragma unroll
            for (int st = 0; st < 12; ++st) {
                const int xi = st / 3, grp = st % 3, cur = st & 1;
                if (st < 11) WINO_READ_A(fa[cur ^ 1], ky, (st + 1) / 3, (st + 1) % 3)
                else if (ky < 2) WINO_READ_A(fa[cur ^ 1], ky + 1, 0, 0)     // across a chunk boundary V is rebuilt first
                if (grp == 0) WINO_READ_B(fb[(xi & 1) ^ 1], s + xi + 1)
                __builtin_amdgcn_sched_barrier(0);         // keep the prefetch ahead of the MFMA group (hipcc sinks it next to its use)
                WINO_MFMA(fa[cur], fb[xi & 1], xi, grp)
                __builtin_amdgcn_sched_barrier(0);
            }
            s += 4;
        }
        if (c + 1 < C16) {
            __syncthreads();                               // every wave is done with V; raw holds chunk c+1 (DMA'd a chunk ago)
            transform();
            __syncthreads();
            if (c + 2 < C16) dma_raw(c + 2);
        }
    }
    __syncthreads();
#undef WINO_READ_A
#undef WINO_READ_B
#undef WINO_MFMA

    // ---- epilogue: output transform in registers, tile through LDS, coalesced stores ------------------------------------
    constexpr int RS = WINO_BN * 4 + 16;                   // padded row stride of the staged tile
    constexpr int C4 = WINO_BN / 4;
    char* const ob = smem;
    const int prow = 2 * a.TXT;                            // output pixels per tile row
#pragma unroll
    for (int i = 0; i < WINO_MG; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int m = i * 16 + g * 4 + jj;
            const int trow = m / a.TXT, txt = m - trow * a.TXT;
            const float m0 = acc[0][i][jj], m1 = acc[1][i][jj], m2 = acc[2][i][jj], m3 = acc[3][i][jj];
            char* o = ob + (trow * prow + 2 * txt) * RS + (wave * 16 + r) * 4;
            *(float*)(o) = (m0 + m1) + m2;
            *(float*)(o + RS) = (m1 - m2) - m3;
        }
    __syncthreads();
    const size_t img_out = (size_t)img * a.H * a.W;
    const int npix = 288;
    for (int u = tid; u < npix * C4; u += WINO_NT) {
        const int p = u / C4, c4 = u - p * C4;
        const int co = n0 + c4 * 4;
        const int py = p / prow, px = p - py * prow;
        const int oy = gy0 + py, ox = 2 * gt0 + px;
        if (oy >= a.H || ox >= a.W) continue;
        f32x4 v = *(const f32x4*)(ob + p * RS + c4 * 16);
        size_t idx;
        if (a.ps) {   // packed channel co = (2*si+sj)*C + c  ->  out[n][2*oy+si][2*ox+sj][c]
            const int C = a.Cout >> 2;
            const int sub = co / C, cc = co - sub * C;
            idx = (((size_t)img * (2 * a.H) + 2 * oy + (sub >> 1)) * (2 * a.W) + 2 * ox + (sub & 1)) * C + cc;
        } else {
            idx = (img_out + (size_t)oy * a.W + ox) * a.Cout + co;
        }
        if (a.ksplit > 1) {   // raw partial sums; the finish kernel applies the epilogue
            *(f32x4*)(a.slab + (size_t)ks * ((size_t)a.N * a.H * a.W * a.Cout) + idx) = v;
            continue;
        }
        if (a.bias) v += *(const f32x4*)(a.bias + co);
        v *= a.alpha;
        if (a.mask) {
            const f32x4 mk = *(const f32x4*)(a.mask + idx);
            v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f; v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
        }
        if (a.skip) v += *(const f32x4*)(a.skip + idx);
        if (a.act == PESR_ACT_RELU) {
            v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
        } else if (a.act == PESR_ACT_LRELU) {
            v.x = v.x > 0.f ? v.x : v.x * a.slope; v.y = v.y > 0.f ? v.y : v.y * a.slope;
            v.z = v.z > 0.f ? v.z : v.z * a.slope; v.w = v.w > 0.f ? v.w : v.w * a.slope;
        }
        *(f32x4*)(a.y + idx) = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// weight transform + packing (wino_pack.h)
__global__ void pack_wino_kernel(const float* __restrict__ w, float* __restrict__ out, int O, int I, int mode, int ps) {
    const long total = 12L * O * I;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x)
        out[e] = pesr_wino_pack_elem(w, O, I, mode, ps, e);
}

int pesr_pack_conv3x3_wino_launch(const float* w, float* out, int O, int I, int mode, int ps, hipStream_t stream) {
    if (O % 16 || I % 16 || (mode != 0 && mode != 1) || (ps && O % 64)) return PESR_EINVAL;
    const long total = 12L * O * I;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_wino_kernel, dim3(grid), dim3(256), 0, stream, w, out, O, I, mode, ps);
    return pesr_launch_status();
}

int pesr_conv3x3_wino_supported_impl(int N, int H, int W, int Cin, int Cout) {
    return N > 0 && H > 0 && W >= 2 && W % 2 == 0 && Cin % 16 == 0 && Cin >= 16 && Cout % WINO_BN == 0;
}

int pesr_conv3x3_wino_launch(const float* x, const float* wp, const float* bias, const float* skip, const float* mask, float* y,
                             int N, int H, int W, int Cin, int Cout, float alpha, int act, float slope, int ps, int ps_in,
                             void* ws, size_t ws_bytes, hipStream_t stream) {
    if (!pesr_conv3x3_wino_supported_impl(N, H, W, Cin, Cout)) return PESR_EINVAL;
    if (ps && (Cout % 16 || skip || mask)) return PESR_EINVAL;
    if (ps_in && Cin % 64) return PESR_EINVAL;
    WinoArgs a{};
    a.x = x; a.wp = wp; a.bias = bias; a.skip = skip; a.mask = mask; a.y = y;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.alpha = alpha; a.slope = slope; a.act = act; a.ps = ps; a.ps_in = ps_in;
    // tile shape: TR x TXT == 144 x-tiles, least out-of-image area, within the LDS budget and the raw-halo DMA pieces
    const int XT = W / 2;
    long best = -1;
    for (int TXT = 1; TXT <= 144; ++TXT) {
        if (144 % TXT) continue;
        const int TR = 144 / TXT;
        const int HT = TR + 2, WT = 2 * TXT + 2;
        const size_t raw_bytes = ((size_t)HT * WT * 64 + 1023) / 1024 * 1024;
        const size_t lds = raw_bytes + (size_t)HT * 4 * TXT * 64 + (size_t)WINO_RING * WINO_SLAB;
        if (lds > 160 * 1024 || HT * WT > WINO_HL * 8 * 16) continue;
        const long cover = (long)pesr_cdiv(H, TR) * TR * pesr_cdiv(XT, TXT) * TXT;
        const long score = cover * 4096 + (long)HT * WT;
        if (best < 0 || score < best) { best = score; a.TR = TR; a.TXT = TXT; }
    }
    if (best < 0) return PESR_EINVAL;
    a.HT = a.TR + 2; a.WT = 2 * a.TXT + 2;
    a.tiles_y = pesr_cdiv(H, a.TR); a.tiles_x = pesr_cdiv(XT, a.TXT); a.n_tiles = Cout / WINO_BN;
    const size_t raw_bytes = ((size_t)a.HT * a.WT * 64 + 1023) / 1024 * 1024;
    size_t lds = raw_bytes + (size_t)a.HT * 4 * a.TXT * 64 + (size_t)WINO_RING * WINO_SLAB;
    const size_t lds_out = (size_t)288 * (WINO_BN * 4 + 16);
    if (lds_out > lds) lds = lds_out;
    static PesrDeviceOnce attr_once;
    attr_once([&] {
        (void)hipFuncSetAttribute((const void*)conv3x3_wino_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    const long tiles = (long)N * a.tiles_y * a.tiles_x * a.n_tiles;
    // split-K over the Cin chunks when the tiles alone cannot fill the 256 CUs (the 24x24 512-channel layers)
    const int C16T = Cin / 16;
    a.ksplit = 1; a.chunks_per_split = C16T; a.slab = (float*)ws;
    const size_t out_bytes = (size_t)N * H * W * Cout * sizeof(float);
    if (ws && tiles < 160 && !ps && C16T >= 8) {
        int want = (int)((256 + tiles - 1) / tiles);
        if (want > 8) want = 8;
        if (want > C16T / 4) want = C16T / 4;
        while (want > 1 && (size_t)want * out_bytes > ws_bytes) --want;
        if (want > 1) {
            a.chunks_per_split = (C16T + want - 1) / want;
            a.ksplit = (C16T + a.chunks_per_split - 1) / a.chunks_per_split;
        }
    }
    hipLaunchKernelGGL(conv3x3_wino_kernel, dim3((unsigned)(tiles * a.ksplit)), dim3(WINO_NT), lds, stream, a);
    if (a.ksplit > 1)
        return pesr_conv_splitk_finish_launch((const float*)ws, bias, skip, mask, y, (long)(out_bytes / sizeof(float)), Cout, a.ksplit,
                                              alpha, act, slope, stream);
    return pesr_launch_status();
}
