// 3x3 convolution as implicit GEMM on the fp32-input MFMA (v_mfma_f32_16x16x4_f32), gfx950.
//
// Replaces the reference's nn.Conv2d(k=3, padding=1, stride in {1,2}) forward
// (reference model/basic.py:4-7) and, with tap-flipped / transposed packed weights, its
// input-gradient (dgrad).  Layout: activations NHWC fp32, weights pre-packed by pack.hip into
// [tap][Cin/16][Cout][16] so that one (tap, 16-channel chunk) "slab" is one contiguous block.
//
// Work decomposition (one workgroup = one tile of MT output pixels x BN output channels):
//   * the (TH x TW) pixel tile's input halo for one 16-channel chunk sits in LDS ([pixel][16ch],
//     64 B per pixel) and is reused by all taps; it is double buffered across chunks;
//   * one weight slab [BN][16] per (chunk, tap) sits in LDS, double buffered across slabs;
//   * every lane feeds 4 consecutive k-steps of the MFMA from ONE ds_read_b128 per operand:
//     lane l holds A[pixel = l&15][k = l>>4]; we let MFMA k-slot g of step kk stand for channel
//     4*g + kk of the chunk, so the lane's four A values (and four B values) are contiguous.
//   * global -> LDS staging is LDS-DMA (global_load_lds_dwordx4) into a 4-slot weight ring, 3-4 slabs ahead;
//     the MFMA fragments are read from LDS one slab ahead into a second register set, so ds_read latency
//     and bank conflicts hide under the previous slab's MFMAs; one barrier per TWO slabs; the epilogue
//     leaves through LDS as coalesced 16-byte stores.  (Register-staged fallbacks for 1-3 tap problems
//     and the 3-channel input.)
// fp32 MFMA is an exact k-ordered fmaf chain, so results differ from a CPU conv only by
// summation order.
#include <mutex>
#include "common.h"
#include "launchers.h"

__device__ __attribute__((aligned(16))) const float g_zero16[4] = {0.f, 0.f, 0.f, 0.f};   // source of the zero padding for LDS-DMA

__device__ __forceinline__ void lds_dma16(const float* gsrc, char* lds_wave_base) {
    // one 1-KiB piece: lane l copies 16 B from its own gsrc to lds_wave_base + 16*l (the LDS base is wave-uniform)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

struct ConvArgs {
    const float* x;     // [N][H][W][Cin]
    const float* wp;    // packed weights [9][Cin/16][Cout][16]
    const float* bias;  // [Cout] or null
    const float* skip;  // [N][OH][OW][Cout] or null : added after scaling
    const float* mask;  // [N][OH][OW][Cout] or null : result zeroed where mask <= 0
    float* y;           // [N][OH][OW][Cout]
    int N, H, W, Cin, Cout, OH, OW;
    int GH, GW;                 // iteration domain per image (== OH, OW for a plain conv)
    int TH, TW, tiles_x, tiles_y, n_tiles;
    int in_oy, in_ox;           // halo origin in the input = g0*S + in_o
    int HT, WT;                 // halo tile extent (rows, cols)
    int out_my, out_ay, out_mx, out_ax;  // output coordinate = g*out_m + out_a
    int ntaps;
    // per tap one byte: dy | dx << 2 | weight_tap << 4 (dy, dx: position inside the halo tile).  Kept in two
    // scalars instead of arrays: an indexed kernarg read is an s_load, whose lgkmcnt(0) wait would also drain
    // the LDS reads that are deliberately left in flight across the MFMA block.
    unsigned long long tap_lo;  // taps 0..7
    unsigned tap_hi;            // tap 8
    float alpha, slope;
    int act;
    int ps;                     // 1: output channels are stored pixel-shuffled (r=2), Cout = 4*C
    int cin_real;               // channels physically present in x (3 for the RGB layers; Cin is then 16, zero padded)
    int cout_store;             // channels physically present in y (3 for the ->RGB layers; Cout is then 64, zero padded)
    int ksplit;                 // > 1: the Cin chunks are split over ksplit workgroups per tile; raw partial sums go to
    int chunks_per_split;       //      slab[ks][...] and conv_splitk_finish_kernel applies the epilogue (small-M layers)
    float* slab;
    size_t slab_bytes;
    int ps_in;                  // 1: x is a pixel-shuffled tensor [N][2H][2W][Cin/4] read as its
                                //    un-shuffled, sub-pixel-major [N][H][W][Cin] view (dgrad of a PS conv)
    // ---- BatchNorm sums from the epilogue (round 6; common.h BnEpi - the kernel's LAST parameter): the first row of this problem ----
    int bn_row0;
    // host only (planning):
    int bn_mode, bn_cap;
    int dry;                    // host only: plan, report bn_rows, do not launch
    long bn_rows;               // host only: rows this problem writes (0: the fused form does not cover it - split-K, odd channel counts)
};

__device__ __forceinline__ unsigned tap_code(const ConvArgs& a, int t) {
    return t < 8 ? (unsigned)(a.tap_lo >> (8 * t)) & 0xffu : a.tap_hi;
}

// The kernel body as a device function of (args, logical block id, blocks in this problem's grid): conv3x3_mfma_kernel is the
// one-problem launch; conv3x3_s2dgrad4_kernel runs the four parity classes of a stride-2 input gradient in ONE launch.
template <int WAVES_M, int WAVES_N, int WM, int WN, int S, int HL, int MODE>
__device__ __forceinline__ void conv3x3_mfma_body(const ConvArgs& a, const int block_id, const int grid_blocks, const unsigned bn_kernarg_off) {
    constexpr int NT = WAVES_M * WAVES_N * 64;
    constexpr int BN = WAVES_N * WN * 16;
    constexpr int WL = (BN * 4 + NT - 1) / NT;  // float4 units per thread per weight slab
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int halo_pix = a.HT * a.WT;
    const int halo_bytes = ((halo_pix * 64 + 255) / 256) * 256;
    char* const halo0 = smem;
    char* const halo1 = smem + halo_bytes;
    char* const wb0 = smem + 2 * halo_bytes;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;
    const int r = lane & 15, g = lane >> 4;

    // Workgroups b and b + 8 share an XCD (round-robin dispatch): give every XCD a contiguous range of logical tiles, n-tile
    // fastest, so that the n-tiles of one pixel tile - which read the same halo - share an L2.
    int lb = block_id;
    if ((grid_blocks & 7) == 0) lb = (lb & 7) * (grid_blocks >> 3) + (lb >> 3);
    const int tiles_total = a.n_tiles * a.tiles_x * a.tiles_y * a.N;
    const int ks = lb / tiles_total;                    // split-K slice (0 when ksplit == 1)
    int bid = lb - ks * tiles_total;
    const int nt = bid % a.n_tiles;  bid /= a.n_tiles;
    const int tx = bid % a.tiles_x;  bid /= a.tiles_x;
    const int ty = bid % a.tiles_y;
    const int img = bid / a.tiles_y;
    const int gy0 = ty * a.TH, gx0 = tx * a.TW;
    const int n0 = nt * BN;
    const int C16T = a.Cin >> 4;                          // chunks in the packed weights
    const int CB = ks * a.chunks_per_split;               // this workgroup's chunk range [CB, CB + C16)
    const int C16 = (C16T - CB) < a.chunks_per_split ? (C16T - CB) : a.chunks_per_split;

    // per-lane LDS offsets of the A (pixel) and B (channel) fragments
    int a_off[WM], b_off[WN];
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        const int m = (wave_m * WM + i) * 16 + r;
        const int py = m / a.TW, px = m - py * a.TW;
        // stride 2: a halo row is stored de-interleaved - its even columns first, then the odd ones (see h_src) - so that the 16
        // pixels of a fragment read are 16 CONSECUTIVE 64-byte entries, as for stride 1 (interleaved, their 128-byte stride put 71 %
        // of the LDS cycles of these layers into bank conflicts: 17.9 M of 25.1 M per launch, SQ_LDS_BANK_CONFLICT)
        a_off[i] = ((py * S) * a.WT + (S == 2 ? px : px * S)) * 64 + g * 16;
    }
    const int WE = (a.WT + 1) >> 1;                      // even columns of a halo row (stride 2)
#pragma unroll
    for (int j = 0; j < WN; ++j) b_off[j] = ((wave_n * WN + j) * 16 + r) * 64 + g * 16;

    // halo staging bookkeeping: element offset of this thread's float4 units inside the image
    const float* const xi = a.x + (size_t)img * a.H * a.W * a.cin_real;
    int h_src[HL];
#pragma unroll
    for (int k = 0; k < HL; ++k) {
        const int e = tid + k * NT;
        const int hp = e >> 2, q = e & 3;
        int off = -2;
        if (hp < halo_pix) {
            const int hy = hp / a.WT, sc = hp - hy * a.WT;
            const int hx = S == 2 ? (sc < WE ? 2 * sc : 2 * (sc - WE) + 1) : sc;      // LDS slot sc of the row holds halo column hx
            const int iy = gy0 * S + a.in_oy + hy, ix = gx0 * S + a.in_ox + hx;
            if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)
                off = a.ps_in ? ((2 * iy) * (2 * a.W) + 2 * ix) * (a.Cin >> 2) + q * 4 : (iy * a.W + ix) * a.cin_real + q * 4;
            else
                off = -1;
        }
        h_src[k] = off;
    }
    const float* const wn = a.wp + (size_t)n0 * 16;
    const size_t slab_stride = (size_t)a.Cout * 16;  // floats between consecutive (tap, chunk) slabs

    f32x4 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    f32x4 hreg[HL], wreg[WL];

    auto load_halo = [&](int c) {
        int coff = (CB + c) * 16;
        if (a.ps_in) {  // chunk c covers packed channels (2*si+sj)*C + cc0 .. +15
            const int C = a.Cin >> 2;
            const int sub = coff / C, cc0 = coff - sub * C;
            coff = ((sub >> 1) * (2 * a.W) + (sub & 1)) * C + cc0;
        }
        if (a.cin_real == 3) {  // RGB input: 3 floats per pixel, chunk 0 only, channels 3..15 are zero
#pragma unroll
            for (int k = 0; k < HL; ++k) {
                const bool ok = h_src[k] >= 0 && ((tid + k * NT) & 3) == 0;
                const int o = ok ? h_src[k] : 0;
                const float v0 = xi[o], v1 = xi[o + 1], v2 = xi[o + 2];
                hreg[k] = ok ? (f32x4){v0, v1, v2, 0.f} : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            return;
        }
#pragma unroll
        for (int k = 0; k < HL; ++k) {
            const int o = h_src[k] < 0 ? 0 : h_src[k];
            f32x4 v = *(const f32x4*)(xi + o + coff);
            hreg[k] = h_src[k] >= 0 ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    auto store_halo = [&](char* hb) {
#pragma unroll
        for (int k = 0; k < HL; ++k)
            if (h_src[k] != -2) *(f32x4*)(hb + (tid + k * NT) * 16) = hreg[k];
    };
    auto load_w = [&](int c, int tw) {
        const float* src = wn + ((size_t)tw * C16T + CB + c) * slab_stride;
#pragma unroll
        for (int k = 0; k < WL; ++k) {
            const int e = tid + k * NT;
            if (BN * 4 % NT == 0 || e < BN * 4) wreg[k] = *(const f32x4*)(src + e * 4);
        }
    };
    auto store_w = [&](char* wb) {
#pragma unroll
        for (int k = 0; k < WL; ++k) {
            const int e = tid + k * NT;
            if (BN * 4 % NT == 0 || e < BN * 4) *(f32x4*)(wb + e * 16) = wreg[k];
        }
    };

    // ---- software pipeline ---------------------------------------------------------------------
    // global -> registers -> LDS runs TWO slabs ahead, LDS -> fragment registers ONE slab ahead: the
    // ds_reads of slab s+1 are issued before the MFMA block of slab s and land under it, so after each
    // barrier the matrix pipe restarts at once on operands that are already in registers.
    // LDS-DMA variants of the two loaders: same lane -> byte mapping as the register path, no VGPR round trip
    auto dma_w = [&](int c, int tw, char* wb) {
        const float* src = wn + ((size_t)tw * C16T + CB + c) * slab_stride;
#pragma unroll
        for (int k = 0; k < WL; ++k) {
            const int e = tid + k * NT;
            if (BN * 4 % NT == 0 || e < BN * 4) lds_dma16(src + e * 4, wb + (k * NT + wave * 64) * 16);
        }
    };
    auto dma_halo = [&](int c, char* hb) {
        int coff = (CB + c) * 16;
        if (a.ps_in) {
            const int C = a.Cin >> 2;
            const int sub = coff / C, cc0 = coff - sub * C;
            coff = ((sub >> 1) * (2 * a.W) + (sub & 1)) * C + cc0;
        }
#pragma unroll
        for (int k = 0; k < HL; ++k) {
            if (h_src[k] != -2) {
                const float* src = h_src[k] >= 0 ? xi + h_src[k] + coff : g_zero16;
                lds_dma16(src, hb + (k * NT + wave * 64) * 16);
            }
        }
    };

    f32x4 fa0[WM], fb0[WN], fa1[WM], fb1[WN];   // two fragment sets, statically indexed (kept in VGPRs)

#define PESR_READ_FRAGS(FA, FB, C_, T_, WB_)                                                    \
    {                                                                                          \
        const char* const hb_ = ((C_) & 1) ? halo1 : halo0;                                    \
        const unsigned tc_ = tap_code(a, T_);                                                  \
        const int dx_ = (int)((tc_ >> 2) & 3u);                                                \
        const int toff_ = ((int)(tc_ & 3u) * a.WT + (S == 2 ? (dx_ == 1 ? WE : dx_ >> 1) : dx_)) * 64; \
        _Pragma("unroll") for (int i = 0; i < WM; ++i) FA[i] = *(const f32x4*)(hb_ + a_off[i] + toff_); \
        _Pragma("unroll") for (int j = 0; j < WN; ++j) FB[j] = *(const f32x4*)((WB_) + b_off[j]); \
    }
#define PESR_MFMA_BLOCK(FA, FB)                                                                 \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk)                                            \
        _Pragma("unroll") for (int i = 0; i < WM; ++i)                                          \
            _Pragma("unroll") for (int j = 0; j < WN; ++j)                                      \
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(FA[i][kk], FB[j][kk], acc[i][j], 0, 0, 0);

    const int nslab = C16 * a.ntaps;
    // advance a (chunk, tap) slab cursor; past the last slab it wraps to slab 0 (harmless re-reads, never used)
    auto adv = [&](int& c_, int& t_) { if (++t_ == a.ntaps) { t_ = 0; if (++c_ == C16) c_ = 0; } };

    if (MODE == 2) {
        // ---- main pipeline (>= 4 taps per chunk, >= 16-channel input) -------------------------------------------
        // Weight slabs travel global -> LDS by LDS-DMA into a 4-slot ring, three to four slabs ahead; halo chunks
        // likewise, one chunk ahead.  MFMA fragments are read from LDS one slab ahead into the second register set,
        // so the matrix pipe restarts right after a barrier on operands that are already in registers.  ONE barrier
        // per TWO slabs: when a double-step starts, the ring slots of slabs s-1 and s are free (their fragments were
        // read - and those reads retired by the barrier's lgkmcnt(0) - during the previous double-step) and receive
        // slabs s+3 and s+4; the barrier's vmcnt(0) retires the DMA before anyone reads them.
        char* const ring = wb0;
        int cd = 0, td = 0, sd = 0;                         // DMA cursor: next slab to fetch
        auto dma_next = [&]() { dma_w(cd, tap_code(a, td) >> 4, ring + (sd & 3) * (BN * 64)); adv(cd, td); ++sd; };
        dma_halo(0, halo0);
        dma_next(); dma_next(); dma_next();                 // slabs 0, 1, 2
        __syncthreads();
        int cr = 0, tr = 0, sr = 0;                         // fragment-read cursor
        PESR_READ_FRAGS(fa0, fb0, cr, tr, ring + (sr & 3) * (BN * 64))  adv(cr, tr); ++sr;      // slab 0
        __syncthreads();                                    // slot 0 may be refilled only after every wave has read it
        int c = 0, t = 0;                                   // cursor of the slab being multiplied
#pragma unroll 1
        for (int sl = 0; sl < nslab; sl += 2) {
            dma_next(); dma_next();                         // slabs sl+3, sl+4 -> slots of slabs sl-1, sl
            {   // a chunk that opens in this double-step triggers the DMA of the NEXT chunk's halo
                int c1 = c, t1 = t; adv(c1, t1);
                const int copen = (t == 0) ? c : ((t1 == 0 && sl + 1 < nslab) ? c1 : -1);
                if (copen >= 0 && copen + 1 < C16 && (t == 0 || c1 != 0)) dma_halo(copen + 1, ((copen + 1) & 1) ? halo1 : halo0);
            }
            PESR_READ_FRAGS(fa1, fb1, cr, tr, ring + (sr & 3) * (BN * 64))  adv(cr, tr); ++sr;  // slab sl+1
            PESR_MFMA_BLOCK(fa0, fb0)                                                           // slab sl
            adv(c, t);
            if (sl + 1 < nslab) {
                PESR_READ_FRAGS(fa0, fb0, cr, tr, ring + (sr & 3) * (BN * 64))  adv(cr, tr); ++sr;  // slab sl+2
                PESR_MFMA_BLOCK(fa1, fb1)                                                           // slab sl+1
                adv(c, t);
            }
            __syncthreads();
        }
    } else {
        // ---- fallback pipeline: register-staged, one barrier per slab -------------------------------------------
        // MODE 1: fragments prefetched one slab ahead (2-3 taps per chunk, or the 3-channel RGB input);
        // MODE 0: one tap per chunk (a parity class of the stride-2 dgrad): the next chunk's halo is staged in the
        //         same iteration that would prefetch from it, so fragments are read in-iteration.
        // Weight buffer of slab s: MODE 1 alternates two (slab s + 2 overwrites slab s, whose fragments every wave read one barrier
        // earlier); MODE 0 reads slab s IN the iteration that stages slab s + 2, so it rotates THREE buffers - with two, a wave that
        // finishes its MFMA block early overwrote slab s under a wave that had not read it yet.  (Round 4: a real bug of rounds 1 - 3
        // for the one configuration whose staging crosses waves - 256-channel workgroups, where waves 0 - 3 store the rows waves 4 - 5
        // read: the one-tap parity class of a stride-2 input gradient with >= 192 such tiles, e.g. dx [3,192,192,256], came out
        // wrong by O(1) in channels 128 - 191.  No layer of the benchmarked networks takes that configuration.)
        auto wslot = [&](int s_) -> char* { return wb0 + (MODE == 0 ? s_ % 3 : (s_ & 1)) * (BN * 64); };
        load_halo(0);
        load_w(0, tap_code(a, 0) >> 4);
        store_halo(halo0);
        store_w(wslot(0));
        if (nslab > 1) {
            int c1 = 0, t1 = 0; adv(c1, t1);
            load_w(c1, tap_code(a, t1) >> 4);
            store_w(wslot(1));
        }
        __syncthreads();
        if (MODE == 1) PESR_READ_FRAGS(fa0, fb0, 0, 0, wslot(0))
        int c = 0, t = 0;
#define PESR_STEP(CA, CB_, NA, NB, SL)                                                          \
        {                                                                                      \
            int c1 = c, t1 = t; adv(c1, t1);                                                   \
            int c2 = c1, t2 = t1; adv(c2, t2);                                                 \
            const bool halo_now = (t == 0) && (c + 1 < C16);                                   \
            load_w(c2, tap_code(a, t2) >> 4);                                                  \
            if (halo_now) load_halo(c + 1);                                                    \
            if (MODE == 1) PESR_READ_FRAGS(NA, NB, c1, t1, wslot((SL) + 1))                    \
            else PESR_READ_FRAGS(CA, CB_, c, t, wslot(SL))                                     \
            PESR_MFMA_BLOCK(CA, CB_)                                                           \
            /* keep the staging ds_writes (and their vmcnt waits) BEHIND the MFMA block: hipcc otherwise hoists   \
               them to its top and exposes the global-load latency once per slab */            \
            __builtin_amdgcn_sched_barrier(0);                                                 \
            store_w(wslot((SL) + 2)); /* MODE 1: slab SL's buffer; MODE 0: the third one */       \
            if (halo_now) store_halo((c & 1) ? halo0 : halo1);                                 \
            __syncthreads();                                                                   \
            c = c1; t = t1;                                                                    \
        }
#pragma unroll 1
        for (int sl = 0; sl < nslab; sl += 2) {
            PESR_STEP(fa0, fb0, fa1, fb1, sl)
            if (sl + 1 < nslab) PESR_STEP(fa1, fb1, fa0, fb0, sl + 1)
        }
#undef PESR_STEP
    }
#undef PESR_MFMA_BLOCK
#undef PESR_READ_FRAGS

    // ---- epilogue ----------------------------------------------------------------------------------
    // D layout of a 16x16 tile: col = lane&15 (channel), row = (lane>>4)*4 + reg (pixel): written straight to
    // memory that is 64-byte fragments per store.  Instead the accumulators go through LDS (the staging buffers
    // are free now) and leave as whole 16-byte-per-lane, pixel-contiguous rows: 4x fewer, fully coalesced stores
    // (and skip / mask loads); the exposed tail of the kernel drops from 24 us to ~10 us on the G-body shape.
    const size_t img_out = (size_t)img * a.OH * a.OW;
    if (a.cout_store % 4 == 0) {
        constexpr int MT = WAVES_M * WM * 16;
        constexpr int C4 = BN / 4;                  // float4 columns of the tile
        constexpr int RS = BN * 4 + 16;             // padded row stride: the 4 pixel rows of a store hit disjoint banks
        constexpr int EIT = (MT * C4 + NT - 1) / NT;   // output float4s per thread
        char* const ob = smem;
        // BatchNorm sums (NT % C4 == 0: a thread keeps its four channels over all its pixels)
        const BnEpi* const bn = pesr_bn_epi(bn_kernarg_off);
        const int bn_mode = a.ksplit == 1 ? bn->mode : 0;
        const bool bn_on = bn_mode != 0;
        const float bn_slope = bn->slope;
        f32x4 st1 = {0.f, 0.f, 0.f, 0.f}, st2 = {0.f, 0.f, 0.f, 0.f};
        f32x4 bmu = st1, bis = st1, bga = st1, bbe = st1;
        // output element u = tid + k * NT of the tile -> (in range, element offset of its four channels)
        auto out_index = [&](int u, size_t* idx) -> bool {
            const int m = u / C4, c4 = u - m * C4;
            const int co = n0 + c4 * 4;
            const int py = m / a.TW, px = m - py * a.TW;
            const int gy = gy0 + py, gx = gx0 + px;
            if (u >= MT * C4 || gy >= a.GH || gx >= a.GW || co >= a.cout_store) return false;
            const int oy = gy * a.out_my + a.out_ay, ox = gx * a.out_mx + a.out_ax;
            if (a.ps) {   // packed channel co = (2*si+sj)*C + c  ->  out[n][2*oy+si][2*ox+sj][c]
                const int C = a.Cout >> 2;
                const int sub = co / C, cc = co - sub * C;
                *idx = (((size_t)img * (2 * a.OH) + 2 * oy + (sub >> 1)) * (2 * a.OW) + 2 * ox + (sub & 1)) * C + cc;
            } else {
                *idx = (img_out + (size_t)oy * a.OW + ox) * a.cout_store + co;
            }
            return true;
        };
        f32x4 zpre[EIT];
        if (bn_mode == 2) {
            const int cq = n0 + (tid % C4) * 4;
            if (cq < a.cout_store) {
                bmu = *(const f32x4*)(bn->mi + cq); bis = *(const f32x4*)(bn->mi + a.cout_store + cq);
                bga = *(const f32x4*)(bn->gamma + cq); bbe = *(const f32x4*)(bn->beta + cq);
            }
        }
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int m = (wave_m * WM + i) * 16 + g * 4 + jj;
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    *(float*)(ob + m * RS + ((wave_n * WN + j) * 16 + r) * 4) = acc[i][j][jj];
            }
        __syncthreads();
        const size_t slab_off = (size_t)ks * ((size_t)a.N * a.OH * a.OW * a.cout_store);
        if (bn_mode == 2) {
            // (no bias / mask / skip / activation here: the launcher refuses them with this mode)
            // mode 2 reads z at every output element: ALL of a thread's loads are issued here, behind the staging barrier, and the store
            // loop consumes them in order.  Measured on the 16 x 192 x 192 x 64 input gradient (157 us plain): loads inside the store
            // loop 218 us (a chain of EIT dependent HBM latencies per thread), all issued in FRONT of the staging barrier 216 (the
            // barrier's vmcnt(0) waits for them with nothing to overlap), all issued here 196 (scripts/bn_fuse_time.py)
            {
                const float* const bn_z = bn->z;
#pragma unroll
                for (int k = 0; k < EIT; ++k) {
                    size_t idx;
                    zpre[k] = out_index(tid + k * NT, &idx) ? *(const f32x4*)(bn_z + idx) : (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
#pragma unroll
            for (int k = 0; k < EIT; ++k) {
                const int u = tid + k * NT;
                size_t idx;
                if (!out_index(u, &idx)) continue;
                const int m = u / C4, c4 = u - m * C4;
                f32x4 v = *(const f32x4*)(ob + m * RS + c4 * 16) * a.alpha;
                const f32x4 xh = (zpre[k] - bmu) * bis;
                const f32x4 zz = bga * xh + bbe;
                v.x = zz.x > 0.f ? v.x : v.x * bn_slope; v.y = zz.y > 0.f ? v.y : v.y * bn_slope;
                v.z = zz.z > 0.f ? v.z : v.z * bn_slope; v.w = zz.w > 0.f ? v.w : v.w * bn_slope;
                st1 += v; st2 += v * xh;
                *(f32x4*)(a.y + idx) = v;
            }
        } else {
            for (int u = tid; u < MT * C4; u += NT) {
                size_t idx;
                if (!out_index(u, &idx)) continue;
                const int m = u / C4, c4 = u - m * C4;
                const int co = n0 + c4 * 4;
                f32x4 v = *(const f32x4*)(ob + m * RS + c4 * 16);
                if (a.ksplit > 1) {   // raw partial sums; the finish kernel applies the epilogue
                    *(f32x4*)(a.slab + slab_off + idx) = v;
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
                if (bn_on) { st1 += v; st2 += v * v; }
                *(f32x4*)(a.y + idx) = v;
            }
        }
        if (bn_on) {
            __syncthreads();                                 // every thread is done with the accumulator tile in `ob`
            f32x4* const red = (f32x4*)smem;                 // [2][NT]
            red[tid] = st1; red[NT + tid] = st2;
            __syncthreads();
            if (tid < C4 && n0 + tid * 4 < a.cout_store) {
                f64x4 d1 = {0.0, 0.0, 0.0, 0.0}, d2 = {0.0, 0.0, 0.0, 0.0};
                for (int k = 0; k < NT / C4; ++k) {
                    d1 += __builtin_convertvector(red[k * C4 + tid], f64x4);
                    d2 += __builtin_convertvector(red[NT + k * C4 + tid], f64x4);
                }
                const int row = a.bn_row0 + (img * a.tiles_y + ty) * a.tiles_x + tx;
                float* const pr = bn->part + (size_t)row * 2 * a.cout_store + n0 + tid * 4;
                *(f32x4*)pr = __builtin_convertvector(d1, f32x4);
                *(f32x4*)(pr + a.cout_store) = __builtin_convertvector(d2, f32x4);
            }
        }
        return;
    }
    // scalar fallback: output channel count not a multiple of 4 (the C -> 3 layers, zero-padded to 64)
    float bias_r[WN];
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int co = n0 + (wave_n * WN + j) * 16 + r;
        bias_r[j] = (a.bias && co < a.cout_store) ? a.bias[co] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < WM; ++i) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int m = (wave_m * WM + i) * 16 + g * 4 + jj;
            const int py = m / a.TW, px = m - py * a.TW;
            const int gy = gy0 + py, gx = gx0 + px;
            if (gy >= a.GH || gx >= a.GW) continue;
            const int oy = gy * a.out_my + a.out_ay, ox = gx * a.out_mx + a.out_ax;
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const int co = n0 + (wave_n * WN + j) * 16 + r;
                if (co >= a.cout_store) continue;
                float v = acc[i][j][jj];
                v += bias_r[j];
                v *= a.alpha;
                size_t idx;
                if (a.ksplit > 1) {   // raw partial sum; bias / scale / mask / skip / activation happen in the finish kernel
                    idx = (img_out + (size_t)oy * a.OW + ox) * a.cout_store + co;
                    a.slab[(size_t)ks * ((size_t)a.N * a.OH * a.OW * a.cout_store) + idx] = acc[i][j][jj];
                    continue;
                }
                if (a.ps) {
                    // packed channel co = (2*si+sj)*C + c  ->  out[n][2*oy+si][2*ox+sj][c]
                    const int C = a.Cout >> 2;
                    const int sub = co / C, cc = co - sub * C;
                    idx = (((size_t)img * (2 * a.OH) + 2 * oy + (sub >> 1)) * (2 * a.OW) + 2 * ox + (sub & 1)) * C + cc;
                } else {
                    idx = (img_out + (size_t)oy * a.OW + ox) * a.cout_store + co;
                }
                if (a.mask) v = a.mask[idx] > 0.f ? v : 0.f;
                if (a.skip) v += a.skip[idx];
                if (a.act == PESR_ACT_RELU) v = v > 0.f ? v : 0.f;
                else if (a.act == PESR_ACT_LRELU) v = v > 0.f ? v : v * a.slope;
                a.y[idx] = v;
            }
        }
    }
}

template <int WAVES_M, int WAVES_N, int WM, int WN, int S, int HL, int MODE>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void conv3x3_mfma_kernel(const ConvArgs a, const BnEpi bn) {
    (void)bn;       // read through pesr_bn_epi() in the epilogue
    conv3x3_mfma_body<WAVES_M, WAVES_N, WM, WN, S, HL, MODE>(a, blockIdx.x, gridDim.x, (unsigned)((sizeof(ConvArgs) + 7) & ~(size_t)7));
}

// Input gradient of a stride-2 conv: its four output parity classes (1 / 2 / 2 / 4 taps, conv3x3_mfma.hip's host side) as ONE
// launch, blockIdx.y = class with the four-tap class first (round 4).  As four launches each class was a small grid with its own
// ramp, tail and launch boundary (the 24 x 24 <- 12 x 12 x 512 layer: 196 us for 10.9 GFLOP); one grid lets the classes fill each
// other's tails.  Every class keeps its own tap table, tile shape and pipeline MODE (4 taps: LDS-DMA ring; 2 taps: prefetched
// fragments; 1 tap: in-iteration), i.e. the same instruction stream per class as before: same bits.
struct ConvArgs4 {
    ConvArgs c[4];          // order: (py, px) = (1,1), (1,0), (0,1), (0,0)
    int tiles[4];
};
template <int WAVES_M, int WAVES_N, int WM, int WN, int HL>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void conv3x3_s2dgrad4_kernel(const ConvArgs4 a4, const BnEpi bn) {
    (void)bn;
    constexpr unsigned BO = (unsigned)((sizeof(ConvArgs4) + 7) & ~(size_t)7);
    const int bx = blockIdx.x;
    switch (blockIdx.y) {
        case 0: if (bx < a4.tiles[0]) conv3x3_mfma_body<WAVES_M, WAVES_N, WM, WN, 1, HL, 2>(a4.c[0], bx, a4.tiles[0], BO); break;
        case 1: if (bx < a4.tiles[1]) conv3x3_mfma_body<WAVES_M, WAVES_N, WM, WN, 1, HL, 1>(a4.c[1], bx, a4.tiles[1], BO); break;
        case 2: if (bx < a4.tiles[2]) conv3x3_mfma_body<WAVES_M, WAVES_N, WM, WN, 1, HL, 1>(a4.c[2], bx, a4.tiles[2], BO); break;
        default: if (bx < a4.tiles[3]) conv3x3_mfma_body<WAVES_M, WAVES_N, WM, WN, 1, HL, 0>(a4.c[3], bx, a4.tiles[3], BO); break;
    }
}

// y = act(alpha * (sum_ks slab[ks] + bias) [masked] + skip): fixed-order sum of the split-K partials + the epilogue
__global__ void conv_splitk_finish_kernel(const float* __restrict__ slab, const float* __restrict__ bias, const float* __restrict__ skip,
                                          const float* __restrict__ mask, float* __restrict__ y, long total, int C, int ksplit,
                                          float alpha, int act, float slope) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        float v = slab[e];
        for (int k = 1; k < ksplit; ++k) v += slab[(size_t)k * total + e];
        if (bias) v += bias[e % C];
        v *= alpha;
        if (mask) v = mask[e] > 0.f ? v : 0.f;
        if (skip) v += skip[e];
        if (act == PESR_ACT_RELU) v = v > 0.f ? v : 0.f;
        else if (act == PESR_ACT_LRELU) v = v > 0.f ? v : v * slope;
        y[e] = v;
    }
}
// the same on 4 consecutive channels (C % 4 == 0): 16-byte loads, all ksplit partials of an element in flight together
__global__ void conv_splitk_finish4_kernel(const f32x4* __restrict__ slab, const float* __restrict__ bias, const f32x4* __restrict__ skip,
                                           const f32x4* __restrict__ mask, f32x4* __restrict__ y, long total4, int C, int ksplit,
                                           float alpha, int act, float slope) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total4; e += (long)gridDim.x * blockDim.x) {
        f32x4 p[8];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k < ksplit) p[k] = slab[(size_t)k * total4 + e];
        f32x4 v = p[0];
#pragma unroll
        for (int k = 1; k < 8; ++k)
            if (k < ksplit) v += p[k];
        if (bias) v += *(const f32x4*)(bias + (e * 4) % C);
        v *= alpha;
        if (mask) {
            const f32x4 mk = mask[e];
            v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f; v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
        }
        if (skip) v += skip[e];
        if (act == PESR_ACT_RELU) {
            v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
        } else if (act == PESR_ACT_LRELU) {
            v.x = v.x > 0.f ? v.x : v.x * slope; v.y = v.y > 0.f ? v.y : v.y * slope;
            v.z = v.z > 0.f ? v.z : v.z * slope; v.w = v.w > 0.f ? v.w : v.w * slope;
        }
        y[e] = v;
    }
}

// The split-K finish WITH the BatchNorm sums of common.h BnEpi (round 6): the layers whose conv kernel cannot leave them because its
// workgroups hold partial sums only (the Discriminator's features.6 / .7 forward, the input gradients of features.4 / .6).  PESR_BN_FINISH_ROWS
// workgroups walk the tensor with a stride that is a multiple of the channel groups, so a thread keeps its four channels; one row of
// [2][C] per workgroup, fp32 per thread, double across the threads of a workgroup in a fixed order.  (No bias / mask / skip / activation
// in mode 2; mode 1 sums what it stores.)
constexpr int PESR_BN_FINISH_ROWS = 256;
template <int MODE>
__global__ __launch_bounds__(256) void conv_splitk_finish4_bn_kernel(const f32x4* __restrict__ slab, const float* __restrict__ bias, f32x4* __restrict__ y,
                                                                      long total4, int C, int ksplit, float alpha, const BnEpi bn) {
    const int C4 = C >> 2;
    const int c4 = (int)(((long)blockIdx.x * 256 + threadIdx.x) % C4);            // (gridDim.x * 256) % C4 == 0: fixed over the walk
    f32x4 st1 = {0.f, 0.f, 0.f, 0.f}, st2 = st1, bmu = st1, bis = st1, bga = st1, bbe = st1, b4 = st1;
    if (MODE == 2) {
        bmu = *(const f32x4*)(bn.mi + c4 * 4); bis = *(const f32x4*)(bn.mi + C + c4 * 4);
        bga = *(const f32x4*)(bn.gamma + c4 * 4); bbe = *(const f32x4*)(bn.beta + c4 * 4);
    }
    if (bias) b4 = *(const f32x4*)(bias + c4 * 4);
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total4; e += (long)gridDim.x * 256) {
        f32x4 p[8];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k < ksplit) p[k] = slab[(size_t)k * total4 + e];
        f32x4 v = p[0];
#pragma unroll
        for (int k = 1; k < 8; ++k)
            if (k < ksplit) v += p[k];
        v = (v + b4) * alpha;
        if (MODE == 2) {
            const f32x4 xh = (((const f32x4*)bn.z)[e] - bmu) * bis;
            const f32x4 zz = bga * xh + bbe;
            v.x = zz.x > 0.f ? v.x : v.x * bn.slope; v.y = zz.y > 0.f ? v.y : v.y * bn.slope;
            v.z = zz.z > 0.f ? v.z : v.z * bn.slope; v.w = zz.w > 0.f ? v.w : v.w * bn.slope;
            st1 += v; st2 += v * xh;
        } else {
            st1 += v; st2 += v * v;
        }
        y[e] = v;
    }
    __shared__ f32x4 red[2][256];
    red[0][threadIdx.x] = st1; red[1][threadIdx.x] = st2;
    __syncthreads();
    // threads t, t + C4', ... hold the same channels (C4' = the channel groups one workgroup covers: min(C4, 256))
    const int cw = C4 < 256 ? C4 : 256;
    if (threadIdx.x < cw) {
        f64x4 d1 = {0.0, 0.0, 0.0, 0.0}, d2 = {0.0, 0.0, 0.0, 0.0};
        for (int k = threadIdx.x; k < 256; k += cw) {
            d1 += __builtin_convertvector(red[0][k], f64x4);
            d2 += __builtin_convertvector(red[1][k], f64x4);
        }
        float* const pr = bn.part + (size_t)blockIdx.x * 2 * C + c4 * 4;
        *(f32x4*)pr = __builtin_convertvector(d1, f32x4);
        *(f32x4*)(pr + C) = __builtin_convertvector(d2, f32x4);
    }
}

// rows the finish kernel above leaves (0: the shape is not covered): one per workgroup
long pesr_conv_splitk_finish_bn_rows(int C, int ksplit) {
    const int C4 = C >> 2;
    if (C % 4 || ksplit > 8 || C4 < 1 || (C4 < 256 ? 256 % C4 : C4 % 256)) return 0;
    return PESR_BN_FINISH_ROWS;
}

int pesr_conv_splitk_finish_bn_launch(const float* slab, const float* bias, float* y, long total, int C, int ksplit, float alpha, const BnEpi& bn,
                                      hipStream_t stream) {
    if (!pesr_conv_splitk_finish_bn_rows(C, ksplit) || !bn.part) return PESR_EINVAL;
    const long total4 = total / 4;
    if (bn.mode == 2)
        hipLaunchKernelGGL(conv_splitk_finish4_bn_kernel<2>, dim3(PESR_BN_FINISH_ROWS), dim3(256), 0, stream, (const f32x4*)slab, bias, (f32x4*)y, total4, C, ksplit,
                           alpha, bn);
    else
        hipLaunchKernelGGL(conv_splitk_finish4_bn_kernel<1>, dim3(PESR_BN_FINISH_ROWS), dim3(256), 0, stream, (const f32x4*)slab, bias, (f32x4*)y, total4, C, ksplit,
                           alpha, bn);
    return pesr_launch_status();
}

int pesr_conv_splitk_finish_launch(const float* slab, const float* bias, const float* skip, const float* mask, float* y, long total,
                                   int C, int ksplit, float alpha, int act, float slope, hipStream_t stream) {
    if (C % 4 == 0 && ksplit <= 8) {
        const long total4 = total / 4;
        const int fgrid = (int)((total4 + 255) / 256 < 4096 ? (total4 + 255) / 256 : 4096);
        hipLaunchKernelGGL(conv_splitk_finish4_kernel, dim3(fgrid), dim3(256), 0, stream, (const f32x4*)slab, bias, (const f32x4*)skip,
                           (const f32x4*)mask, (f32x4*)y, total4, C, ksplit, alpha, act, slope);
        return pesr_launch_status();
    }
    const int fgrid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(conv_splitk_finish_kernel, dim3(fgrid), dim3(256), 0, stream, slab, bias, skip, mask, y, total, C, ksplit, alpha,
                       act, slope);
    return pesr_launch_status();
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
namespace {

// the BnEpi of the call in progress on this thread (set by set_bn at the top of the two entry points, read by the launchers below:
// it rides beside ConvArgs through the dispatch templates without widening every signature)
static thread_local BnEpi g_bn_epi;

static void set_tap(ConvArgs& a, int t, int dy, int dx, int w) {
    const unsigned long long code = (unsigned)(dy | (dx << 2) | (w << 4));
    if (t < 8) a.tap_lo |= code << (8 * t); else a.tap_hi = (unsigned)code;
}

struct TileChoice { int TH, TW; };

// pick (TH, TW) with TH*TW == MT minimising wasted (out-of-image) tile area, subject to the halo
// fitting the per-thread staging registers (HL float4 per thread) and the LDS budget.
static bool choose_tile(int MT, int S, int hext, int wext, int GH, int GW, int max_halo_pix, TileChoice* out) {
    long best = -1;
    for (int TW = 1; TW <= MT; ++TW) {
        if (MT % TW) continue;
        const int TH = MT / TW;
        const int HTl = (TH - 1) * S + hext, WTl = (TW - 1) * S + wext;
        if ((long)HTl * WTl > max_halo_pix) continue;
        const long cover = (long)pesr_cdiv(GH, TH) * TH * pesr_cdiv(GW, TW) * TW;
        // prefer less waste, then the smaller halo
        const long score = cover * 4096 + (long)HTl * WTl;
        if (best < 0 || score < best) { best = score; out->TH = TH; out->TW = TW; }
    }
    return best >= 0;
}

// Fill in the tile shape, halo extent, split-K decision of one problem for a kernel configuration; -> LDS bytes, pipeline mode.
// target_wgs: the workgroup count the split-K rule aims at (256 = one per CU; 768 for the 48-pixel tiles, three per CU)
template <int WAVES_M, int WAVES_N, int WM, int WN, int S, int HL>
static int prep_cfg(ConvArgs& a, int hext, int wext, int target_wgs, size_t* lds_out, int* mode_out, long* grid_out) {
    constexpr int NT = WAVES_M * WAVES_N * 64;
    constexpr int MT = WAVES_M * WM * 16;
    constexpr int BN = WAVES_N * WN * 16;
    if (a.Cout % BN) return PESR_EINVAL;
    TileChoice tc;
    const int max_halo = (HL * NT) / 4;
    if (!choose_tile(MT, S, hext, wext, a.GH, a.GW, max_halo, &tc)) return PESR_EINVAL;
    a.TH = tc.TH; a.TW = tc.TW;
    a.HT = (a.TH - 1) * S + hext; a.WT = (a.TW - 1) * S + wext;
    a.tiles_y = pesr_cdiv(a.GH, a.TH); a.tiles_x = pesr_cdiv(a.GW, a.TW);
    a.n_tiles = a.Cout / BN;
    const int halo_bytes = ((a.HT * a.WT * 64 + 255) / 256) * 256;
    // MODE 2 (LDS-DMA, 4-slot weight ring, one barrier per two slabs) needs >= 4 taps per chunk so that a chunk's halo
    // is resident a full double-step before its first fragment read, and a >= 16-channel input (16-byte DMA pieces)
    const int mode = (a.ntaps >= 4 && a.cin_real != 3) ? 2 : (a.ntaps > 1 ? 1 : 0);
    size_t lds = 2 * (size_t)halo_bytes + (mode == 2 ? 4 : (mode == 0 ? 3 : 2)) * (size_t)BN * 64;
    const size_t lds_acc = (size_t)MT * (BN * 4 + 16);     // accumulator tile staged for the coalesced epilogue
    if (lds_acc > lds) lds = lds_acc;
    if (lds > 160 * 1024) return PESR_EINVAL;
    const long tiles = (long)a.N * a.tiles_y * a.tiles_x * a.n_tiles;
    // split-K over the Cin chunks when the tiles alone cannot fill the 256 CUs (12x12 / 24x24 512-channel layers)
    const int C16T = a.Cin / 16;
    a.ksplit = 1; a.chunks_per_split = C16T;
    const size_t out_bytes = (size_t)a.N * a.OH * a.OW * a.cout_store * sizeof(float);
    if (a.slab && tiles < (long)target_wgs * 5 / 8 && a.out_my == 1 && a.out_mx == 1 && !a.ps && C16T >= 8) {
        int want = (int)((target_wgs + tiles - 1) / tiles);
        if (want > 8) want = 8;
        if (want > C16T / 4) want = C16T / 4;
        while (want > 1 && (size_t)want * out_bytes > a.slab_bytes) --want;
        if (want > 1) {
            a.chunks_per_split = (C16T + want - 1) / want;
            a.ksplit = (C16T + a.chunks_per_split - 1) / a.chunks_per_split;
        }
    }
    *lds_out = lds; *mode_out = mode; *grid_out = tiles * a.ksplit;
    return PESR_OK;
}

template <int WAVES_M, int WAVES_N, int WM, int WN, int S, int HL>
static int launch_cfg(ConvArgs& a, int hext, int wext, hipStream_t stream, int target_wgs = 256) {
    constexpr int NT = WAVES_M * WAVES_N * 64;
    size_t lds; int mode; long grid;
    const int rc = prep_cfg<WAVES_M, WAVES_N, WM, WN, S, HL>(a, hext, wext, target_wgs, &lds, &mode, &grid);
    if (rc) return rc;
    // rows of BatchNorm partial sums the epilogue can leave: one per pixel tile (not with split-K: the finish kernel sums slabs)
    // (with split-K: from the finish kernel, which sums the slabs - one row per finish workgroup)
    a.bn_rows = (a.cout_store % 4 == 0 && NT % (WAVES_N * WN * 4) == 0 && !a.ps)
                    ? (a.ksplit == 1 ? (long)a.N * a.tiles_y * a.tiles_x : pesr_conv_splitk_finish_bn_rows(a.cout_store, a.ksplit)) : 0;
    if (a.dry) return PESR_OK;
    if (a.bn_mode && (a.bn_rows == 0 || a.bn_row0 + a.bn_rows > a.bn_cap || !g_bn_epi.part)) return PESR_EINVAL;
    const size_t out_bytes = (size_t)a.N * a.OH * a.OW * a.cout_store * sizeof(float);
    const float* bias = a.bias; const float* skip = a.skip; const float* mask = a.mask;
    BnEpi a_bn = g_bn_epi;
    if (a.ksplit > 1) a_bn.mode = 0;              // the conv kernel stores raw partial sums; the finish kernel does the BatchNorm part
#define PESR_LAUNCH_MODE(M_)                                                                              \
    {                                                                                                      \
        auto kern = conv3x3_mfma_kernel<WAVES_M, WAVES_N, WM, WN, S, HL, M_>;                              \
        static PesrDeviceOnce attr_once;                                        \
        attr_once([&] {                                                           \
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        });                                                                                                  \
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NT), lds, stream, a, a_bn);                    \
    }
    if (mode == 2) PESR_LAUNCH_MODE(2) else if (mode == 1) PESR_LAUNCH_MODE(1) else PESR_LAUNCH_MODE(0)
#undef PESR_LAUNCH_MODE
    if (a.ksplit > 1 && g_bn_epi.mode) {
        if (skip || mask || a.act != PESR_ACT_NONE) return PESR_EINVAL;
        return pesr_conv_splitk_finish_bn_launch((const float*)a.slab, bias, a.y, (long)(out_bytes / sizeof(float)), a.cout_store, a.ksplit, a.alpha,
                                                 g_bn_epi, stream);
    }
    if (a.ksplit > 1)
        return pesr_conv_splitk_finish_launch((const float*)a.slab, bias, skip, mask, a.y, (long)(out_bytes / sizeof(float)), a.cout_store,
                                              a.ksplit, a.alpha, a.act, a.slope, stream);
    return pesr_launch_status();
}

// The four parity classes of a stride-2 input gradient (cls[0..3] = (1,1), (1,0), (0,1), (0,0)) as one launch of one configuration.
template <int WAVES_M, int WAVES_N, int WM, int WN, int HL>
static int launch_s2dgrad4(ConvArgs4& a4, const int* hext, const int* wext, hipStream_t stream) {
    constexpr int NT = WAVES_M * WAVES_N * 64;
    size_t lds = 0; long gmax = 0;
    long rows = 0;
    const bool rows_ok = a4.c[0].cout_store % 4 == 0 && NT % (WAVES_N * WN * 4) == 0;
    for (int k = 0; k < 4; ++k) {
        a4.tiles[k] = 0;
        a4.c[k].bn_row0 = (int)rows;                     // the four classes' pixel tiles one after the other
        if (a4.c[k].GH <= 0 || a4.c[k].GW <= 0) continue;
        size_t l; int mode; long grid;
        const int rc = prep_cfg<WAVES_M, WAVES_N, WM, WN, 1, HL>(a4.c[k], hext[k], wext[k], 256, &l, &mode, &grid);
        if (rc) return rc;
        if (mode != (k == 0 ? 2 : (k == 3 ? 0 : 1)) || a4.c[k].ksplit != 1) return PESR_EINVAL;
        a4.tiles[k] = (int)grid;
        rows += (long)a4.c[k].N * a4.c[k].tiles_y * a4.c[k].tiles_x;
        if (l > lds) lds = l;
        if (grid > gmax) gmax = grid;
    }
    a4.c[0].bn_rows = rows_ok ? rows : 0;
    if (a4.c[0].dry) return PESR_OK;
    if (a4.c[0].bn_mode && (!rows_ok || rows > a4.c[0].bn_cap || !g_bn_epi.part)) return PESR_EINVAL;
    if (gmax == 0) return PESR_OK;
    auto kern = conv3x3_s2dgrad4_kernel<WAVES_M, WAVES_N, WM, WN, HL>;
    static PesrDeviceOnce attr_once;
    attr_once([&] { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    hipLaunchKernelGGL(kern, dim3((unsigned)gmax, 4), dim3(NT), lds, stream, a4, g_bn_epi);
    return pesr_launch_status();
}

// choose the tile configuration from the channel count and the amount of parallel work
template <int S>
static int dispatch(ConvArgs& a, int hext, int wext, hipStream_t stream) {
    const long M = (long)a.N * a.GH * a.GW;
    // stride-2 conv to 64 channels (D features.1): the 144-pixel tile's halo (2x2 input pixels per output) leaves room for one
    // 4-wave workgroup per CU only; 64-pixel tiles fit three (141 vs 185 us on 16x192x192x64)
    if (S == 2 && a.ntaps == 9 && a.Cout == 64 && M >= 64 * 512) return launch_cfg<1, 4, 4, 1, S, 5>(a, hext, wext, stream);
    // Layers whose 144-pixel x 64-channel tiles give at most ~one 4-wave workgroup per CU (D features.5 / .7: 48^2 -> 24^2 x 256,
    // 24^2 -> 12^2 x 512, and the parity classes of the small stride-2 input gradients): 48-pixel tiles instead - three workgroups
    // = 12 waves per CU, split-K towards 768 workgroups where a slab is available (round 3: 148 -> 130 us and 151 -> 134 us on
    // the two forward layers, 161 -> 147 us and 287 -> 199 us on their input gradients, same bits; profiles/r03_stride2_tiles.txt)
    if (a.Cout % 64 == 0 && a.Cout >= 128 && (long)pesr_cdiv(M, 144) * (a.Cout / 64) <= 320)
        return launch_cfg<1, 4, 3, 1, S, (S == 1 ? 2 : 4)>(a, hext, wext, stream, 768);
    if (a.Cout % 256 == 0) {
        // enough tiles to fill 256 CUs with the big tile?
        const long tiles_big = (M / 144) * (a.Cout / 256);
        if (tiles_big >= 192) return launch_cfg<1, 8, 9, 2, S, (S == 1 ? 2 : 6)>(a, hext, wext, stream);
    }
    if (a.Cout % 128 == 0) {
        const long tiles = (M / 144) * (a.Cout / 128);
        if (tiles >= 192) return launch_cfg<1, 8, 9, 1, S, (S == 1 ? 2 : 6)>(a, hext, wext, stream);
    }
    if (a.Cout % 64 == 0) return launch_cfg<1, 4, 9, 1, S, (S == 1 ? 4 : 11)>(a, hext, wext, stream);
    // <= 16 output channels (the C -> 3 layers, zero-padded to 16): 256 pixels x 16 channels per workgroup, the four
    // waves split the pixels - 4x less padding work than the 64-channel tile
    if (a.Cout == 16 && S == 1) return launch_cfg<4, 1, 4, 1, 1, 6>(a, hext, wext, stream);
    return PESR_EINVAL;
}

}  // namespace

// Plain conv (forward).  Also serves the stride-1 dgrad when given dgrad-packed weights
// (pack.hip mode 1: Cin/Cout swapped) and flip=1 (tap t reads weight tap 8-t).
static void set_bn(ConvArgs& a, PesrBnFuseArgs* f) {
    a.bn_mode = 0; a.dry = 0; a.bn_rows = 0; a.bn_row0 = 0; a.bn_cap = 0;
    g_bn_epi = BnEpi{};
    if (!f) return;
    a.dry = f->dry;
    if (f->dry || !f->mode) return;
    a.bn_mode = f->mode; a.bn_cap = f->rows;
    g_bn_epi.mode = f->mode; g_bn_epi.slope = f->slope; g_bn_epi.part = f->part;
    g_bn_epi.z = f->z; g_bn_epi.mi = f->mean_invstd; g_bn_epi.gamma = f->gamma; g_bn_epi.beta = f->beta;
}

int pesr_conv3x3_launch(const float* x, const float* wp, const float* bias, const float* skip, const float* mask,
                        float* y, int N, int H, int W, int Cin, int Cout, int stride, float alpha, int act,
                        float slope, int ps, int ps_in, int flip, int cin_real, int cout_store, void* ws, size_t ws_bytes,
                        hipStream_t stream, PesrBnFuseArgs* fuse) {
    // cin_real / cout_store: physical channel counts of x / y (0 = same as Cin / Cout).  The RGB layers
    // (3 -> N, N -> 3) run here zero-padded to Cin = 16 / Cout = 64 with 3-channel tensors in memory.
    if (cin_real == 0) cin_real = Cin;
    if (cout_store == 0) cout_store = Cout;
    if ((cin_real != Cin && (cin_real != 3 || Cin != 16 || ps_in)) || (cout_store != Cout && ps)) return PESR_EINVAL;
    if (ps_in && (Cin % 64 || stride != 1)) return PESR_EINVAL;
    if (ps && Cout % 4) return PESR_EINVAL;
    if (Cin % 16 || (stride != 1 && stride != 2) || N <= 0 || H <= 0 || W <= 0) return PESR_EINVAL;
    ConvArgs a{};
    a.x = x; a.wp = wp; a.bias = bias; a.skip = skip; a.mask = mask; a.y = y;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.OH = (H - 1) / stride + 1; a.OW = (W - 1) / stride + 1;
    a.GH = a.OH; a.GW = a.OW;
    a.in_oy = -1; a.in_ox = -1;
    a.out_my = 1; a.out_ay = 0; a.out_mx = 1; a.out_ax = 0;
    a.ntaps = 9;
    for (int t = 0; t < 9; ++t) set_tap(a, t, t / 3, t % 3, flip ? 8 - t : t);
    a.alpha = alpha; a.slope = slope; a.act = act; a.ps = ps; a.ps_in = ps_in;
    a.cin_real = cin_real; a.cout_store = cout_store;
    a.slab = (float*)ws; a.slab_bytes = ws_bytes;
    set_bn(a, fuse);
    if (fuse && fuse->mode == 2 && (mask || skip || bias || act != PESR_ACT_NONE)) return PESR_EINVAL;
    const int rc = stride == 1 ? dispatch<1>(a, 3, 3, stream) : dispatch<2>(a, 3, 3, stream);
    if (fuse) fuse->rows_out = rc ? 0 : a.bn_rows;
    return rc;
}

// Input gradient of a stride-2 3x3 conv (pad 1): dx[y][x] = sum over taps with (y+1-ky), (x+1-kx)
// even of dy[(y+1-ky)/2][(x+1-kx)/2] * W[ky][kx].  The four output parity classes (y&1, x&1) are
// four small stride-1 problems over dy with 1, 2, 2 and 4 taps - no multiply-by-zero work.
// wp is the dgrad packing (pack.hip mode 1; tap index = ky*3+kx of the forward weights).
int pesr_conv3x3_s2_dgrad_launch(const float* dy, const float* wp, const float* mask, float* dx, int N, int H, int W,
                                 int Cout_fwd, int Cin_fwd, float alpha, hipStream_t stream, PesrBnFuseArgs* fuse) {
    // H, W: spatial size of dx (the forward input); dy is [N][OH][OW][Cout_fwd]
    if (Cout_fwd % 16) return PESR_EINVAL;
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    ConvArgs4 a4{};
    int hext[4], wext[4];
    for (int k = 0; k < 4; ++k) {
        const int py = k < 2 ? 1 : 0, px = (k == 0 || k == 2) ? 1 : 0;      // (1,1), (1,0), (0,1), (0,0): most taps first
        ConvArgs& a = a4.c[k];
        a.x = dy; a.wp = wp; a.bias = nullptr; a.skip = nullptr; a.mask = mask; a.y = dx;
        a.N = N; a.H = OH; a.W = OW; a.Cin = Cout_fwd; a.Cout = Cin_fwd;
        a.OH = H; a.OW = W;
        a.GH = (H - py + 1) / 2; a.GW = (W - px + 1) / 2;  // number of u with 2u+py < H
        a.in_oy = 0; a.in_ox = 0;
        a.out_my = 2; a.out_ay = py; a.out_mx = 2; a.out_ax = px;
        // rows: py==0 -> ky=1 reads dy row u ; py==1 -> ky=0 reads u+1, ky=2 reads u
        int kys[2], dys[2], nky, kxs[2], dxs[2], nkx;
        if (py == 0) { nky = 1; kys[0] = 1; dys[0] = 0; } else { nky = 2; kys[0] = 0; dys[0] = 1; kys[1] = 2; dys[1] = 0; }
        if (px == 0) { nkx = 1; kxs[0] = 1; dxs[0] = 0; } else { nkx = 2; kxs[0] = 0; dxs[0] = 1; kxs[1] = 2; dxs[1] = 0; }
        a.ntaps = 0;
        for (int i = 0; i < nky; ++i)
            for (int j = 0; j < nkx; ++j) {
                set_tap(a, a.ntaps, dys[i], dxs[j], kys[i] * 3 + kxs[j]);
                ++a.ntaps;
            }
        a.alpha = alpha; a.slope = 0.f; a.act = PESR_ACT_NONE; a.ps = 0; a.ps_in = 0;
        a.cin_real = Cout_fwd; a.cout_store = Cin_fwd;
        a.slab = nullptr; a.slab_bytes = 0; a.ksplit = 1;
        set_bn(a, fuse);
        hext[k] = py + 1; wext[k] = px + 1;
    }
    if (fuse && fuse->mode == 2 && mask) return PESR_EINVAL;
    if (fuse) fuse->rows_out = 0;
    // One configuration for the four classes.  The 48-pixel tiles wherever ONE class alone would leave the 144-pixel tiles at about
    // one workgroup per CU (measured with the four classes in one grid, same box: 256 <- 256 @48: 112.7 us against 154.6 with the
    // 144 x 256 tiles; 512 <- 512 @24: 120.6 against 183.2; four launches: 139.7 / 185.8), else the largest tile that fills the chip
    // with the four classes counted together.
    const int Cout = Cin_fwd;
    const long M = (long)N * a4.c[0].GH * a4.c[0].GW;          // pixels of one class
    if (Cout % 64) {
        // (channel counts the merged kernel has no configuration for: one launch per class, as before)
        if (fuse && !fuse->dry && fuse->mode) return PESR_EINVAL;
        if (fuse && fuse->dry) return PESR_OK;
        for (int k = 0; k < 4; ++k) {
            if (a4.c[k].GH <= 0 || a4.c[k].GW <= 0) continue;
            const int rc = dispatch<1>(a4.c[k], hext[k], wext[k], stream);
            if (rc) return rc;
        }
        return PESR_OK;
    }
    int rc;
    if (Cout >= 128 && (long)pesr_cdiv(M, 144) * (Cout / 64) <= 320) rc = launch_s2dgrad4<1, 4, 3, 1, 2>(a4, hext, wext, stream);
    else if (Cout % 256 == 0 && 4 * (M / 144) * (Cout / 256) >= 192) rc = launch_s2dgrad4<1, 8, 9, 2, 2>(a4, hext, wext, stream);
    else if (Cout % 128 == 0 && 4 * (M / 144) * (Cout / 128) >= 192) rc = launch_s2dgrad4<1, 8, 9, 1, 2>(a4, hext, wext, stream);
    else rc = launch_s2dgrad4<1, 4, 9, 1, 4>(a4, hext, wext, stream);
    if (fuse) fuse->rows_out = rc ? 0 : a4.c[0].bn_rows;
    return rc;
}
