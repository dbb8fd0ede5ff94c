// 3x3 stride-1 convolution with a 1-D Winograd F(4,3) transform along x, on the fp32-input MFMA, gfx950.
//
// Same contract as conv3x3_mfma.hip / conv3x3_wino.hip (reference nn.Conv2d(k=3, padding=1), model/basic.py:4-7, forward
// and - with dgrad-transformed weights - input gradient) for widths that are multiples of 4 and Cout % 64 == 0, with HALF of
// the direct conv's multiplies (F(2,3): 2/3).  For an x-tile of four output pixels (4t .. 4t+3) of a row and the six input
// columns d0..d5 = x[4t-1 .. 4t+4]  (interpolation points 0, +-1, +-2, inf):
//     V0 = 4d0 - 5d2 + d4        V1 = (d4 - 4d2) + (d3 - 4d1)     V2 = (d4 - 4d2) - (d3 - 4d1)
//     V3 = (d4 - d2) + 2(d3 - d1)  V4 = (d4 - d2) - 2(d3 - d1)     V5 = 4d1 - 5d3 + d5
//     U  = G g  (pack time, wino4_pack.h)
//     M_xi = sum_{ky, ci} V_xi[row + ky - 1][ci] * U_xi[ky][ci]          (6 accumulators instead of 4 outputs x 3 taps)
//     y0 = M0 + (M1+M2) + (M3+M4),  y1 = (M1-M2) + 2(M3-M4),  y2 = (M1+M2) + 4(M3+M4),  y3 = (M1-M2) + 8(M3-M4) + M5.
// Measured against an fp64 conv the relative error is ~1.2e-6..1.7e-6 of the output's maximum at 256 input channels (direct
// / F(2,3): 3..4e-7; scripts/wino4_error.py), a factor 60 inside the stated gradient tolerance (1e-4).
//
// One workgroup = 144 x-tiles (TR rows x TXT tiles = 576 output pixels) x 64 output channels, 8 waves.  Wave w owns the
// 16 channels cb = w & 3 and HALF of the xi planes (xh = w >> 2: xi 3xh .. 3xh+2) for all 9 m-tiles: 27 accumulator tiles
// (108 VGPRs).  The two halves meet in the epilogue (partial output transforms summed through LDS, fixed order).
// Per 16-channel chunk:
//   * A = V[halo row][xi][x-tile][16 ch] in LDS, double buffered.  The chunk's raw input goes global -> registers (issued
//     at the top of the previous chunk) -> transform on the VALU -> ds_write into the OTHER V buffer two thirds into the
//     previous chunk's MFMA stream: no raw image in LDS, no transform pass, ONE barrier per chunk.
//   * B = the wave's nine [16 ch][16 k] weight slabs, loaded straight from global memory into registers two slabs ahead
//     (a wave reads only its own 1 KiB of a slab, so there is nothing to share through LDS).
//   * fragments in groups of 3 m-tiles, the next group's A reads issued under the current group's 12 MFMAs.
// The 16-byte k-groups of a V entry are XOR-swizzled by ((x-tile >> 1) ^ row term) so that the ds_read_b128 fragment reads
// are bank-conflict free for TXT = 12 (48-wide images: row term 2 * (halo row & 1)) and TXT = 8 / 16 / 24 (no row term).
// Layers with too few tiles split the Cin chunks over workgroups (raw partial sums + the direct kernel's finish kernel).
#include <mutex>
#include "common.h"
#include "launchers.h"
#include "wino4_pack.h"

struct Wino4Args {
    const float* x;     // [N][H][W][Cin]
    const float* wp;    // packed, transformed weights [3*6][Cin/16][Cout][16]
    const float* bias;  // [Cout] or null
    const float* skip;  // [N][H][W][Cout] or null
    const float* mask;  // [N][H][W][Cout] or null : result zeroed where mask <= 0
    float* y;           // [N][H][W][Cout]
    int N, H, W, Cin, Cout;
    int TR, TXT;        // tile: TR output rows x TXT x-tiles (TR * TXT == 144)
    int tiles_x, tiles_y, n_tiles;
    int HT;             // V rows: TR + 2
    int row_key;        // 2 when the swizzle key carries the halo-row parity (TXT % 8 == 4), else 0
    float alpha, slope;
    int act;
    int ps;             // 1: output stored pixel-shuffled (r = 2): packed channel (2*si+sj)*C + c -> y[n][2oy+si][2ox+sj][c], C = Cout/4
    int ps_in;          // 1: x is a pixel-shuffled tensor [N][2H][2W][Cin/4] read as its sub-pixel-major [N][H][W][Cin] view
    int ksplit;         // > 1: the Cin chunks are split over ksplit workgroups per tile; raw partial sums go to slab[ks][...]
    int chunks_per_split;
    float* slab;
    int stack;          // 0, or H + 1: the N images are tiled as ONE image of N * (H + 1) - 1 rows, a zero row between neighbours
    int stack_n;        //   (that row is the bottom halo of one image and the top halo of the next); N above is then 1
    int v_row;          // bytes of one V row: 6 * TXT * 64, plus the pad of the dense layout
    };

constexpr int W4_BN = 64, W4_MG = 9;

// Wave w owns the 16 channels w & 3 and HALF of the xi planes (w >> 2: xi 0..2 / 3..5): 27 accumulator tiles, 2 waves per SIMD.
// (A 12-wave variant - a third of the xi planes per wave, 3 waves per SIMD in the 168-VGPR budget - measured 2 % slower.)

// DENSE = false: the swizzle key of the rows of 8 / 12 / 16 / 24 x-tiles (the layers that matter), a_off ^ kxor for odd ky.
// DENSE = true: any row length.  A V row is padded so that the 64-byte entries of consecutive x-tiles m = row * TXT + txt of one
// xi plane fall into 64-byte slots m mod 4 of the 256-byte bank window, and the 16-byte sub-slot is rotated by (m >> 2) & 3:
// the 16 lanes of a fragment read (16 consecutive m) then touch every bank once.  The key of the row ky below is a different
// function of the lane, so its XOR with the ky = 0 key comes from two packed per-lane tables (2 bits per m-tile and ky).
// TXTC: the row length in x-tiles as a compile-time constant (12 / 24: the 48- and 96-wide layers, i.e. the G body and upsample.2),
// or 0 = a.TXT at run time.  With it the (ky, xi) part of a fragment read's address is an immediate of the ds_read and "this
// chunk's V buffer" is one add per fragment offset and chunk: the loop loses its per-read address add (87 -> 38 v_add_u32 per
// chunk and wave; G body forward 185.1 -> 182.6 us, 227.7 -> 229.4 patches/s).
// BNF: the instantiation carries the BatchNorm-sums epilogue (common.h BnEpi; the Discriminator's layers).  false: none of that code -
// the Generator's / VGG's kernels keep the register allocation they had (with it, the 18 prefetched z values of mode 2 cost the
// 48-wide instantiation 52 bytes of scratch in its epilogue).
template <bool DENSE, int TXTC, bool BNF = false>
__global__ __launch_bounds__(512) void conv3x3_wino4_kernel(const Wino4Args a, const BnEpi bn_arg) {
    (void)bn_arg;                                          // BatchNorm sums from the epilogue (common.h BnEpi): read through pesr_bn_epi() behind the main loop
    static_assert(!(DENSE && TXTC), "the constant-row-length form is for the non-dense layout");
    constexpr int NT = 512;                                // threads of the workgroup
    constexpr int NXL = 3;                                 // xi planes per wave
    constexpr int NSLAB = 3 * NXL;                         // weight slabs per wave and chunk: (ky, xl)
    constexpr int NU = 2;                                  // staging items per thread (HT * TXT * 4 <= NU * NT)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int TXTv = TXTC ? TXTC : a.TXT;                   // (a compile-time row length also turns the index divisions below into multiplies)
    const int plane = TXTv * 64;       // bytes of one xi plane of a V row
    const int v_row = TXTC ? 6 * TXTC * 64 : a.v_row;
    const int v_bytes = a.HT * v_row;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int cb = wave & 3, xt = wave >> 2;               // channel block, xi group
    // xi plane of this wave's local index xl
    auto xi_of = [&](int xl) -> int { return xt * 3 + xl; };

    // blockIdx -> (split-K slice, pixel tile, n-tile).  Workgroups b and b + 8 share an XCD (round-robin dispatch): give every
    // XCD a contiguous range of logical tiles, n-tile fastest, so the n_tiles workgroups that read the same pixels share an L2.
    int b = blockIdx.x;
    if ((gridDim.x & 7) == 0) b = (b & 7) * (gridDim.x >> 3) + (b >> 3);
    const int tiles_total = a.n_tiles * a.tiles_x * a.tiles_y * a.N;
    const int ks = b / tiles_total;
    int bid = b - ks * tiles_total;
    const int nt = bid % a.n_tiles;  bid /= a.n_tiles;
    const int tx = bid % a.tiles_x;  bid /= a.tiles_x;
    const int ty = bid % a.tiles_y;
    const int img = bid / a.tiles_y;
    const int gy0 = ty * a.TR, gt0 = tx * TXTv;          // first output row / first x-tile of the tile
    const int n0 = nt * W4_BN;
    const int C16T = a.Cin >> 4;
    const int CB = ks * a.chunks_per_split;                // this workgroup's chunk range [CB, CB + C16)
    const int C16 = (C16T - CB) < a.chunks_per_split ? (C16T - CB) : a.chunks_per_split;

    // ---- A fragment offsets: lane (r, g) reads k-group g of x-tile m = 16 i + r (for an even ky; odd ky: ^ kxor) -------------
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;   // LDS address of the dynamic segment (a multiple of 256)
    int a_off[W4_MG];
    unsigned ktab1 = 0, ktab2 = 0;                         // DENSE: (key(ky) ^ key(0)) << 4 for m-tile i at bits [2i + 4, 2i + 6)
    // (filled in behind the prologue's loads: nothing needs the offsets before the first fragment read, and the index arithmetic
    // then runs under the first chunk's load latency instead of in front of it)
    auto compute_a_off = [&]() {
#pragma unroll
        for (int i = 0; i < W4_MG; ++i) {
            const int m = i * 16 + r;
            const int trow = m / TXTv, txt = m - trow * TXTv;
            if (DENSE) {
                const int k0 = (m >> 2) & 3;
                a_off[i] = trow * v_row + txt * 64 + ((g ^ k0) & 3) * 16;
                ktab1 |= (unsigned)((((m + TXTv) >> 2) & 3) ^ k0) << (2 * i + 4);
                ktab2 |= (unsigned)((((m + 2 * TXTv) >> 2) & 3) ^ k0) << (2 * i + 4);
            } else {
                a_off[i] = trow * v_row + txt * 64 + ((g ^ (txt >> 1) ^ (a.row_key * (trow & 1))) & 3) * 16;
                if (TXTC) a_off[i] += xt * 3 * plane;          // the wave's xi half: the rest of (ky, xi) is an immediate
            }
            a_off[i] += (int)lds0;
        }
    };
    const int kxor = a.row_key * 16;                       // the key's row term flips with the parity of ky

    // ---- B: this lane's 16 bytes of slab (ky, xi, chunk): wave-uniform slab base (SGPRs) + one per-lane 32-bit byte offset ----
    // (a buffer load: the slab offset is a scalar operand, so a weight fetch needs no 64-bit vector address arithmetic; the packed
    // weights are < 4 GB: 18 * Cin * Cout floats)
    const unsigned slab_bytes = (unsigned)a.Cout * 64;     // bytes between consecutive chunks of one (ky, xi)
    const unsigned b_lane = (unsigned)(((n0 + cb * 16 + r) * 16 + g * 4) * 4);
    const __amdgpu_buffer_rsrc_t w_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.wp, 0, (unsigned)((size_t)18 * a.Cin * a.Cout * 4), 0x00020000);
    auto ldb = [&](int ky, int xl, int cc) -> f32x4 {      // cc = absolute chunk
        const unsigned so = (unsigned)((ky * 6 + xi_of(xl)) * C16T + cc) * slab_bytes;
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, b_lane, so, 0));
    };

    // ---- staging items: (halo row, x-tile, 4-channel group); six input columns each; NU items per thread --------------------
    const float* const x_img = a.x + (size_t)img * a.H * a.W * a.Cin;    // stacked: img == 0, rows run over all images
    const int n_items = a.HT * TXTv * 4;
    const int Cq = a.Cin >> 2;
    // Out-of-image columns (and whole halo rows) are fetched at offset 2^31, beyond the buffer descriptor's range: the load
    // returns zeros, no masking afterwards.  (Raw buffer: the range check is on the VGPR offset; images are < 2 GB - w4_plan.)
    unsigned st_off[NU][6];                                // byte offsets inside the image
    int st_dst[NU];                                        // < 0: the item does not exist
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int it = tid + u * NT;
        const int q = it & 3, rest = it >> 2;
        const int hrow = rest / TXTv, txt = rest - hrow * TXTv;
        int iy = gy0 - 1 + hrow;
        const int ix0 = 4 * (gt0 + txt) - 1;
        bool item_ok = it < n_items && iy >= 0 && iy < a.H;
        if (a.stack) {                                      // virtual row -> (image, row); the separator rows read zeros
            const int im = iy / a.stack, yy = iy - im * a.stack;
            item_ok = it < n_items && iy >= 0 && im < a.stack_n && yy < a.H;
            iy = im * a.H + yy;
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int ix = ix0 + j;
            const bool ok = item_ok && ix >= 0 && ix < a.W;
            const int pix = a.ps_in ? ((2 * iy) * (2 * a.W) + 2 * ix) * Cq : (iy * a.W + ix) * a.Cin;
            st_off[u][j] = ok ? (unsigned)((pix + q * 4) * 4) : 0x80000000u;
        }
        const int skey = DENSE ? ((hrow * TXTv + txt) >> 2) : ((txt >> 1) ^ (a.row_key * (hrow & 1)));
        st_dst[u] = it < n_items ? hrow * v_row + txt * 64 + ((q ^ skey) & 3) * 16 : -1;
    }
    const __amdgpu_buffer_rsrc_t x_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)x_img, 0, (unsigned)((size_t)(a.stack ? a.stack_n : 1) * a.H * a.W * a.Cin * 4), 0x00020000);
    auto chunk_off = [&](int cc) -> int {                  // channel part of an input address (bytes), chunk cc (absolute)
        int coff = cc * 16;
        if (a.ps_in) {   // chunk = channels [16cc, 16cc+16) of sub-pixel `sub`: one pixel of the shuffled tensor
            const int sub = coff / Cq, cc0 = coff - sub * Cq;
            coff = ((sub >> 1) * (2 * a.W) + (sub & 1)) * Cq + cc0;
        }
        return coff * 4;
    };
    // The two items of a thread go through the SAME six staging registers one after the other: item 0 is loaded
    // at the top of a chunk and stored a third in, item 1 is loaded right there and stored two thirds in.
    u32x4 sx[6];
    auto stage_load = [&](int u, int cc) {
        const int so = __builtin_amdgcn_readfirstlane(chunk_off(cc));
#pragma unroll
        for (int j = 0; j < 6; ++j) sx[j] = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, st_off[u][j], so, 0);
    };
    auto stage_store = [&](int u, char* vdst) {
        if (st_dst[u] >= 0) {
            const f32x4 d0 = __builtin_bit_cast(f32x4, sx[0]), d1 = __builtin_bit_cast(f32x4, sx[1]),
                        d2 = __builtin_bit_cast(f32x4, sx[2]), d3 = __builtin_bit_cast(f32x4, sx[3]),
                        d4 = __builtin_bit_cast(f32x4, sx[4]), d5 = __builtin_bit_cast(f32x4, sx[5]);
            char* p = vdst + st_dst[u];
            // (signed constants: written as d4 - 5.0f * d2 hipcc negates d2 with a v_xor per register in front of each v_pk_fma - 20 of
            // the loop's 154 VALU instructions per wave and chunk, and every VALU instruction costs the fp32 MFMA pipe ~3.7 cycles)
            const f32x4 m5 = {-5.0f, -5.0f, -5.0f, -5.0f}, m4 = {-4.0f, -4.0f, -4.0f, -4.0f}, p4 = {4.0f, 4.0f, 4.0f, 4.0f};
            *(f32x4*)(p) = __builtin_elementwise_fma(p4, d0, __builtin_elementwise_fma(m5, d2, d4));
            *(f32x4*)(p + 5 * plane) = __builtin_elementwise_fma(p4, d1, __builtin_elementwise_fma(m5, d3, d5));
            const f32x4 t1 = __builtin_elementwise_fma(m4, d2, d4), t2 = __builtin_elementwise_fma(m4, d1, d3);
            *(f32x4*)(p + plane) = t1 + t2;
            *(f32x4*)(p + 2 * plane) = t1 - t2;
            const f32x4 t3 = d4 - d2, t4 = d3 - d1;
            const f32x4 p2 = {2.0f, 2.0f, 2.0f, 2.0f}, m2 = {-2.0f, -2.0f, -2.0f, -2.0f};
            *(f32x4*)(p + 3 * plane) = __builtin_elementwise_fma(p2, t4, t3);
            *(f32x4*)(p + 4 * plane) = __builtin_elementwise_fma(m2, t4, t3);
        }
    };

    f32x4 acc[NXL][W4_MG];
#pragma unroll
    for (int xl = 0; xl < NXL; ++xl)
#pragma unroll
        for (int i = 0; i < W4_MG; ++i) acc[xl][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    f32x4 fa[2][3], fb[3];
    // the xor of the odd-ky reads is redone at every use: hoisted out of the loop it would cost nine more live registers
    auto opaque = [](int v) -> int { asm volatile("" : "+s"(v)); return v; };
    auto a_key = [&](const int i, const int ky) -> int {   // fragment offset of m-tile i for the V row ky below the tile row
        if (DENSE) {
            if (ky == 0) return a_off[i];
            const unsigned tab = ky == 1 ? ktab1 : ktab2;
            return a_off[i] ^ (int)((tab >> (2 * i)) & 0x30u);
        }
        return (ky & 1) ? (a_off[i] ^ opaque(kxor)) : a_off[i];
    };
    // Fragment reads through 32-bit LDS addresses (a_off carries the dynamic-LDS base, added once): written as `smem + offset`
    // every read paid a v_add_u32 of the base's relocation - a literal 0 hipcc cannot fold (36 per chunk and wave).
#define W4_READ_A(FA, VB, KY, XL, GRP)                                                                   \
    {                                                                                                    \
        const unsigned vb_ = TXTC ? (unsigned)((KY) * v_row + (XL) * plane)                              \
                                  : (unsigned)((VB) - smem) + (unsigned)((KY) * v_row + xi_of(XL) * plane); \
        _Pragma("unroll") for (int i = 0; i < 3; ++i)                                                    \
            FA[i] = *(const __attribute__((address_space(3))) f32x4*)(size_t)((unsigned)a_key((GRP) * 3 + i, KY) + vb_); \
    }
#define W4_MFMA(FA, FB, XL, GRP)                                                                         \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk)                                                     \
        _Pragma("unroll") for (int i = 0; i < 3; ++i)                                                    \
            acc[XL][(GRP) * 3 + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(FB[kk], FA[i][kk], acc[XL][(GRP) * 3 + i], 0, 0, 0);

    // ---- prologue: chunk CB staged synchronously, the first two weight slabs -------------------------------------------------
    {
        u32x4 sy[6];                                       // item 1 in registers of its own here: one load latency, not two
        const int so = __builtin_amdgcn_readfirstlane(chunk_off(CB));
        stage_load(0, CB);
#pragma unroll
        for (int j = 0; j < 6; ++j) sy[j] = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, st_off[1][j], so, 0);
        fb[0] = ldb(0, 0, CB);
        fb[1] = ldb(0, 1, CB);
        compute_a_off();
        stage_store(0, smem);
#pragma unroll
        for (int j = 0; j < 6; ++j) sx[j] = sy[j];
        stage_store(1, smem);
    }
    __syncthreads();

#pragma unroll 1
    for (int c = 0; c < C16; ++c) {
        char* const vcur = smem + (c & 1) * v_bytes;      // (TXTC: the fragment offsets carry the buffer, moved below)
        char* const vnext = smem + ((c & 1) ^ 1) * v_bytes;
        // No branch in the loop: the last chunk "prefetches" itself again (loads, transforms and LDS stores nobody consumes).  With
        // the prefetch under `if (more)` the staging registers and weight fragments had two definitions merging at the loop header,
        // and hipcc could not count the outstanding loads exactly.
        const int cn = CB + (c + 1 < C16 ? c + 1 : c);
        stage_load(0, cn);                                 // lands while this chunk computes
        W4_READ_A(fa[0], vcur, 0, 0, 0)
#pragma unroll
        for (int s = 0; s < NSLAB; ++s) {                  // slab s = (ky, xl)
            const int ky = s / NXL, xl = s - ky * NXL;
            // weight slab s + 2 (of this chunk, or the first ones of the next)
            {
                if (s + 2 < NSLAB) fb[(s + 2) % 3] = ldb((s + 2) / NXL, (s + 2) % NXL, CB + c);
                else fb[(s + 2) % 3] = ldb(0, s + 2 - NSLAB, cn);
            }
#pragma unroll
            for (int grp = 0; grp < 3; ++grp) {
                const int t = s * 3 + grp, cur = t & 1;
                if (grp < 2) W4_READ_A(fa[cur ^ 1], vcur, ky, xl, grp + 1)
                else if (s + 1 < NSLAB) W4_READ_A(fa[cur ^ 1], vcur, (s + 1) / NXL, (s + 1) % NXL, 0)
                __builtin_amdgcn_sched_barrier(0);         // keep the prefetch ahead of the MFMA group (hipcc sinks it next to its use)
                W4_MFMA(fa[cur], fb[s % 3], xl, grp)
                __builtin_amdgcn_sched_barrier(0);
            }
            // A third / two thirds in, the staged loads have landed: transform them into the other V buffer under the MFMAs.
            // (raised priority while a wave stages: its VALU instructions and LDS stores issue ahead of the other wave's MFMAs instead of
            // between them - G body 172.0 -> 171.2 us with bias + ReLU, 176.0 -> 174.3 with the skip, 177.3 -> 175.2 with the mask)
            if (s == 2) { __builtin_amdgcn_s_setprio(3); stage_store(0, vnext); stage_load(1, cn); __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_setprio(0); }
            if (s == 6) { __builtin_amdgcn_s_setprio(3); stage_store(1, vnext); __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_setprio(0); }
        }
        if (TXTC) {
            const int dv = (c & 1) ? -v_bytes : v_bytes;
#pragma unroll
            for (int i = 0; i < W4_MG; ++i) a_off[i] += dv;
        }
        __syncthreads();                                   // V[next] complete and visible; everyone is done with V[cur]
    }
#undef W4_READ_A
#undef W4_MFMA
    // ---- epilogue ------------------------------------------------------------------------------------------------------------
    // With the weights as the MFMA's A operand a lane holds, per m-tile i, FOUR CONSECUTIVE CHANNELS (cb*16 + 4g ..) of x-tile
    // 16 i + r for its three xi planes.  y0 = M0 + (M1+M2) + (M3+M4), y1 = (M1-M2) + 2(M3-M4), y2 = (M1+M2) + 4(M3+M4),
    // y3 = (M1-M2) + 8(M3-M4) + M5: the xi 0..2 wave finishes y0, y1 and the xi 3..5 wave y2, y3; each passes the other its two
    // partial terms through LDS (lane-linear 16-byte slots: same lane of the partner wave), adds what it receives and stores
    // 16 bytes per lane straight to global memory - the four channel-block waves fill a pixel's 256-byte line between them.
    char* const xb = smem;
    // slot (sender xh, i, k, cb, lane)
    auto slot = [&](int sender, int i, int k) -> char* { return xb + ((((sender * W4_MG + i) * 2 + k) * 4 + cb) * 64 + lane) * 16; };
    f32x4 keep[W4_MG][2];
#pragma unroll
    for (int i = 0; i < W4_MG; ++i) {
        const f32x4 q0 = acc[0][i], q1 = acc[1][i], q2 = acc[2][i];
        if (xt == 0) {
            const f32x4 sm = q1 + q2, df = q1 - q2;
            keep[i][0] = q0 + sm; keep[i][1] = df;
            *(f32x4*)slot(0, i, 0) = sm; *(f32x4*)slot(0, i, 1) = df;
        } else {
            const f32x4 sm = q0 + q1, df = q0 - q1;
            keep[i][0] = 4.0f * sm; keep[i][1] = 8.0f * df + q2;
            *(f32x4*)slot(1, i, 0) = sm; *(f32x4*)slot(1, i, 1) = 2.0f * df;
        }
    }
    const size_t img_out = (size_t)img * a.H * a.W;
    const int co = n0 + cb * 16 + g * 4;
    // output e = (m-tile i, k) of this lane -> (inside the image, element offset of its four channels)
    auto out_index = [&](const int i, const int k, size_t* idx) -> bool {
        const int m = i * 16 + r;
        const int trow = m / TXTv, txt = m - trow * TXTv;
        int oy = gy0 + trow;
        const int ox = 4 * (gt0 + txt) + 2 * xt + k;
        bool ok = oy < a.H && ox < a.W;
        if (a.stack) {                                  // virtual row -> row of the [N * H] row space; separator rows are dropped
            const int im = oy / a.stack, yy = oy - im * a.stack;
            ok = im < a.stack_n && yy < a.H && ox < a.W;
            oy = im * a.H + yy;
        }
        if (a.ps) {   // packed channel co = (2*si+sj)*C + c  ->  out[n][2*oy+si][2*ox+sj][c]
            const int C = a.Cout >> 2;
            const int sub = co / C, cc = co - sub * C;
            *idx = (((size_t)img * (2 * a.H) + 2 * oy + (sub >> 1)) * (2 * a.W) + 2 * ox + (sub & 1)) * C + cc;
        } else {
            *idx = (img_out + (size_t)oy * a.W + ox) * a.Cout + co;
        }
        if (!ok) *idx = 0;
        return ok;
    };
    __syncthreads();
    // BatchNorm mode 2 reads z at every output element: all 18 loads of a lane are issued here, right behind the exchange barrier (in
    // front of it the barrier's vmcnt(0) would wait for them with nothing to overlap - conv3x3_mfma.hip measured both)
    const BnEpi* const bn = pesr_bn_epi((unsigned)((sizeof(Wino4Args) + 7) & ~(size_t)7));
    const int bn_mode = (BNF && a.ksplit == 1) ? bn->mode : 0;
    f32x4 zall[BNF ? 2 * W4_MG : 1];
    if (BNF && bn_mode == 2) {
        const float* const bn_z = bn->z;
#pragma unroll
        for (int e = 0; e < 2 * W4_MG; ++e) {
            size_t idx;
            out_index(e >> 1, e & 1, &idx);
            zall[e] = *(const f32x4*)(bn_z + idx);           // (an out-of-image element reads offset 0: valid memory, value unused)
        }
    }
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (a.bias && a.ksplit == 1) bias4 = *(const f32x4*)(a.bias + co);
    const bool bn_on = BNF && bn_mode != 0;
    const float bn_slope = BNF ? bn->slope : 0.f;
    f32x4 st1 = {0.f, 0.f, 0.f, 0.f}, st2 = {0.f, 0.f, 0.f, 0.f};
    f32x4 bmu = st1, bis = st1, bga = st1, bbe = st1;
    if (BNF && bn_mode == 2) {
        bmu = *(const f32x4*)(bn->mi + co); bis = *(const f32x4*)(bn->mi + a.Cout + co);
        bga = *(const f32x4*)(bn->gamma + co); bbe = *(const f32x4*)(bn->beta + co);
    }
    // batches of three m-tiles x two outputs: a batch's LDS reads and skip / mask loads are issued before its first store.
    // (Round 4 measured the skip / mask loads issued ONE BATCH AHEAD, the first batch's in front of the exchange barrier: 175.9 vs
    // 172.6 us with bias + ReLU, 179.4 vs 177.1 with the skip, 179.0 vs 177.3 with the mask - slower in every form; and a PERSISTENT
    // launch form for the layers with several rounds of workgroups per CU - 256 workgroups walking their tiles, the next tile's first
    // chunk staged under the last chunk, the exchange in three batches inside the dead V buffer: bit-identical and 7 - 32 % SLOWER
    // (scripts/diag/conv3x3_wino4_persist.hip): a wave's vmcnt retires in order, so a persistent wave cannot wait for its next loads
    // without waiting for its own output stores, while the NEXT workgroup's prologue on the same CU overlaps them for free.
    // profiles/r04_ab_notes.txt)
#pragma unroll
    for (int ib = 0; ib < W4_MG; ib += 3) {
        f32x4 v[6], mkv[6], skv[6];
        size_t idx[6];
        bool ok[6];
#pragma unroll
        for (int e = 0; e < 6; ++e) {
            const int i = ib + (e >> 1), k = e & 1;
            ok[e] = out_index(i, k, &idx[e]);
            // same order of additions as a sequential y = (xi 0..2 part) + (xi 3..5 part)
            v[e] = xt == 0 ? keep[i][k] + *(const f32x4*)slot(1, i, k) : *(const f32x4*)slot(0, i, k) + keep[i][k];
            if (a.ksplit == 1) {
                if (a.mask) mkv[e] = *(const f32x4*)(a.mask + idx[e]);
                if (a.skip) skv[e] = *(const f32x4*)(a.skip + idx[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < 6; ++e) {
            if (!ok[e]) continue;
            f32x4 o = v[e];
            if (a.ksplit > 1) {   // raw partial sums; the finish kernel applies the epilogue
                *(f32x4*)(a.slab + (size_t)ks * ((size_t)(a.stack ? a.stack_n : a.N) * a.H * a.W * a.Cout) + idx[e]) = o;
                continue;
            }
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
            if (bn_on) {
                if (bn_mode == 2) {
                    const f32x4 xh = (zall[BNF ? ib * 2 + e : 0] - bmu) * bis;
                    const f32x4 zz = bga * xh + bbe;
                    o.x = zz.x > 0.f ? o.x : o.x * bn_slope; o.y = zz.y > 0.f ? o.y : o.y * bn_slope;
                    o.z = zz.z > 0.f ? o.z : o.z * bn_slope; o.w = zz.w > 0.f ? o.w : o.w * bn_slope;
                    st1 += o; st2 += o * xh;
                } else {
                    st1 += o; st2 += o * o;
                }
            }
            *(f32x4*)(a.y + idx[e]) = o;
        }
    }
    if (bn_on) {
        // a lane's 18 outputs are pixels of ITS four channels (co .. co + 3): the 16 r-lanes x 2 xi-half waves of a (cb, g) pair
        // meet through LDS and are added in double, in the fixed order (xt, r)
        __syncthreads();                                   // the exchange slots are dead
        f32x4* const red = (f32x4*)smem;                   // [2][8 waves][64 lanes]
        red[wave * 64 + lane] = st1; red[512 + wave * 64 + lane] = st2;
        __syncthreads();
        if (tid < 16) {                                    // tid = cb * 4 + g: channels n0 + 4 tid .. + 3
            const int cb_ = tid >> 2, g_ = tid & 3;
            f64x4 d1 = {0.0, 0.0, 0.0, 0.0}, d2 = {0.0, 0.0, 0.0, 0.0};
            for (int xt_ = 0; xt_ < 2; ++xt_)
                for (int r_ = 0; r_ < 16; ++r_) {
                    const int e = (xt_ * 4 + cb_) * 64 + g_ * 16 + r_;
                    d1 += __builtin_convertvector(red[e], f64x4);
                    d2 += __builtin_convertvector(red[512 + e], f64x4);
                }
            const int row = (img * a.tiles_y + ty) * a.tiles_x + tx;
            float* const pr = bn->part + (size_t)row * 2 * a.Cout + n0 + tid * 4;
            *(f32x4*)pr = __builtin_convertvector(d1, f32x4);
            *(f32x4*)(pr + a.Cout) = __builtin_convertvector(d2, f32x4);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// weight transform + packing (wino4_pack.h)
__global__ void pack_wino4_kernel(const float* __restrict__ w, float* __restrict__ out, int O, int I, int mode, int ps) {
    const long total = 18L * O * I;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x)
        out[e] = pesr_wino4_pack_elem(w, O, I, mode, ps, e);
}

int pesr_pack_conv3x3_wino4_launch(const float* w, float* out, int O, int I, int mode, int ps, hipStream_t stream) {
    if (O % 16 || I % 16 || (mode != 0 && mode != 1) || (ps && O % 64)) return PESR_EINVAL;
    const long total = 18L * O * I;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_wino4_kernel, dim3(grid), dim3(256), 0, stream, w, out, O, I, mode, ps);
    return pesr_launch_status();
}

namespace {
struct W4Plan { int TR, TXT, tiles_x, tiles_y, n_tiles, ksplit, chunks_per_split; long tiles; size_t lds; int score; int stack, dense, v_row; };

static bool w4_clean(int TXT) { return TXT == 12 || TXT == 8 || TXT == 16 || TXT == 24; }   // rows the non-dense swizzle key serves
static int w4_v_row(int TXT) {   // bytes of a V row; other row lengths use the dense layout, padded to TXT * 64 (mod 256)
    const int raw = 6 * TXT * 64;
    return w4_clean(TXT) ? raw : raw + (256 - (5 * TXT * 64) % 256) % 256;
}

// Tile shape TR x TXT == 144 x-tiles with the least out-of-image area that fits LDS (two V buffers) and the 2 staging items per
// thread, for NI images of HI rows (stacked: one image of all rows, `real_rows` of them real); split-K over the Cin chunks when
// the tiles alone cannot fill the 256 CUs.  score = per-mille of the tiles' x-tile slots that hold real pixels, or 0 when the
// shape yields fewer than 192 workgroups.
static bool w4_plan_one(int NI, int HI, long real_rows, int W, int Cin, int Cout, bool allow_split, size_t ws_bytes, size_t out_bytes,
                        W4Plan* p) {
    const int XT = W / 4;
    long best = -1;
    for (int TXT = 1; TXT <= 144; ++TXT) {
        if (144 % TXT) continue;
        const int TR = 144 / TXT, HT = TR + 2;
        const size_t vb = (size_t)2 * HT * w4_v_row(TXT);
        if (vb > 160 * 1024 || HT * TXT * 4 > 2 * 512) continue;  // two staging items per thread
        const long cover = (long)pesr_cdiv(HI, TR) * TR * pesr_cdiv(XT, TXT) * TXT;
        // least waste first; then a row length served by the cheaper non-dense key (conflict-free for 12 with the row term and
        // for 8 / 16 / 24 without, scripts/lds_bank_probe.hip); then the smallest halo
        const long score = cover * 8192 + (w4_clean(TXT) ? 0 : 4096) + (long)HT * TXT;
        if (best < 0 || score < best) { best = score; p->TR = TR; p->TXT = TXT; }
    }
    if (best < 0) return false;
    p->dense = !w4_clean(p->TXT);
    p->v_row = w4_v_row(p->TXT);
    p->tiles_y = pesr_cdiv(HI, p->TR); p->tiles_x = pesr_cdiv(XT, p->TXT); p->n_tiles = Cout / W4_BN;
    p->tiles = (long)NI * p->tiles_y * p->tiles_x * p->n_tiles;
    const size_t vb = (size_t)2 * (p->TR + 2) * p->v_row, ob = (size_t)2 * W4_MG * 2 * 4 * 64 * 16;   // the epilogue's exchange slots
    p->lds = vb > ob ? vb : ob;
    const int C16T = Cin / 16;
    p->ksplit = 1; p->chunks_per_split = C16T;
    if (allow_split && p->tiles < 160 && C16T >= 8) {
        // the split that minimises (rounds of 256 workgroups) x (chunks per workgroup), each workgroup's prologue + epilogue
        // counted as one more chunk; at least 4 chunks per slice
        long best_cost = -1;
        for (int want = 1; want <= 8 && want <= C16T / 4; ++want) {
            if (want > 1 && (size_t)want * out_bytes > ws_bytes) break;
            const int cps = (C16T + want - 1) / want, ks = (C16T + cps - 1) / cps;
            const long cost = ((p->tiles * ks + 255) / 256) * (cps + 1);
            if (best_cost < 0 || cost < best_cost) { best_cost = cost; p->chunks_per_split = cps; p->ksplit = ks; }
        }
    }
    // The F(2,3) kernel's tiles (288 pixels x 128 channels) give the same workgroup count, so chip fill does not separate the
    // two; below ~3/4 of a round neither beats the direct kernel's smaller tiles.
    const long wgs = p->tiles * p->ksplit;
    const double cover_eff = (double)(real_rows * XT) / ((double)NI * p->tiles_y * p->TR * p->tiles_x * p->TXT);
    p->score = wgs >= 192 ? (int)(1000.0 * cover_eff) : 0;
    return true;
}

// Images shorter than a tile waste most of its rows (a 12 x 12 image fills a quarter of a 48-row x 3-x-tile tile).  When it pays,
// the N images are laid out as ONE image of N * (H + 1) - 1 rows with a zero row between neighbours - the bottom halo of one
// image and the top halo of the next - and tiled together (allow_stack: not with a fused PixelShuffle on either side).
static bool w4_plan(int N, int H, int W, int Cin, int Cout, bool allow_split, size_t ws_bytes, bool allow_stack, W4Plan* p) {
    if (N < 1 || H < 1 || W < 4 || W % 4 || Cin % 16 || Cin < 16 || Cout % W4_BN) return false;
    if ((size_t)H * W * Cin * 4 >= ((size_t)1 << 31)) return false;   // one image per buffer descriptor, offsets below 2^31
    const size_t out_bytes = (size_t)N * H * W * Cout * sizeof(float);
    if (!w4_plan_one(N, H, (long)N * H, W, Cin, Cout, allow_split, ws_bytes, out_bytes, p)) return false;
    p->stack = 0;
    if (allow_stack && N > 1 && (size_t)N * H * W * Cin * 4 < ((size_t)1 << 31)) {
        W4Plan q;
        if (w4_plan_one(1, N * (H + 1) - 1, (long)N * H, W, Cin, Cout, allow_split, ws_bytes, out_bytes, &q) && q.score > p->score + 50) {
            *p = q;
            p->stack = H + 1;
        }
    }
    return true;
}
}  // namespace

// per-mille of tile area inside the image (0: unsupported shape or too few workgroups).  allow_split = 1 assumes the caller
// passes the split-K workspace.
int pesr_conv3x3_wino4_score_impl(int N, int H, int W, int Cin, int Cout, int allow_split) {
    W4Plan p;
    if (!w4_plan(N, H, W, Cin, Cout, allow_split != 0, (size_t)-1, true, &p)) return 0;
    return p.score;
}

int pesr_conv3x3_wino4_launch(const float* x, const float* wp, const float* bias, const float* skip, const float* mask, float* y,
                              int N, int H, int W, int Cin, int Cout, float alpha, int act, float slope, int ps, int ps_in,
                              void* ws, size_t ws_bytes, hipStream_t stream, PesrBnFuseArgs* fuse) {
    W4Plan p;
    if (fuse) fuse->rows_out = 0;
    if (!w4_plan(N, H, W, Cin, Cout, (ws != nullptr || (fuse && fuse->dry)) && !ps, (fuse && fuse->dry) ? (size_t)-1 : ws_bytes, !ps && !ps_in, &p)) return PESR_EINVAL;
    // BatchNorm sums: one row per pixel tile from the epilogue; with split-K from the finish kernel that sums the slabs (one row per finish
    // workgroup); not with a shuffled store
    const long bn_rows = ps ? 0 : (p.ksplit == 1 ? p.tiles / p.n_tiles : pesr_conv_splitk_finish_bn_rows(Cout, p.ksplit));
    if (fuse) {
        fuse->rows_out = bn_rows;
        if (fuse->dry) return PESR_OK;
        if (fuse->mode && (bn_rows == 0 || bn_rows > fuse->rows || !fuse->part)) return PESR_EINVAL;
        if (fuse->mode == 2 && (mask || skip || bias || act != PESR_ACT_NONE)) return PESR_EINVAL;
    }
    if (ps && (Cout % 256 || skip || mask)) return PESR_EINVAL;    // a 64-channel n-tile must stay inside one sub-pixel plane
    if (ps_in && Cin % 64) return PESR_EINVAL;
    Wino4Args a{};
    a.x = x; a.wp = wp; a.bias = bias; a.skip = skip; a.mask = mask; a.y = y;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.alpha = alpha; a.slope = slope; a.act = act; a.ps = ps; a.ps_in = ps_in;
    a.TR = p.TR; a.TXT = p.TXT; a.HT = p.TR + 2;
    a.tiles_x = p.tiles_x; a.tiles_y = p.tiles_y; a.n_tiles = p.n_tiles;
    a.row_key = (p.TXT % 8 == 4) ? 2 : 0;
    a.ksplit = p.ksplit; a.chunks_per_split = p.chunks_per_split; a.slab = (float*)ws;
    a.stack = p.stack; a.stack_n = N; a.v_row = p.v_row;
    if (p.stack) a.N = 1;
    BnEpi bn{}, bn_fin{};
    if (fuse && fuse->mode) {
        bn.mode = fuse->mode; bn.part = fuse->part; bn.z = fuse->z; bn.mi = fuse->mean_invstd; bn.gamma = fuse->gamma;
        bn.beta = fuse->beta; bn.slope = fuse->slope;
        if (p.ksplit > 1) { bn_fin = bn; bn = BnEpi{}; }      // the conv kernel stores raw partial sums: the finish kernel does the BatchNorm part
    }
    static PesrDeviceOnce attr_once;
    attr_once([&] {
        (void)hipFuncSetAttribute((const void*)conv3x3_wino4_kernel<false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv3x3_wino4_kernel<false, 12>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv3x3_wino4_kernel<false, 24>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv3x3_wino4_kernel<true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv3x3_wino4_kernel<false, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv3x3_wino4_kernel<false, 12, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv3x3_wino4_kernel<false, 24, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv3x3_wino4_kernel<true, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    const dim3 grid((unsigned)(p.tiles * p.ksplit));
    if (bn.mode) {
        if (p.dense) hipLaunchKernelGGL((conv3x3_wino4_kernel<true, 0, true>), grid, dim3(512), p.lds, stream, a, bn);
        else if (p.TXT == 12) hipLaunchKernelGGL((conv3x3_wino4_kernel<false, 12, true>), grid, dim3(512), p.lds, stream, a, bn);
        else if (p.TXT == 24) hipLaunchKernelGGL((conv3x3_wino4_kernel<false, 24, true>), grid, dim3(512), p.lds, stream, a, bn);
        else hipLaunchKernelGGL((conv3x3_wino4_kernel<false, 0, true>), grid, dim3(512), p.lds, stream, a, bn);
    } else if (p.dense) hipLaunchKernelGGL((conv3x3_wino4_kernel<true, 0>), grid, dim3(512), p.lds, stream, a, bn);
    else if (p.TXT == 12) hipLaunchKernelGGL((conv3x3_wino4_kernel<false, 12>), grid, dim3(512), p.lds, stream, a, bn);
    else if (p.TXT == 24) hipLaunchKernelGGL((conv3x3_wino4_kernel<false, 24>), grid, dim3(512), p.lds, stream, a, bn);
    else hipLaunchKernelGGL((conv3x3_wino4_kernel<false, 0>), grid, dim3(512), p.lds, stream, a, bn);
    if (p.ksplit > 1 && bn_fin.mode)
        return pesr_conv_splitk_finish_bn_launch((const float*)ws, bias, y, (long)N * H * W * Cout, Cout, p.ksplit, alpha, bn_fin, stream);
    if (p.ksplit > 1)
        return pesr_conv_splitk_finish_launch((const float*)ws, bias, skip, mask, y, (long)N * H * W * Cout, Cout, p.ksplit, alpha, act,
                                              slope, stream);
    return pesr_launch_status();
}
