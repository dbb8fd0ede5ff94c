// Weight gradient of the stride-1 3x3 conv with the transposed 1-D Winograd F(4,3) along x, fp32-input MFMA, gfx950.
//
// Same contract as conv3x3_wgrad.hip (ATen convolution_backward's grad_weight for the reference `Conv`,
// model/basic.py:4-7) for widths that are multiples of 4 (>= 48; the 32x32x2 kernel also 24 / 16 / 12 / 8: images side by side in a strip)
// and channel counts that are multiples of 64, with HALF of the
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
// One workgroup owns a 64(co) x 32(ci) x 18 block of dU.  Its 8 waves are 4 pair groups (2 co m-tiles x 1 ci n-tile) x the two
// halves of the xi planes (xh = wave >> 2: xi 3xh .. 3xh+2): 2 x 9 accumulator tiles = 72 VGPRs per wave - with all 18 (ky, xi)
// sets of a 64 x 64 block in one wave (144 VGPRs) hipcc spills accumulators inside the loop.  The workgroup sweeps a range of
// segments (TWO output rows x 12 x-tiles = 96 pixels).  Operands are staged global -> registers -> LDS with the transforms
// applied on the way: V rows live in a 6-slot ring - a segment needs rows r-1 .. r+2 and the next one adds rows r+3, r+4 -
// and the two dM rows are double buffered; one barrier per segment.  The LDS image interleaves the four x-tiles of a k-step
// at 16-float granularity, so the ds_read_b32 fragments are bank-conflict free without padding.  The G^T output transform
// happens in registers (the two xi halves meet through LDS, fixed order) before the partial block leaves, so the split-K slab
// has the direct kernel's [split][9][Cout][Cin] layout and its fixed-order reduce kernel (alpha, PixelShuffle channel
// un-permutation, OIHW store, bias) is shared.  The bias gradient is accumulated on the VALU from the dM_1 fragments
// (dy0+dy1+dy2+dy3).
#include <mutex>
#include "common.h"
#include "launchers.h"

struct Wg4Args {
    const float* x;    // [N][H][W][Cin]
    const float* dy;   // [N][H][W][Cout]   (or shuffled [N][2H][2W][Cout/4] when ps_in)
    float* slab;       // [split][9][Cout][Cin]
    int N, H, W, Cin, Cout;
    int segs_x;        // 12-x-tile strips per image row
    int segs_y;        // row pairs per strip: ceil(H / 2)
    int total_segs;    // N * segs_x * segs_y, ordered (image, strip, row pair): consecutive segments walk down a strip
    int segs_per_split;
    int co_tiles, ci_tiles;
    int ps_in;
    float* bias_part;  // [split][Cout] partial column sums of dy, or null
    int side;          // 32x32x2 kernel only: images per strip.  1, or (rows of W / 4 < 12 x-tiles: W = 24 / 16 / 12 ...) 12 / (W / 4) images
                       // laid SIDE BY SIDE in one 12-x-tile strip (x-tile vt = image vt / (W/4), tile vt % (W/4)); `image` indices of the
                       // segment walk are then groups of `side` images, N the real image count
};

constexpr int G4_NT = 512, G4_TXT = 12, G4_K4 = G4_TXT / 4;
constexpr int G4_VPLANE = G4_K4 * 128, G4_VROW = 6 * G4_VPLANE;       // floats: V row slot [6 xi][3 blocks][4 x-tiles x 32 ci]
constexpr int G4_DPLANE = G4_K4 * 256, G4_DROW = 6 * G4_DPLANE;       // floats: dM row     [6 xi][3 blocks][4 x-tiles x 64 co]
constexpr int G4_RING = 6;


__global__ __launch_bounds__(G4_NT) void conv3x3_wgrad_wino4_kernel(const Wg4Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const vring = lds;                             // [6 slots] V rows
    float* const dmbuf = lds + G4_RING * G4_VROW;         // [2 buffers][2 rows] dM rows

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int xh = wave >> 2, co_half = (wave >> 1) & 1, ci_tile = wave & 1;

    // blockIdx -> (split slice, co tile, ci tile).  Workgroups b and b + 8 share an XCD: hand every XCD a contiguous range of
    // logical ids (channel tiles fastest), so the workgroups that stream the same pixels share an L2.
    int bid = blockIdx.x;
    if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
    const int cit = bid % a.ci_tiles;  bid /= a.ci_tiles;
    const int cot = bid % a.co_tiles;
    const int sp = bid / a.co_tiles;
    const int ci0 = cit * 32, co0 = cot * 64;

    const int seg_begin = sp * a.segs_per_split;
    int seg_end = seg_begin + a.segs_per_split;
    if (seg_end > a.total_segs) seg_end = a.total_segs;

    f32x4 acc[9][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[t][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum[2] = {0.f, 0.f};

    // ---- staging: per segment 2 V rows x 12 x-tiles x 8 ci-groups = 192 items (threads 0..191: six input columns each) and
    //      2 dM rows x 12 x-tiles x 16 co-groups = 384 items (threads 128..511: four output-gradient columns each) ------------
    const bool v_thr = tid < 192, d_thr = tid >= 128;
    const int vi = v_thr ? tid : 0;
    const int v_rr = vi / 96, vt = (vi % 96) >> 3, vc4 = vi & 7;          // which of the 2 new rows, x-tile, 4-channel group
    const int di = d_thr ? tid - 128 : 0;
    const int d_rr = di / 192, dt = (di % 192) >> 4, dc4 = di & 15;
    // position of an item inside a plane: block of 4 x-tiles, 16-float groups interleaved over the 4 x-tiles.  The x-tile slot
    // is rotated by the channel tile (V: by 2 per 16-ci tile, dM: by 1 per 16-co tile): the 16 lanes that write one x-tile's
    // 64 (32) channels then hit 64 different banks instead of the same 16 four (two) times, and a fragment read - one channel
    // tile, x-tiles g = 0..3 - still covers 64 consecutive floats.
    const int v_pos = (vt >> 2) * 128 + (((vc4 >> 2) * 4 + ((vt + 2 * (vc4 >> 2)) & 3)) * 16) + (vc4 & 3) * 4;
    const int d_pos = (dt >> 2) * 256 + (((dc4 >> 2) * 4 + ((dt + (dc4 >> 2)) & 3)) * 16) + (dc4 & 3) * 4;
    const int d_C = a.ps_in ? (a.Cout >> 2) : a.Cout;
    int d_choff;                                          // channel part of a dy address (floats)
    {
        const int pch = co0 + dc4 * 4;
        if (a.ps_in) { const int sub = pch / d_C, cc = pch - sub * d_C; d_choff = ((sub >> 1) * (2 * a.W) + (sub & 1)) * d_C + cc; }
        else d_choff = pch;
    }

    // Loads go through a buffer descriptor over ONE tensor row: a column outside the row is fetched at offset 2^31 and a row
    // outside the image through an empty descriptor - both return zeros, so store_*() transforms what arrived without masking.
    // Only the waves that own items issue them (V: waves 0..2, dM: waves 2..7).
    u32x4 vx[6], dd[4];
    unsigned v_off[6], d_off[4];
    auto set_strip = [&](int xs) {   // per strip, a thread's column offsets (bytes from the start of a row)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int ix = xs * 48 + 4 * vt - 1 + j;
            const bool ok = v_thr && ix >= 0 && ix < a.W;
            v_off[j] = ok ? (unsigned)((ix * a.Cin + ci0 + vc4 * 4) * 4) : 0x80000000u;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ox = xs * 48 + 4 * dt + j;
            const bool ok = d_thr && ox < a.W;
            d_off[j] = ok ? (unsigned)(((a.ps_in ? 2 * ox * d_C : ox * a.Cout) + d_choff) * 4) : 0x80000000u;
        }
    };
    const unsigned x_row_bytes = (unsigned)a.W * a.Cin * 4;
    // a dy row of the shuffled tensor spans the two sub-pixel rows 2oy, 2oy+1: 2 * (2W) * (Cout/4) floats = W * Cout as well
    const unsigned d_row_bytes = (unsigned)a.W * a.Cout * 4;
    // A wave's descriptor must be wave-uniform - or hipcc wraps every buffer load in a waterfall loop (it did: 62 of them).  The
    // two new V rows of a segment are spread over threads 0..95 / 96..191, i.e. they MIX inside wave 1: the V descriptor
    // therefore spans both rows (base = the first one, which may lie outside the image: nothing is fetched through it then) and
    // a lane adds its row's pitch, or 2^31 when its row is outside the image.  dM rows change at a wave boundary (thread 320).
    auto uniform_ptr = [](const float* p) -> const float* {
        const unsigned long long v = (unsigned long long)p;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return (const float*)(((unsigned long long)hi << 32) | lo);
    };
    auto load_v = [&](int img, int iy0) {                  // input rows iy0 (threads 0..95) and iy0 + 1 (96..191); outside the image: zeros
        if (wave < 3) {
            const float* const rowp = a.x + ((long)img * a.H + iy0) * ((long)a.W * a.Cin);
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)uniform_ptr(rowp), 0, 2 * x_row_bytes, 0x00020000);
            const int iy = iy0 + v_rr;
            const bool row_ok = iy >= 0 && iy < a.H;
#pragma unroll
            for (int j = 0; j < 6; ++j)
                vx[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, row_ok ? v_off[j] + (unsigned)v_rr * x_row_bytes : 0x80000000u, 0, 0);
        }
    };
    auto store_v = [&](int slot) {
        if (wave < 3) {
            const f32x4 d0 = __builtin_bit_cast(f32x4, vx[0]), d1 = __builtin_bit_cast(f32x4, vx[1]), d2 = __builtin_bit_cast(f32x4, vx[2]),
                        d3 = __builtin_bit_cast(f32x4, vx[3]), d4 = __builtin_bit_cast(f32x4, vx[4]), d5 = __builtin_bit_cast(f32x4, vx[5]);
            float* p = vring + slot * G4_VROW + v_pos;
            const f32x4 t1 = d4 - 4.0f * d2, t2 = d3 - 4.0f * d1, t3 = d4 - d2, t4 = d3 - d1;
            *(f32x4*)(p) = 4.0f * d0 + (d4 - 5.0f * d2);
            *(f32x4*)(p + G4_VPLANE) = t1 + t2;
            *(f32x4*)(p + 2 * G4_VPLANE) = t1 - t2;
            *(f32x4*)(p + 3 * G4_VPLANE) = t3 + 2.0f * t4;
            *(f32x4*)(p + 4 * G4_VPLANE) = t3 - 2.0f * t4;
            *(f32x4*)(p + 5 * G4_VPLANE) = 4.0f * d1 + (d5 - 5.0f * d3);
        }
    };
    auto load_d = [&](int img, int oy) {                   // output-gradient row oy (>= H: zeros)
        if (wave >= 2) {
            const bool row_ok = oy < a.H;
            const int ry = row_ok ? oy : 0;
            const float* const rowp = a.ps_in ? a.dy + ((size_t)img * (2 * a.H) + 2 * ry) * (2 * a.W) * d_C
                                              : a.dy + ((size_t)img * a.H + ry) * a.W * a.Cout;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)uniform_ptr(rowp), 0,
                                                                                __builtin_amdgcn_readfirstlane(row_ok ? d_row_bytes : 0u), 0x00020000);
#pragma unroll
            for (int j = 0; j < 4; ++j) dd[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, d_off[j], 0, 0);
        }
    };
    auto store_d = [&](int buf) {
        if (wave >= 2) {
            const f32x4 g0 = __builtin_bit_cast(f32x4, dd[0]), g1 = __builtin_bit_cast(f32x4, dd[1]), g2 = __builtin_bit_cast(f32x4, dd[2]),
                        g3 = __builtin_bit_cast(f32x4, dd[3]);
            float* p = dmbuf + (buf * 2 + d_rr) * G4_DROW + d_pos;
            const f32x4 e02 = g0 + g2, e13 = g1 + g3, f02 = g0 + 4.0f * g2, f13 = g1 + 4.0f * g3;
            *(f32x4*)(p) = g0;
            *(f32x4*)(p + G4_DPLANE) = e02 + e13;
            *(f32x4*)(p + 2 * G4_DPLANE) = e02 - e13;
            *(f32x4*)(p + 3 * G4_DPLANE) = f02 + 2.0f * f13;
            *(f32x4*)(p + 4 * G4_DPLANE) = f02 - 2.0f * f13;
            *(f32x4*)(p + 5 * G4_DPLANE) = g3;
        }
    };
    auto seg_coords = [&](int seg, int& img, int& xs, int& row) {
        const int strip = seg / a.segs_y;
        row = 2 * (seg - strip * a.segs_y);
        img = strip / a.segs_x;
        xs = strip - img * a.segs_x;
    };
    auto stage_strip_start = [&](int img, int row, int buf) {   // halo rows row-1 .. row+2 -> slots 0 .. 3; dM(row, row+1) -> buf
        load_v(img, row - 1); store_v(v_rr);
        load_v(img, row + 1); store_v(2 + v_rr);
        load_d(img, row + d_rr); store_d(buf);
    };

    // ---- fragment addresses (floats): lane (r, g) reads x-tile 4k + g, channel 16*tile + r ------------------------------
    const int b_lane = (ci_tile * 4 + ((g + 2 * ci_tile) & 3)) * 16 + r + xh * 3 * G4_VPLANE;
    const int a_lane0 = (co_half * 2 * 4 + ((g + co_half * 2) & 3)) * 16 + r + xh * 3 * G4_DPLANE;        // m-tile 2 co_half
    const int a_lane1 = ((co_half * 2 + 1) * 4 + ((g + co_half * 2 + 1) & 3)) * 16 + r + xh * 3 * G4_DPLANE;   // m-tile 2 co_half + 1

    if (seg_begin >= seg_end) return;                       // (never: the planner hands every workgroup at least one segment)
    int img, xs, row;
    seg_coords(seg_begin, img, xs, row);
    set_strip(xs);
    stage_strip_start(img, row, 0);
    // Staging runs a full segment ahead of its ds_writes: the loads for segment s+1's new rows are issued at the store point
    // of segment s-1 (or right after a strip start) and land while segment s-1 / s computes; with the loads issued at the top
    // of segment s they were waited for a third of a segment later - under load an L2 / MALL round trip is longer than that.
    if (seg_begin + 1 < seg_end && row + 2 < a.H) { load_v(img, row + 3); load_d(img, row + 2 + d_rr); }
    __syncthreads();
    int base = 0;                                           // ring slot of the segment's top halo row (row - 1)
    // One common store point for all waves, two thirds into the segment, with the next loads issued right behind it.  Measured
    // alternatives: different store points for the two waves of a SIMD 1.4 % slower; the loads spread over three k-steps instead
    // of one burst 12 % slower (those issued two k-steps before the store have not landed - a loaded L2 round trip is > 1 us).
    constexpr int store_step = 4;

#pragma unroll 1
    for (int seg = seg_begin; seg < seg_end; ++seg) {
        const int par = (seg - seg_begin) & 1;
        const bool more = seg + 1 < seg_end;
        const bool cont = more && row + 2 < a.H;            // the next segment is the next row pair of the same strip
        const bool cont2 = cont && seg + 2 < seg_end && row + 4 < a.H;   // ... and so is the one after it

        const float* const db = dmbuf + (par * 2) * G4_DROW + a_lane0;
        const float* const db1 = dmbuf + (par * 2) * G4_DROW + a_lane1;
        const float* vb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int sl = base + q; if (sl >= G4_RING) sl -= G4_RING;
            vb[q] = vring + sl * G4_VROW + b_lane;
        }
        // step = (row of the pair rr, k-step k4): 6 A fragments (3 xi x 2 m-tiles) + 9 B fragments (3 ky x 3 xi), 18 MFMAs
        float av0[6], bv0[9], av1[6], bv1[9];
#define G4_READ(AV, BV, STEP)                                                                            \
        {                                                                                                \
            const int rr_ = (STEP) / G4_K4, k4_ = (STEP) % G4_K4;                                         \
            _Pragma("unroll") for (int xl = 0; xl < 3; ++xl) {                                           \
                AV[xl * 2 + 0] = db[rr_ * G4_DROW + xl * G4_DPLANE + k4_ * 256];                         \
                AV[xl * 2 + 1] = db1[rr_ * G4_DROW + xl * G4_DPLANE + k4_ * 256];                        \
                _Pragma("unroll") for (int ky = 0; ky < 3; ++ky)                                         \
                    BV[ky * 3 + xl] = vb[rr_ + ky][xl * G4_VPLANE + k4_ * 128];                          \
            }                                                                                            \
        }
#define G4_MFMA(AV, BV)                                                                                  \
        _Pragma("unroll") for (int ky = 0; ky < 3; ++ky)                                                 \
            _Pragma("unroll") for (int xl = 0; xl < 3; ++xl)                                             \
                _Pragma("unroll") for (int i = 0; i < 2; ++i)                                            \
                    acc[ky * 3 + xl][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV[xl * 2 + i], BV[ky * 3 + xl], acc[ky * 3 + xl][i], 0, 0, 0); \
        if (xh == 0) { _Pragma("unroll") for (int i = 0; i < 2; ++i) bsum[i] += AV[2 + i]; }   /* dM_1 = dy0+dy1+dy2+dy3 */
        G4_READ(av0, bv0, 0)
#pragma unroll
        for (int stp = 0; stp < 2 * G4_K4; stp += 2) {
            G4_READ(av1, bv1, stp + 1)
            __builtin_amdgcn_sched_barrier(0);
            G4_MFMA(av0, bv0)
            __builtin_amdgcn_sched_barrier(0);
            if (stp + 2 < 2 * G4_K4) G4_READ(av0, bv0, stp + 2)
            if (stp == store_step) {
                // The staging stores go to LDS that nobody reads in this segment (the two free ring slots, the other dM
                // buffer); right behind them the loads for the segment after next reuse the staging registers.
                __builtin_amdgcn_sched_barrier(0);
                if (cont) {
                    int sl = base + 4 + v_rr; if (sl >= G4_RING) sl -= G4_RING;
                    { store_v(sl); store_d(par ^ 1); }
                    if (cont2) { load_v(img, row + 5); load_d(img, row + 4 + d_rr); }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            G4_MFMA(av1, bv1)
            __builtin_amdgcn_sched_barrier(0);
        }
#undef G4_READ
#undef G4_MFMA
        if (cont) {                                         // rows +3, +4 went to the slots of rows -1, 0, which the next segment drops
            __syncthreads();
            base += 2; if (base >= G4_RING) base -= G4_RING;
            row += 2;
        } else if (more) {                                  // new strip / image: its four halo rows are staged from scratch
            __syncthreads();
            seg_coords(seg + 1, img, xs, row);
            set_strip(xs);
            stage_strip_start(img, row, par ^ 1);
            if (seg + 2 < seg_end && row + 2 < a.H) { load_v(img, row + 3); load_d(img, row + 2 + d_rr); }
            __syncthreads();
            base = 0;
        }
    }
    __syncthreads();

    if (a.bias_part && cit == 0) {   // combine the 4 k-slot lane groups through LDS (the staging buffers are free now), fixed order
        float* red = lds;
        if (ci_tile == 0 && xh == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i) red[g * 64 + (co_half * 2 + i) * 16 + r] = bsum[i];
        }
        __syncthreads();
        if (tid < 64 && co0 + tid < a.Cout)
            a.bias_part[(size_t)sp * a.Cout + co0 + tid] = ((red[tid] + red[64 + tid]) + red[128 + tid]) + red[192 + tid];
        __syncthreads();
    }
    // G^T in registers: each xi half contributes a partial dw; the halves meet in LDS [tap 9][co 64][ci 32] (xh = 0 writes,
    // xh = 1 adds and stores).  slab[sp][ky*3+kx][co][ci]: D tile row = co (= (lane>>4)*4 + reg), col = ci (= lane&15).
    float* const ob = lds;
    float* const out = a.slab + (size_t)sp * 9 * a.Cout * a.Cin;
    const unsigned tap = (unsigned)a.Cout * a.Cin;
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
        if (xh == ph) {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const f32x4 u0 = acc[ky * 3 + 0][i], u1 = acc[ky * 3 + 1][i], u2 = acc[ky * 3 + 2][i];
                    f32x4 w0, w1, w2;
                    if (ph == 0) {       // U0, U1, U2
                        const f32x4 s12 = u1 + u2, d12 = u1 - u2;
                        w0 = 0.25f * u0 - (1.0f / 6.0f) * s12;
                        w1 = (-1.0f / 6.0f) * d12;
                        w2 = (-1.0f / 6.0f) * s12;
                    } else {             // U3, U4, U5
                        const f32x4 s34 = u0 + u1, d34 = u0 - u1;
                        w0 = (1.0f / 24.0f) * s34;
                        w1 = (1.0f / 12.0f) * d34;
                        w2 = (1.0f / 6.0f) * s34 + u2;
                    }
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const int col = (co_half * 2 + i) * 16 + g * 4 + jj, cil = ci_tile * 16 + r;
                        float* o = ob + ((ky * 3) * 64 + col) * 32 + cil;
                        if (ph == 0) { o[0] = w0[jj]; o[64 * 32] = w1[jj]; o[2 * 64 * 32] = w2[jj]; }
                        else {
                            const unsigned go = ((unsigned)(ky * 3) * a.Cout + co0 + col) * a.Cin + ci0 + cil;
                            out[go] = o[0] + w0[jj]; out[go + tap] = o[64 * 32] + w1[jj]; out[go + 2 * tap] = o[2 * 64 * 32] + w2[jj];
                        }
                    }
                }
        }
        if (ph == 0) __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Round-3 variant on v_mfma_f32_32x32x2_f32 (PESR_WGRAD_WINO4X): same transform, same 64(co) x 32(ci) x 18 block, same segment
// sweep, ring, split-K slab and reduce.  What changes is who owns what: TWELVE waves = 2 co halves (32 channels each) x the 6 xi
// planes; a wave keeps dU_xi[ky = 0..2] of its 32 x 32 tile (3 accumulator tiles = 48 VGPRs) and, per k-step (TWO x-tiles),
// reads 2 dM fragments (the segment's two rows) and 4 V fragments (input rows r-1 .. r+2) for SIX 64-cycle MFMAs - one
// ds_read_b32 per MFMA where the 16x16x4 form above needs 15 per 18 32-cycle MFMAs, i.e. 0.4 of the LDS read bytes per flop.
// The LDS planes are plain [x-tile][channel] arrays: a fragment read touches 2 x 32 consecutive floats, an item's staging store
// 8 lanes x 16 bytes contiguous - conflict-free without any swizzle.  Three waves per SIMD hide what two could not.
// G^T needs all six xi of a tap: the waves park their accumulators in LDS (147 KB, the staging buffers are free by then) and all
// 768 threads finish the nine taps of the block in the same order of additions as the kernel above.
// ---------------------------------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int X4_NT = 768;
constexpr int X4_VPLANE = G4_TXT * 32, X4_VROW = 6 * X4_VPLANE;       // floats: V row slot [6 xi][12 x-tiles][32 ci]
constexpr int X4_DPLANE = G4_TXT * 64, X4_DROW = 6 * X4_DPLANE;       // floats: dM row     [6 xi][12 x-tiles][64 co]
constexpr int X4_RING = 8;                                            // V row slots: four in use, four for the segment behind (a strip start needs all four)

// ---------------------------------------------------------------------------------------------------------------------------
// Round 4, NEST = true (the default): the transform NESTED in y - F(2,3) along y on top of F(4,3) along x - at no cost in staging.
// A segment is two output rows; per k-step a wave already holds, for its xi plane, the x-transformed gradients D0, D1 of the two
// rows and the x-transformed inputs X0 .. X3 of the four input rows they touch, and the 1-D form spends SIX MFMAs on
//     dU[ky] += D0 * X[ky] + D1 * X[ky + 1],  ky = 0, 1, 2
// - a 3-tap correlation of 2 against 4 values, i.e. exactly the F(2,3) weight-gradient problem.  Its four products
//     P0 += D0 * (X0 - X2)    P1 += (D0 + D1) * (X1 + X2)    P2 += (D0 - D1) * (X2 - X1)    P3 += D1 * (X3 - X1)
// give dU[0] = P0 + (P1 + P2) / 2, dU[1] = (P1 - P2) / 2, dU[2] = (P1 + P2) / 2 + P3 (G2^T, applied once, in registers, in front of
// the G4^T epilogue): FOUR MFMAs per k-step for the same fragment reads, the same staging, ring and LDS image - five VALU adds per
// k-step buy a third of the matrix work (24 products per 2 x 4 output pixels: 1/3 of the direct form's 72).  One more accumulator
// tile per wave (64 VGPRs).  Numerics (scripts/wino2d_wgrad_study.py, fp32 emulation vs fp64 at 16 x 48 x 48 pixels): 2.0 - 2.3e-6
// of the gradient's maximum against 2.1 - 3.5e-6 for the 1-D form - the F(2,3) matrices are 0, +-1, 1/2.
// ---------------------------------------------------------------------------------------------------------------------------
template <bool NEST>
__global__ __launch_bounds__(X4_NT) void conv3x3_wgrad_wino4x_kernel(const Wg4Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const vring = lds;                             // [8 slots] V rows
    float* const dmbuf = lds + X4_RING * X4_VROW;         // [2 buffers][2 rows] dM rows
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c32 = lane & 31, ks = lane >> 5;            // fragment lane: channel, k-slot (x-tile 2q + ks of k-step q)
    const int cot2 = wave / 6, xi = wave - cot2 * 6;

    int bid = blockIdx.x;
    if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
    const int cit = bid % a.ci_tiles;  bid /= a.ci_tiles;
    const int cot = bid % a.co_tiles;
    const int sp = bid / a.co_tiles;
    const int ci0 = cit * 32, co0 = cot * 64;

    const int seg_begin = sp * a.segs_per_split;
    int seg_end = seg_begin + a.segs_per_split;
    if (seg_end > a.total_segs) seg_end = a.total_segs;

    constexpr int NACC = NEST ? 4 : 3;
    f32x16 acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
    float bsum = 0.f;

    // ---- staging roles: threads 0..191 (waves 0..2) the 192 V items (2 rows x 12 x-tiles x 8 ci groups, six input columns each),
    //      threads 192..575 (waves 3..8) the 384 dM items (2 rows x 12 x-tiles x 16 co groups, four columns each); one register set.
    //      Waves 9..11 take the V role for the SECOND pair of halo rows whenever a segment does not continue the previous one's
    //      strip (the workgroup's first segment, a new strip, a new image): its four halo rows then arrive in one load round.
    //      A thread has ONE role, so the per-thread staging constants live in role-neutral registers (the register budget of three
    //      waves per SIMD is 168): s_rr = which of the item's two rows, s_pos = float position inside an LDS plane, s_cb = byte offset
    //      of the item's channel group (and, side > 1, of its image inside the strip's group), s_x = first column (strip-relative
    //      column + 48 * strip; pushed out of range for an image that does not exist), s_ps = byte pitch of a pixel.
    // Roles by SIMD (waves w, w + 4, w + 8 share one): every VALU instruction costs the fp32 MFMA pipe of ITS SIMD ~3.7 cycles
    // (scripts/mfma_valu_probe.hip), and the segment's barrier waits for the SIMD with the most staging work - so V (48 VALU per
    // item) goes to waves 0, 1, 2, dM (32) to waves 4, 5, 6 and - three to the SIMD that has no V wave - 3, 7, 11; waves 8, 9, 10
    // (one on each V SIMD) stage nothing in a strip and take the second pair of halo rows at its start.
    const bool v_hi = wave >= 8 && wave < 11, v_role = wave < 3 || v_hi;
    const int d_wave = wave == 11 ? 5 : wave - 3;             // dM waves 3, 4, 5, 6, 7, 11 -> item blocks 0 .. 5
    const int XTW = a.W >> 2;
    const int d_C = a.ps_in ? (a.Cout >> 2) : a.Cout;
    int s_rr, s_pos;
    unsigned s_cb, s_ps;
    int s_x;
    {
        const int item = v_role ? (tid < 192 ? tid : tid - 512) : d_wave * 64 + lane;
        if (v_role) {
            const int vt = (item % 96) >> 3, vc4 = item & 7;
            const int sub = a.side > 1 ? vt / XTW : 0;
            s_rr = item / 96;
            s_pos = vt * 32 + vc4 * 4;
            s_x = 4 * (vt - sub * XTW) - 1;
            s_ps = (unsigned)a.Cin * 4;
            s_cb = (unsigned)(((sub * a.H) * a.W) * a.Cin + ci0 + vc4 * 4) * 4;
        } else {
            const int dt = (item % 192) >> 4, dc4 = item & 15;
            const int sub0 = a.side > 1 ? dt / XTW : 0;
            s_rr = item / 192;
            s_pos = dt * 64 + dc4 * 4;
            s_x = 4 * (dt - sub0 * XTW);
            const int pch = co0 + dc4 * 4;
            if (a.ps_in) {
                const int sub = pch / d_C, cc = pch - sub * d_C;
                s_cb = (unsigned)(((sub >> 1) * (2 * a.W) + (sub & 1)) * d_C + cc) * 4;
                s_ps = (unsigned)(2 * d_C) * 4;
            } else {
                s_cb = (unsigned)((sub0 * a.H * a.W) * a.Cout + pch) * 4;
                s_ps = (unsigned)a.Cout * 4;
            }
        }
        s_ps = __builtin_amdgcn_readfirstlane(s_ps);         // (the same for every lane of a wave)
    }
    u32x4 st[6];                  // V thread: six input columns; dM thread: four gradient columns
    unsigned off[6];              // their byte offsets from the start of the item's FIRST row (V: the lane's row pitch folded in); 2^31 = outside
    // side > 1: x-tile t of the strip is tile t % XTW of image (group * side + t / XTW); its columns never leave that image (the
    // neighbour's pixels are NOT its halo: out-of-image columns read zeros as at a real image border)
    const unsigned x_row_bytes = (unsigned)a.W * a.Cin * 4;
    const unsigned d_row_bytes = (unsigned)a.W * a.Cout * 4;
    int s_xs = 0;                                             // the strip s_x currently points into
    auto set_strip = [&](int xs, int grp) {                   // (once per strip: the only place the offsets cost VALU instructions)
        s_x += (xs - s_xs) * 48;
        s_xs = xs;
        int x0 = s_x;
        if (a.side > 1) {                                     // (one strip per row there: only the image group changes)
            const int t = v_role ? ((tid < 192 ? tid : tid - 512) % 96) >> 3 : ((d_wave * 64 + lane) % 192) >> 4;
            if (grp * a.side + t / XTW >= a.N) x0 += 0x100000;      // an image of the last group that does not exist: its columns leave the row
        }
        const unsigned rowterm = v_role ? s_cb + (unsigned)s_rr * x_row_bytes : s_cb;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const unsigned x = (unsigned)(x0 + j);
            off[j] = (x < (unsigned)a.W && (v_role || j < 4)) ? rowterm + x * s_ps : 0x80000000u;
        }
    };
    const unsigned x_side_bytes = (unsigned)(a.side - 1) * a.H * x_row_bytes;     // the descriptors reach over the strip's other images
    const unsigned d_side_bytes = (unsigned)(a.side - 1) * a.H * d_row_bytes;
    auto uniform_ptr = [](const float* p) -> const float* {
        const unsigned long long v = (unsigned long long)p;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return (const float*)(((unsigned long long)hi << 32) | lo);
    };
    // A segment of the walk: image (or image group), strip, top output row; cont = it continues the previous segment's strip (its
    // rows row-1, row are already in the ring), valid = it exists in this workgroup's slice.
    struct Seg { int img, xs, row; bool cont, valid; };
    auto seg_after = [&](const Seg& c, int index) {           // the segment behind c; index = its number in the walk
        Seg n;
        n.valid = index < seg_end;
        if (c.row + 2 < a.H) { n.img = c.img; n.xs = c.xs; n.row = c.row + 2; n.cont = true; }
        else { n.cont = false; n.row = 0; n.xs = c.xs + 1; n.img = c.img; if (n.xs == a.segs_x) { n.xs = 0; n.img = c.img + 1; } }
        return n;
    };
    // The staging loads of segment g, unconditional (a segment that does not exist is fetched through zero-size descriptors: nothing
    // moves): a continuing segment adds input rows row+1, row+2 (waves 0..2), a segment that starts a strip needs row-1 .. row+2
    // (waves 0..2: row-1, row; waves 8..10: row+1, row+2); the dM waves its two gradient rows.  Loads go through a buffer
    // descriptor over the item's rows: a column outside the row sits at offset 2^31, a row behind the image's last one is cut off
    // by the descriptor's size, a segment that does not exist gets size 0 - all return zeros, and store_stage() transforms what
    // arrived without masking.  Inside a strip (the hot path) this costs NO vector instruction besides the loads: base and size are
    // scalar, the lane offsets are the strip's constants (every VALU instruction takes ~3.7 cycles from the fp32 MFMA pipe of its
    // SIMD).  Only a strip's first segment - whose halo row above the image needs a per-lane test - and the side-by-side form
    // (side > 1: the descriptor spans several images, its size cannot cut off one image's last row) select per lane.
    // (descriptors wave-uniform: the V descriptor spans the two rows that mix inside a wave, the dM rows change at a wave boundary)
    // load_prep(g) makes the segment's descriptor (scalar), load_step(j) issues column j: the main loop puts ONE load in front of
    // each of six consecutive MFMAs - as a burst, the ~42 wave-level loads of a segment queued up behind each other in the CU's
    // one address unit and the issuing waves (and their MFMAs) stood 400 - 1100 cycles.
    __amdgpu_buffer_rsrc_t ld_rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, 0, 0x00020000);
    int ld_on = 0;                                            // 1: this wave loads for the segment (scalar)
    bool ld_lane_ok = true;                                   // this lane's row is inside the image (false only on the per-lane path below)
    auto load_prep = [&](const Seg& g) {
        if (v_role) {
            const int v_iy0 = (g.cont || v_hi) ? g.row + 1 : g.row - 1;
            const bool on = g.valid && !(v_hi && g.cont);     // (waves 8..10 inside a strip, segments that do not exist: no loads at all)
            const float* const rowp = a.x + ((long)g.img * a.side * a.H + v_iy0) * ((long)a.W * a.Cin);
            const int iy = v_iy0 + s_rr;
            unsigned bytes;
            if (g.cont && a.side == 1) {                      // inside a strip: the descriptor's size cuts off what lies behind the image
                const int rows = a.H - v_iy0 > 2 ? 2 : (a.H - v_iy0 < 0 ? 0 : a.H - v_iy0);
                bytes = (unsigned)rows * x_row_bytes;
                ld_lane_ok = true;
            } else {                                          // a strip's first segment (its halo row ABOVE the image), the side-by-side form
                bytes = x_side_bytes + 2 * x_row_bytes;
                ld_lane_ok = iy >= 0 && iy < a.H;
            }
            ld_on = __builtin_amdgcn_readfirstlane(on ? 1 : 0);
            ld_rs = __builtin_amdgcn_make_buffer_rsrc((void*)uniform_ptr(rowp), 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
        } else {
            const int oy = g.row + __builtin_amdgcn_readfirstlane(s_rr);      // (dM rows change at a wave boundary)
            const bool row_ok = g.valid && oy < a.H;
            const int ry = row_ok ? oy : 0;
            const float* const rowp = a.ps_in ? a.dy + ((size_t)g.img * (2 * a.H) + 2 * ry) * (2 * a.W) * d_C
                                              : a.dy + ((size_t)g.img * a.side * a.H + ry) * a.W * a.Cout;
            ld_lane_ok = true;
            ld_on = __builtin_amdgcn_readfirstlane(g.valid ? 1 : 0);
            ld_rs = __builtin_amdgcn_make_buffer_rsrc((void*)uniform_ptr(rowp), 0,
                                                      __builtin_amdgcn_readfirstlane(row_ok ? d_side_bytes + d_row_bytes : 0u), 0x00020000);
        }
    };
    auto load_step = [&](const int j) {                       // (ONE load site per column: a second definition of st[] costs copies behind a vmcnt(0))
        if (ld_on && (v_role || j < 4)) st[j] = __builtin_amdgcn_raw_buffer_load_b128(ld_rs, ld_lane_ok ? off[j] : 0x80000000u, 0, 0);
    };
#define X4_LOAD_ALL(G) { load_prep(G); _Pragma("unroll") for (int j_ = 0; j_ < 6; ++j_) load_step(j_); }      /* all columns at once (the workgroup's first two segments) */
    // ... and their transform + LDS stores: V rows -> ring slots v_slot0 (+ 2 for waves 8..10) + row (mod 8), dM rows -> buffer d_buf.
    // In SIX steps, one LDS plane each (step n: the few VALU instructions plane n needs, then its ds_write_b128), so that the main
    // loop can put one step in front of each of six consecutive MFMAs: as one block per wave (round 4) the nine staging waves ran
    // ~850 cycles of transform + stores and 400 - 1100 cycles of queued-up buffer loads all at the same point of the segment, and the
    // matrix pipes of their SIMDs starved meanwhile (in-kernel stamps, profiles/r05_wgrad_notes.txt).
    f32x4 tA, tB;                                             // transform terms that live from one step to the next
    // (signed constants: written as d4 - 5.0f * d2 hipcc negates d2 with a v_xor per register in front of each v_pk_fma)
    const f32x4 c_m5 = {-5.0f, -5.0f, -5.0f, -5.0f}, c_m4 = {-4.0f, -4.0f, -4.0f, -4.0f}, c_p4 = {4.0f, 4.0f, 4.0f, 4.0f};
    float* sp_ = nullptr;                                     // the item's LDS address for the segment being stored
    auto stage_step = [&](const int n, int v_slot0, int d_buf, bool four) {
        if (v_role) {
            if (!v_hi || four) {
                const f32x4 d0 = __builtin_bit_cast(f32x4, st[0]), d1 = __builtin_bit_cast(f32x4, st[1]), d2 = __builtin_bit_cast(f32x4, st[2]),
                            d3 = __builtin_bit_cast(f32x4, st[3]), d4 = __builtin_bit_cast(f32x4, st[4]), d5 = __builtin_bit_cast(f32x4, st[5]);
                if (n == 0) {
                    sp_ = vring + ((v_slot0 + (v_hi ? 2 : 0) + s_rr) & (X4_RING - 1)) * X4_VROW + s_pos;
                    *(f32x4*)(sp_) = __builtin_elementwise_fma(c_p4, d0, __builtin_elementwise_fma(c_m5, d2, d4));
                } else if (n == 1) {
                    tA = __builtin_elementwise_fma(c_m4, d2, d4); tB = __builtin_elementwise_fma(c_m4, d1, d3);
                    *(f32x4*)(sp_ + X4_VPLANE) = tA + tB;
                } else if (n == 2) {
                    *(f32x4*)(sp_ + 2 * X4_VPLANE) = tA - tB;
                } else if (n == 3) {
                    tA = d4 - d2; tB = 2.0f * (d3 - d1);
                    *(f32x4*)(sp_ + 3 * X4_VPLANE) = tA + tB;
                } else if (n == 4) {
                    *(f32x4*)(sp_ + 4 * X4_VPLANE) = tA - tB;
                } else {
                    *(f32x4*)(sp_ + 5 * X4_VPLANE) = __builtin_elementwise_fma(c_p4, d1, __builtin_elementwise_fma(c_m5, d3, d5));
                }
            }
        } else {
            const f32x4 g0 = __builtin_bit_cast(f32x4, st[0]), g1 = __builtin_bit_cast(f32x4, st[1]), g2 = __builtin_bit_cast(f32x4, st[2]),
                        g3 = __builtin_bit_cast(f32x4, st[3]);
            if (n == 0) {
                sp_ = dmbuf + (d_buf * 2 + s_rr) * X4_DROW + s_pos;
                *(f32x4*)(sp_) = g0;
            } else if (n == 1) {
                tA = g0 + g2; tB = g1 + g3;
                *(f32x4*)(sp_ + X4_DPLANE) = tA + tB;
            } else if (n == 2) {
                *(f32x4*)(sp_ + 2 * X4_DPLANE) = tA - tB;
            } else if (n == 3) {
                tA = g0 + 4.0f * g2; tB = 2.0f * (g1 + 4.0f * g3);
                *(f32x4*)(sp_ + 3 * X4_DPLANE) = tA + tB;
            } else if (n == 4) {
                *(f32x4*)(sp_ + 4 * X4_DPLANE) = tA - tB;
            } else {
                *(f32x4*)(sp_ + 5 * X4_DPLANE) = g3;
            }
        }
    };
    auto store_stage = [&](int v_slot0, int d_buf, bool four) {      // all six at once (the workgroup's first segment)
#pragma unroll
        for (int n = 0; n < 6; ++n) stage_step(n, v_slot0, d_buf, four);
    };

    // ---- fragment addresses (floats): lane (c32, ks) reads x-tile 2q + ks, channel c32 of its tile ---------------------------
    const int b_lane = xi * X4_VPLANE + ks * 32 + c32;
    const int a_lane = xi * X4_DPLANE + ks * 64 + cot2 * 32 + c32;

    if (seg_begin >= seg_end) return;
    Seg cur;
    {
        const int strip = seg_begin / a.segs_y;
        cur.row = 2 * (seg_begin - strip * a.segs_y);
        cur.img = strip / a.segs_x;
        cur.xs = strip - cur.img * a.segs_x;
        cur.cont = false; cur.valid = true;
    }
    Seg n2 = seg_after(cur, seg_begin + 1);
    bool n1_valid = n2.valid, n1_cont = n2.cont;              // the segment behind the one being computed (n2: the one after that)
    // The walk is ONE software pipeline over all segments of the slice, strip changes included (round 5): at its store point a
    // segment stores what the segment behind it needs (two new rows, or the four halo rows of a new strip: the ring has EIGHT
    // slots, four in use, four free) and loads for the one after that.  Until round 4 a strip change was staged synchronously
    // between two barriers (3.8 us per change at the G-body shape).
    set_strip(cur.xs, cur.img);
    X4_LOAD_ALL(cur)
    store_stage(0, 0, true);
    if (n2.valid && !n2.cont) set_strip(n2.xs, n2.img);
    X4_LOAD_ALL(n2)
    n2 = seg_after(n2, seg_begin + 2);
    __syncthreads();
    int base = 0, par = 0;                                  // ring slot of the segment's top halo row (row - 1); its dM buffer
    constexpr int KQ = G4_TXT / 2;                          // k-steps per row
    // Static priority for the second-dispatched half (waves 6..11), set once: the two halves run the same program in lockstep
    // behind one barrier per segment, and the younger half loses every arbitration; raised, it pulls ahead and the halves'
    // LDS read bursts and MFMA blocks de-phase (MI355X_MICROARCH.md, "Two waves per SIMD", item 4): 209.6 -> 206.2 us; three
    // levels (w, w + 4, w + 8 share a SIMD) 206.9, the first half raised instead 208.8 (profiles/r03_wgrad_variants.txt).

    float fa0[2], fb0[4], fa1[2], fb1[4];
#define X4_FENCE() __builtin_amdgcn_sched_barrier(0)
#define X4_READ(FA, FB, DB, VB, Q)      /* three ds_read2st64_b32: the dM rows, V rows 0 / 1, V rows 2 / 3 */ \
        {                                                                                        \
            FA[0] = DB[(Q) * 128]; FA[1] = DB[X4_DROW + (Q) * 128];                              \
            FB[0] = VB[0][(Q) * 64]; FB[1] = VB[0][X4_VROW + (Q) * 64];                          \
            FB[2] = VB[1][(Q) * 64]; FB[3] = VB[1][X4_VROW + (Q) * 64];                          \
        }
    // One k-step: the y-nesting's VALU work first, then its MFMAs with one hook in front of each (HB < 0: none).  hook(n), n = 0 .. 11
    // over three consecutive k-steps: staging steps 0 .. 5 (one LDS plane each), then the six loads of the segment after next.
#define X4_KSTEP(FA, FB, HB)                                                                     \
        if (NEST) {                                                                              \
            const float ds_ = FA[0] + FA[1], dd_ = FA[0] - FA[1];                                \
            const float x0_ = FB[0] - FB[2], x1_ = FB[1] + FB[2], x2_ = FB[2] - FB[1], x3_ = FB[3] - FB[1]; \
            if (xi == 1) { asm volatile("" : "+v"(bsum)); bsum += ds_; }   /* dM_1 = dy0+dy1+dy2+dy3, both rows; the empty asm keeps this a scalar BRANCH: if-converted (v_add + v_cndmask on all twelve waves) it cost every wave two VALU instructions per k-step */ \
            X4_FENCE(); if ((HB) >= 0) hook((HB) + 0); X4_FENCE();                               \
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[0], x0_, acc[0], 0, 0, 0);          \
            X4_FENCE(); if ((HB) >= 0) hook((HB) + 1); X4_FENCE();                               \
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ds_, x1_, acc[1], 0, 0, 0);            \
            X4_FENCE(); if ((HB) >= 0) hook((HB) + 2); X4_FENCE();                               \
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(dd_, x2_, acc[2], 0, 0, 0);            \
            X4_FENCE(); if ((HB) >= 0) hook((HB) + 3); X4_FENCE();                               \
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[1], x3_, acc[3], 0, 0, 0);          \
        } else {                                                                                 \
            if ((HB) >= 0) { hook((HB) + 0); hook((HB) + 1); hook((HB) + 2); hook((HB) + 3); }   \
            _Pragma("unroll") for (int ky = 0; ky < 3; ++ky) {                                   \
                acc[ky] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[0], FB[ky], acc[ky], 0, 0, 0); \
                acc[ky] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[1], FB[ky + 1], acc[ky], 0, 0, 0); \
            }                                                                                    \
            if (xi == 1) { asm volatile("" : "+v"(bsum)); bsum += FA[0] + FA[1]; }   /* dM_1 = dy0+dy1+dy2+dy3 */ \
        }
    const float* db = dmbuf + a_lane;
    const float* vb[2] = {vring + b_lane, vring + 2 * X4_VROW + b_lane};     // rows (base, base + 1) and (base + 2, base + 3): the base is even, a pair never wraps
    X4_READ(fa0, fb0, db, vb, 0)
    int seg = seg_begin;
#pragma unroll 1
    for (;;) {
        // the segment behind this one: ring base (two slots on inside a strip, four at a strip start) and dM buffer
        const int nbase = (base + (n1_cont ? 2 : 4)) & (X4_RING - 1);
        // The staging of the segment BEHIND this one (stores; they go to LDS that nobody reads in this segment: the free ring
        // slots, the other dM buffer) and the loads of the one after that, one step per MFMA of k-steps 2, 3 and 4.
        auto hook = [&](const int n) {
            if (n < 6) { if (n1_valid) stage_step(n, base + 4, par ^ 1, !n1_cont); }
            if (n == 5) {
                // ... and right behind the last store all loads of the segment after next: woven one per MFMA they were issued up to
                // 1.5 k-steps later, and that much less lead over their own store point cost 3 us per launch.
                if (n2.valid && !n2.cont) set_strip(n2.xs, n2.img);
                // (every staging register has been stored by now, or belongs to a segment that does not exist: told to the compiler
                // once, or it waits vmcnt(0) in front of EACH of the six loads - for the load before it)
                __builtin_amdgcn_s_waitcnt(0x0F70);
                load_prep(n2);
#pragma unroll
                for (int t = 0; t < 6; ++t) load_step(t);
            }
        };
        X4_READ(fa1, fb1, db, vb, 1)
        X4_FENCE();
        X4_KSTEP(fa0, fb0, -1)
        X4_FENCE();
        X4_READ(fa0, fb0, db, vb, 2)
        X4_FENCE();
        X4_KSTEP(fa1, fb1, -1)
        X4_FENCE();
        X4_READ(fa1, fb1, db, vb, 3)
        X4_FENCE();
        X4_KSTEP(fa0, fb0, 0)
        X4_FENCE();
        X4_READ(fa0, fb0, db, vb, 4)
        X4_FENCE();
        X4_KSTEP(fa1, fb1, 4)
        X4_FENCE();
        X4_READ(fa1, fb1, db, vb, 5)
        X4_FENCE();
        X4_KSTEP(fa0, fb0, -1)
        X4_FENCE();
        // The ONE barrier of a segment sits in front of its last k-step: every wave has its last fragments of this segment in
        // registers (so the NEXT segment's stores may overwrite this segment's rows), every staging store of this segment is
        // done (so the next segment's first fragments can be read here, under that k-step's MFMAs - until round 4 they were read
        // behind a barrier at the segment's end, with the matrix pipe idle for their latency).
        // (behind the slice's last segment the barrier and the read run once more, on rows nobody needs: no branch here)
        __syncthreads();
        db = dmbuf + ((par ^ 1) * 2) * X4_DROW + a_lane;       // (this segment's last fragments are in registers: the pointers move on)
        vb[0] = vring + nbase * X4_VROW + b_lane;
        vb[1] = vring + ((nbase + 2) & (X4_RING - 1)) * X4_VROW + b_lane;
        X4_READ(fa0, fb0, db, vb, 0)
        X4_FENCE();
        X4_KSTEP(fa1, fb1, -1)
        X4_FENCE();
        base = nbase; par ^= 1;
        if (!n1_valid) break;
        n1_valid = n2.valid; n1_cont = n2.cont; ++seg;
        n2 = seg_after(n2, seg + 2);
    }
#undef X4_READ
#undef X4_KSTEP
#undef X4_FENCE
    __syncthreads();

    if (a.bias_part && cit == 0) {     // the xi = 1 waves hold column sums of dy: lane pairs (c, c + 32) meet in LDS, fixed order
        float* red = lds;
        if (xi == 1) red[cot2 * 64 + lane] = bsum;
        __syncthreads();
        if (tid < 64 && co0 + tid < a.Cout) {
            const int h = tid >> 5, c = tid & 31;
            a.bias_part[(size_t)sp * a.Cout + co0 + tid] = red[h * 64 + c] + red[h * 64 + 32 + c];
        }
        __syncthreads();
    }
    // ---- G^T: park the accumulators as ob[cot2][xi][ky][row 32][col 32], then every thread finishes 8 (co, ci) positions ------
    float* const ob = lds;
    if (NEST) {      // G2^T first, in registers: the four y-planes of this wave's (co half, xi) become its three ky taps
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float hs = 0.5f * (acc[1][j] + acc[2][j]), hd = 0.5f * (acc[1][j] - acc[2][j]);
            acc[0][j] = acc[0][j] + hs; acc[1][j] = hd; acc[2][j] = hs + acc[3][j];
        }
    }
    {
        float* o = ob + ((cot2 * 6 + xi) * 3) * 1024 + c32;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int rw = (j >> 2) * 8 + ks * 4 + (j & 3);          // D layout of the 32x32 tile: row = co, col = lane & 31 = ci
                o[ky * 1024 + rw * 32] = acc[ky][j];
            }
    }
    __syncthreads();
    float* const out = a.slab + (size_t)sp * 9 * a.Cout * a.Cin;
    const unsigned tap = (unsigned)a.Cout * a.Cin;
    for (int e = tid; e < 2 * 3 * 1024; e += X4_NT) {
        const int col = e & 31, rw = (e >> 5) & 31, ky = (e >> 10) % 3, h = e / 3072;
        const float* u = ob + (h * 6 * 3 + ky) * 1024 + rw * 32 + col;       // + xi * 3 * 1024
        const float u0 = u[0], u1 = u[3072], u2 = u[2 * 3072], u3 = u[3 * 3072], u4 = u[4 * 3072], u5 = u[5 * 3072];
        const float s12 = u1 + u2, d12 = u1 - u2, s34 = u3 + u4, d34 = u3 - u4;
        const float w0 = (0.25f * u0 - (1.0f / 6.0f) * s12) + (1.0f / 24.0f) * s34;
        const float w1 = ((-1.0f / 6.0f) * d12) + (1.0f / 12.0f) * d34;
        const float w2 = ((-1.0f / 6.0f) * s12) + ((1.0f / 6.0f) * s34 + u5);
        const unsigned go = ((unsigned)(ky * 3) * a.Cout + co0 + h * 32 + rw) * a.Cin + ci0 + col;
        out[go] = w0; out[go + tap] = w1; out[go + 2 * tap] = w2;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Round 5: the same kernel with the staging on waves of its OWN (conv3x3_wgrad_wino4p_kernel, 16 waves: 12 consumers + 4 producers,
// one producer per SIMD).  fp32 MFMAs and VALU instructions share the SIMD's datapath, so the staging's ~400 VALU instructions per
// segment cost matrix time wherever they run - but in the 12-wave form they sat in the MFMA waves' own instruction streams: nine of the
// twelve waves left the MFMA loop for ~1000 - 1500 cycles at the same point of every segment, the other three ran ahead into the barrier
// and waited (in-kernel stamps, profiles/r05_wgrad_notes.txt).  Here a consumer wave's stream is fragment reads, six adds and four
// MFMAs per k-step and nothing else; a producer wave loads, transforms and stores what the NEXT segment needs and waits at the
// segment's barrier; its VALU work is spread evenly over the four SIMDs (V wave-item p + dM wave-item p on producers 0..2, the three
// row-1 dM wave-items on producer 3: 98 / 98 / 98 / 111 VALU-equivalents per segment).  Same LDS image, ring, segment walk, split-K
// slab and epilogue as above; 128 VGPRs per wave (four waves per SIMD).
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int P4_NT = 1024;

struct P4Seg { int img, xs, row; bool cont, valid; };
__device__ __forceinline__ P4Seg p4_seg_after(const Wg4Args& a, const P4Seg& c, int index, int seg_end) {   // the segment behind c; index = its number in the walk
    P4Seg n;
    n.valid = index < seg_end;
    if (c.row + 2 < a.H) { n.img = c.img; n.xs = c.xs; n.row = c.row + 2; n.cont = true; }
    else { n.cont = false; n.row = 0; n.xs = c.xs + 1; n.img = c.img; if (n.xs == a.segs_x) { n.xs = 0; n.img = c.img + 1; } }
    return n;
}

// The producer waves' whole life (P3: the wave that stages the three row-1 dM wave-items; otherwise V wave-item pw + dM wave-item pw).
template <bool P3>
__device__ __forceinline__ void p4_producer(const Wg4Args& a, float* const vring, float* const dmbuf, const int pw, const int lane,
                                            const int ci0, const int co0, const int seg_begin, const int seg_end, const P4Seg cur) {
    using Seg = P4Seg;
    auto seg_after = [&](const Seg& c, int index) { return p4_seg_after(a, c, index, seg_end); };
    // Wave-items (64 items each): V wave-item k = items 64k .. 64k+63 of the 192 (row rr = item / 96, x-tile (item % 96) >> 3,
    // ci group item & 7: six input columns of four channels), dM wave-item k of the 384 (row k / 3, x-tile (item % 192) >> 4,
    // co group item & 15: four gradient columns).  Producers 0..2: slot V = V wave-item p (r[0..5]), slot H = the same item of the
    // SECOND pair of halo rows where a segment starts a strip (r[6..11]; same columns), slot D = dM wave-item p (r[12..15]).
    // Producer 3: dM wave-items 3, 4, 5 (r[0..3], r[4..7], r[8..11]).
    const int XTW = a.W >> 2;
    const int d_C = a.ps_in ? (a.Cout >> 2) : a.Cout;
    const unsigned x_row_bytes = (unsigned)a.W * a.Cin * 4, d_row_bytes = (unsigned)a.W * a.Cout * 4;
    const unsigned x_side_bytes = (unsigned)(a.side - 1) * a.H * x_row_bytes, d_side_bytes = (unsigned)(a.side - 1) * a.H * d_row_bytes;
    u32x4 r[16];
    unsigned off[12];                 // byte offsets of the slots' columns from the start of their first row; 2^31 = outside
    constexpr int NS = P3 ? 3 : 2;    // slots of this producer: three dM wave-items, or a V wave-item (slot 0, the halo pair shares it) and a dM one
    int x0[NS];                       // strip-relative first column of the slots' items (+ 48 * strip)
    unsigned cb[NS], psz[NS];         // channel-group byte offset (+ row pitch of the lane's row for V), pixel pitch in bytes
    int pos[NS], rrw[NS], tix[NS];    // LDS float position inside a plane, row of the pair, x-tile in the strip
    auto slot_is_v = [&](int s_) { return !P3 && s_ == 0; };
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_) {
        if (!P3 && s_ == 0) {                         // V wave-item pw
            const int item = 64 * pw + lane, vt = (item % 96) >> 3, vc4 = item & 7;
            const int sub = a.side > 1 ? vt / XTW : 0;
            rrw[s_] = item / 96; tix[s_] = vt; pos[s_] = vt * 32 + vc4 * 4;
            x0[s_] = 4 * (vt - sub * XTW) - 1;
            psz[s_] = (unsigned)a.Cin * 4;
            cb[s_] = (unsigned)(((sub * a.H) * a.W) * a.Cin + ci0 + vc4 * 4) * 4 + (unsigned)rrw[s_] * x_row_bytes;
        } else {                                      // a dM wave-item: pw (slot 1 of producers 0..2) or 3 + s_ (producer 3)
            const int k = P3 ? 3 + s_ : pw;
            const int item = 64 * k + lane, dt = (item % 192) >> 4, dc4 = item & 15;
            const int sub0 = a.side > 1 ? dt / XTW : 0;
            rrw[s_] = item / 192; tix[s_] = dt; pos[s_] = dt * 64 + dc4 * 4;
            x0[s_] = 4 * (dt - sub0 * XTW);
            const int pch = co0 + dc4 * 4;
            if (a.ps_in) {
                const int sub = pch / d_C, cc = pch - sub * d_C;
                cb[s_] = (unsigned)(((sub >> 1) * (2 * a.W) + (sub & 1)) * d_C + cc) * 4;
                psz[s_] = (unsigned)(2 * d_C) * 4;
            } else {
                cb[s_] = (unsigned)((sub0 * a.H * a.W) * a.Cout + pch) * 4;
                psz[s_] = (unsigned)a.Cout * 4;
            }
        }
    }
    int s_xs = 0;
    auto set_strip = [&](int xs, int grp) {               // (once per strip: the only place the offsets cost VALU instructions)
        const int dx = (xs - s_xs) * 48;
        s_xs = xs;
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) {
            x0[s_] += dx;
            int xx = x0[s_];
            if (a.side > 1 && grp * a.side + tix[s_] / XTW >= a.N) xx += 0x100000;       // an image of the last group that does not exist
            const int ncol = slot_is_v(s_) ? 6 : 4;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                if (j >= ncol) continue;
                const unsigned x = (unsigned)(xx + j);
                const unsigned o = x < (unsigned)a.W ? cb[s_] + x * psz[s_] : 0x80000000u;
                off[(P3 ? 4 : 6) * s_ + j] = o;
            }
        }
    };
    auto uniform_ptr = [](const float* p) -> const float* {
        const unsigned long long v = (unsigned long long)p;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return (const float*)(((unsigned long long)hi << 32) | lo);
    };
    // loads of segment g (a segment that does not exist: nothing is issued)
    auto load_seg = [&](const Seg& g) {
        if (!g.valid) return;
        if constexpr (!P3) {
            // V pair(s): a continuing segment adds rows row+1, row+2; a strip start needs row-1, row (slot V) and row+1, row+2 (slot H)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (h == 1 && g.cont) continue;
                const int v_iy0 = g.cont ? g.row + 1 : (h ? g.row + 1 : g.row - 1);
                const float* const rowp = a.x + ((long)g.img * a.side * a.H + v_iy0) * ((long)a.W * a.Cin);
                const int iy = v_iy0 + rrw[0];
                bool lane_ok = true;
                unsigned bytes;
                if (g.cont && a.side == 1) {
                    const int rows = a.H - v_iy0 > 2 ? 2 : (a.H - v_iy0 < 0 ? 0 : a.H - v_iy0);
                    bytes = (unsigned)rows * x_row_bytes;
                } else {
                    bytes = x_side_bytes + 2 * x_row_bytes;
                    lane_ok = iy >= 0 && iy < a.H;
                }
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)uniform_ptr(rowp), 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
#pragma unroll
                for (int j = 0; j < 6; ++j) r[6 * h + j] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane_ok ? off[j] : 0x80000000u, 0, 0);
            }
        }
        // dM wave-items
#pragma unroll
        for (int s_ = P3 ? 0 : 1; s_ < NS; ++s_) {
            const int oy = g.row + __builtin_amdgcn_readfirstlane(rrw[s_]);
            const bool row_ok = oy < a.H;
            const int ry = row_ok ? oy : 0;
            const float* const rowp = a.ps_in ? a.dy + ((size_t)g.img * (2 * a.H) + 2 * ry) * (2 * a.W) * d_C
                                              : a.dy + ((size_t)g.img * a.side * a.H + ry) * a.W * a.Cout;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)uniform_ptr(rowp), 0,
                                                  __builtin_amdgcn_readfirstlane(row_ok ? d_side_bytes + d_row_bytes : 0u), 0x00020000);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (!P3) r[12 + j] = __builtin_amdgcn_raw_buffer_load_b128(rs, off[6 + j], 0, 0);
                else r[4 * s_ + j] = __builtin_amdgcn_raw_buffer_load_b128(rs, off[4 * s_ + j], 0, 0);
            }
        }
    };
    const f32x4 c_m5 = {-5.0f, -5.0f, -5.0f, -5.0f}, c_m4 = {-4.0f, -4.0f, -4.0f, -4.0f}, c_p4 = {4.0f, 4.0f, 4.0f, 4.0f};
    auto store_v = [&](const int rb, float* p) {           // six planes of one V item
        const f32x4 d0 = __builtin_bit_cast(f32x4, r[rb]), d1 = __builtin_bit_cast(f32x4, r[rb + 1]), d2 = __builtin_bit_cast(f32x4, r[rb + 2]),
                    d3 = __builtin_bit_cast(f32x4, r[rb + 3]), d4 = __builtin_bit_cast(f32x4, r[rb + 4]), d5 = __builtin_bit_cast(f32x4, r[rb + 5]);
        *(f32x4*)(p) = __builtin_elementwise_fma(c_p4, d0, __builtin_elementwise_fma(c_m5, d2, d4));
        *(f32x4*)(p + 5 * X4_VPLANE) = __builtin_elementwise_fma(c_p4, d1, __builtin_elementwise_fma(c_m5, d3, d5));
        const f32x4 t1 = __builtin_elementwise_fma(c_m4, d2, d4), t2 = __builtin_elementwise_fma(c_m4, d1, d3);
        *(f32x4*)(p + X4_VPLANE) = t1 + t2;
        *(f32x4*)(p + 2 * X4_VPLANE) = t1 - t2;
        const f32x4 t3 = d4 - d2, t4 = 2.0f * (d3 - d1);
        *(f32x4*)(p + 3 * X4_VPLANE) = t3 + t4;
        *(f32x4*)(p + 4 * X4_VPLANE) = t3 - t4;
    };
    auto store_d = [&](const int rb, float* p) {           // six planes of one dM item
        const f32x4 g0 = __builtin_bit_cast(f32x4, r[rb]), g1 = __builtin_bit_cast(f32x4, r[rb + 1]), g2 = __builtin_bit_cast(f32x4, r[rb + 2]),
                    g3 = __builtin_bit_cast(f32x4, r[rb + 3]);
        const f32x4 e02 = g0 + g2, e13 = g1 + g3, f02 = g0 + 4.0f * g2, f13 = 2.0f * (g1 + 4.0f * g3);
        *(f32x4*)(p) = g0;
        *(f32x4*)(p + X4_DPLANE) = e02 + e13;
        *(f32x4*)(p + 2 * X4_DPLANE) = e02 - e13;
        *(f32x4*)(p + 3 * X4_DPLANE) = f02 + f13;
        *(f32x4*)(p + 4 * X4_DPLANE) = f02 - f13;
        *(f32x4*)(p + 5 * X4_DPLANE) = g3;
    };
    // what was loaded for segment g goes to ring slots v_slot0 .. (+3 at a strip start) and dM buffer d_buf
    auto store_seg = [&](const Seg& g, int v_slot0, int d_buf) {
        if (!g.valid) return;
        if constexpr (!P3) {
            store_v(0, vring + ((v_slot0 + rrw[0]) & (X4_RING - 1)) * X4_VROW + pos[0]);
            if (!g.cont) store_v(6, vring + ((v_slot0 + 2 + rrw[0]) & (X4_RING - 1)) * X4_VROW + pos[0]);
            store_d(12, dmbuf + (d_buf * 2 + rrw[1]) * X4_DROW + pos[1]);
        } else {
#pragma unroll
            for (int s_ = 0; s_ < 3; ++s_) store_d(4 * s_, dmbuf + (d_buf * 2 + rrw[s_]) * X4_DROW + pos[s_]);
        }
    };
    Seg n1 = seg_after(cur, seg_begin + 1);
    set_strip(cur.xs, cur.img);
    load_seg(cur);
    store_seg(cur, 0, 0);
    if (n1.valid && !n1.cont) set_strip(n1.xs, n1.img);
    load_seg(n1);
    Seg n2 = seg_after(n1, seg_begin + 2);
    __syncthreads();
    int base = 0, par = 0;
#pragma unroll 1
    for (int seg = seg_begin; seg < seg_end; ++seg) {
        // during segment `seg`: store what segment seg + 1 needs (its loads were issued a segment ago), load for seg + 2
        store_seg(n1, base + 4, par ^ 1);
        if (n2.valid && !n2.cont) set_strip(n2.xs, n2.img);
        load_seg(n2);
        __syncthreads();                                   // the segment's one barrier (the consumers reach it in front of their last k-step)
        base = (base + (n1.cont ? 2 : 4)) & (X4_RING - 1);
        par ^= 1;
        n1 = n2;
        n2 = seg_after(n2, seg + 3);
    }
}

template <bool NEST>
__global__ __launch_bounds__(P4_NT) void conv3x3_wgrad_wino4p_kernel(const Wg4Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const vring = lds;                             // [8 slots] V rows
    float* const dmbuf = lds + X4_RING * X4_VROW;         // [2 buffers][2 rows] dM rows
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave >= 12;
    const int pw = wave - 12;                             // producer index 0..3
    const int c32 = lane & 31, ks = lane >> 5;            // consumer fragment lane: channel, k-slot (x-tile 2q + ks of k-step q)
    const int cot2 = producer ? 0 : wave / 6, xi = producer ? 0 : wave - cot2 * 6;

    int bid = blockIdx.x;
    if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
    const int cit = bid % a.ci_tiles;  bid /= a.ci_tiles;
    const int cot = bid % a.co_tiles;
    const int sp = bid / a.co_tiles;
    const int ci0 = cit * 32, co0 = cot * 64;

    const int seg_begin = sp * a.segs_per_split;
    int seg_end = seg_begin + a.segs_per_split;
    if (seg_end > a.total_segs) seg_end = a.total_segs;
    if (seg_begin >= seg_end) return;

    using Seg = P4Seg;
    auto seg_after = [&](const Seg& c, int index) { return p4_seg_after(a, c, index, seg_end); };
    Seg cur;
    {
        const int strip = seg_begin / a.segs_y;
        cur.row = 2 * (seg_begin - strip * a.segs_y);
        cur.img = strip / a.segs_x;
        cur.xs = strip - cur.img * a.segs_x;
        cur.cont = false; cur.valid = true;
    }

    constexpr int NACC = NEST ? 4 : 3;
    f32x16 acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
    float bsum = 0.f;

    if (producer) {
        // The producers run ABOVE the MFMA waves' priority: their ~100 VALU instructions and stores per segment then issue when their
        // loads are there instead of queueing behind three waves' MFMAs (at equal priority the consumers met the segment's barrier
        // before the producers: 151.6 us; priority 1 / 2 / 3: 130.1 / 129.5 / 131.6; the round-4 kernel: 146.3 - profiles/r05_wgrad_notes.txt)
        __builtin_amdgcn_s_setprio(2);
        if (pw == 3) p4_producer<true>(a, vring, dmbuf, pw, lane, ci0, co0, seg_begin, seg_end, cur);
        else p4_producer<false>(a, vring, dmbuf, pw, lane, ci0, co0, seg_begin, seg_end, cur);
    } else {
        // ================================================= consumers =================================================
        const int b_lane = xi * X4_VPLANE + ks * 32 + c32;
        const int a_lane = xi * X4_DPLANE + ks * 64 + cot2 * 32 + c32;
        Seg n1 = seg_after(cur, seg_begin + 1);
        __syncthreads();                                       // the first segment is staged
        int base = 0, par = 0;
        constexpr int KQ = G4_TXT / 2;
        float fa0[2], fb0[4], fa1[2], fb1[4];
#define P4_READ(FA, FB, DB, VB, Q)                                                               \
        {                                                                                        \
            FA[0] = DB[(Q) * 128]; FA[1] = DB[X4_DROW + (Q) * 128];                              \
            FB[0] = VB[0][(Q) * 64]; FB[1] = VB[0][X4_VROW + (Q) * 64];                          \
            FB[2] = VB[1][(Q) * 64]; FB[3] = VB[1][X4_VROW + (Q) * 64];                          \
        }
#define P4_KSTEP(FA, FB)                                                                         \
        if (NEST) {                                                                              \
            const float ds_ = FA[0] + FA[1], dd_ = FA[0] - FA[1];                                \
            const float x0_ = FB[0] - FB[2], x1_ = FB[1] + FB[2], x2_ = FB[2] - FB[1], x3_ = FB[3] - FB[1]; \
            if (xi == 1) { asm volatile("" : "+v"(bsum)); bsum += ds_; }                         \
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[0], x0_, acc[0], 0, 0, 0);          \
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ds_, x1_, acc[1], 0, 0, 0);            \
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(dd_, x2_, acc[2], 0, 0, 0);            \
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[1], x3_, acc[3], 0, 0, 0);          \
        } else {                                                                                 \
            _Pragma("unroll") for (int ky = 0; ky < 3; ++ky) {                                   \
                acc[ky] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[0], FB[ky], acc[ky], 0, 0, 0); \
                acc[ky] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[1], FB[ky + 1], acc[ky], 0, 0, 0); \
            }                                                                                    \
            if (xi == 1) { asm volatile("" : "+v"(bsum)); bsum += FA[0] + FA[1]; }               \
        }
        const float* db = dmbuf + a_lane;
        const float* vb[2] = {vring + b_lane, vring + 2 * X4_VROW + b_lane};
        P4_READ(fa0, fb0, db, vb, 0)
#pragma unroll 1
        for (int seg = seg_begin; seg < seg_end; ++seg) {
            const int nbase = (base + (n1.cont ? 2 : 4)) & (X4_RING - 1);
#pragma unroll
            for (int q = 0; q < KQ - 2; q += 2) {
                P4_READ(fa1, fb1, db, vb, q + 1)
                P4_KSTEP(fa0, fb0)
                P4_READ(fa0, fb0, db, vb, q + 2)
                P4_KSTEP(fa1, fb1)
            }
            P4_READ(fa1, fb1, db, vb, KQ - 1)
            P4_KSTEP(fa0, fb0)
            // the segment's one barrier, in front of its last k-step: this segment's last fragments are in registers, the next
            // segment's rows are stored - its first fragments are read here, under that k-step's MFMAs
            __syncthreads();
            db = dmbuf + ((par ^ 1) * 2) * X4_DROW + a_lane;
            vb[0] = vring + nbase * X4_VROW + b_lane;
            vb[1] = vring + ((nbase + 2) & (X4_RING - 1)) * X4_VROW + b_lane;
            P4_READ(fa0, fb0, db, vb, 0)
            P4_KSTEP(fa1, fb1)
            base = nbase; par ^= 1;
            n1 = seg_after(n1, seg + 2);
        }
#undef P4_READ
#undef P4_KSTEP
    }
    __syncthreads();

    if (a.bias_part && cit == 0) {     // the xi = 1 waves hold column sums of dy: lane pairs (c, c + 32) meet in LDS, fixed order
        float* red = lds;
        if (!producer && xi == 1) red[cot2 * 64 + lane] = bsum;
        __syncthreads();
        if (tid < 64 && co0 + tid < a.Cout) {
            const int h = tid >> 5, c = tid & 31;
            a.bias_part[(size_t)sp * a.Cout + co0 + tid] = red[h * 64 + c] + red[h * 64 + 32 + c];
        }
        __syncthreads();
    }
    float* const ob = lds;
    if (!producer) {
        if (NEST) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float hs = 0.5f * (acc[1][j] + acc[2][j]), hd = 0.5f * (acc[1][j] - acc[2][j]);
                acc[0][j] = acc[0][j] + hs; acc[1][j] = hd; acc[2][j] = hs + acc[3][j];
            }
        }
        float* o = ob + ((cot2 * 6 + xi) * 3) * 1024 + c32;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int rw = (j >> 2) * 8 + ks * 4 + (j & 3);
                o[ky * 1024 + rw * 32] = acc[ky][j];
            }
    }
    __syncthreads();
    float* const out = a.slab + (size_t)sp * 9 * a.Cout * a.Cin;
    const unsigned tap = (unsigned)a.Cout * a.Cin;
    for (int e = tid; e < 2 * 3 * 1024; e += P4_NT) {
        const int col = e & 31, rw = (e >> 5) & 31, ky = (e >> 10) % 3, h = e / 3072;
        const float* u = ob + (h * 6 * 3 + ky) * 1024 + rw * 32 + col;
        const float u0 = u[0], u1 = u[3072], u2 = u[2 * 3072], u3 = u[3 * 3072], u4 = u[4 * 3072], u5 = u[5 * 3072];
        const float s12 = u1 + u2, d12 = u1 - u2, s34 = u3 + u4, d34 = u3 - u4;
        const float w0 = (0.25f * u0 - (1.0f / 6.0f) * s12) + (1.0f / 24.0f) * s34;
        const float w1 = ((-1.0f / 6.0f) * d12) + (1.0f / 12.0f) * d34;
        const float w2 = ((-1.0f / 6.0f) * s12) + ((1.0f / 6.0f) * s34 + u5);
        const unsigned go = ((unsigned)(ky * 3) * a.Cout + co0 + h * 32 + rw) * a.Cin + ci0 + col;
        out[go] = w0; out[go + tap] = w1; out[go + 2 * tap] = w2;
    }
}

namespace {
struct Wg4Plan { int co_tiles, ci_tiles, segs_x, segs_y, total_segs, split, segs_per_split, side; size_t slab_bytes, total_bytes; };

static bool wg4_plan(int N, int H, int W, int Cin, int Cout, Wg4Plan* p) {
    if (W % 4 || Cin % 64 || Cout % 64 || N < 1 || H < 1) return false;
    p->co_tiles = Cout / 64; p->ci_tiles = Cin / 32;
    p->segs_y = (H + 1) / 2;
    p->side = 1;
    if (W < 48) {
        // Rows shorter than a strip (round 4; the 32x32x2 kernel only): 12 / (W / 4) images side by side in one strip, e.g. two of the
        // Discriminator's 24-pixel-wide images (features.6: 194 us on the direct kernel, which issues three times the multiplies).
        const int xtw = W / 4;
        if (xtw < 2 || G4_TXT % xtw) return false;
        p->side = G4_TXT / xtw;
        if ((size_t)p->side * H * W * (Cin > Cout ? Cin : Cout) * 4 >= ((size_t)1 << 31)) return false;   // 32-bit offsets inside a strip
        const int groups = (N + p->side - 1) / p->side;
        if ((long)groups * G4_TXT * 8 > (long)N * xtw * 9) return false;          // a mostly empty last group
        p->segs_x = 1;
        p->total_segs = groups * p->segs_y;
    } else {
        p->segs_x = (W / 4 + G4_TXT - 1) / G4_TXT;
        // a ragged last strip wastes MFMAs on zeros: accept up to ~1/8
        if ((long)p->segs_x * G4_TXT * 8 > (long)(W / 4) * 9) return false;
        p->total_segs = N * p->segs_x * p->segs_y;
    }
    const int tiles = p->co_tiles * p->ci_tiles;
    int split = (256 + tiles - 1) / tiles;
    if (split > p->total_segs) split = p->total_segs;
    if (split < 1) split = 1;
    p->segs_per_split = (p->total_segs + split - 1) / split;
    // whole strips per workgroup where possible: a strip change re-stages four halo rows synchronously
    if (p->segs_per_split > p->segs_y) p->segs_per_split = (p->segs_per_split + p->segs_y - 1) / p->segs_y * p->segs_y;
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
// variant 0: the 16x16x4 kernel (8 waves); 1: the 32x32x2 kernel (12 waves), 1-D transform; 2: the same kernel with the transform nested in y;
// 3: the y-nested transform with the staging on four producer waves (16 waves; the product)
int pesr_conv3x3_wgrad_wino4_launch(const float* x, const float* dy, float* dw, float* db, int N, int H, int W, int Cin, int Cout,
                                    float alpha, int ps_in, int accumulate, int variant, void* ws, size_t ws_bytes, hipStream_t stream) {
    Wg4Plan p;
    if (!wg4_plan(N, H, W, Cin, Cout, &p)) return PESR_EINVAL;
    if (p.side > 1 && (variant == 0 || ps_in)) return PESR_EINVAL;     // side-by-side strips: the 32x32x2 kernel, plain gradients
    if (!ws || ws_bytes < p.total_bytes) return PESR_EWORKSPACE;
    if (ps_in && Cout % 256) return PESR_EINVAL;
    Wg4Args a{};
    a.x = x; a.dy = dy; a.slab = (float*)ws;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.segs_x = p.segs_x; a.segs_y = p.segs_y; a.total_segs = p.total_segs; a.segs_per_split = p.segs_per_split;
    a.co_tiles = p.co_tiles; a.ci_tiles = p.ci_tiles; a.ps_in = ps_in; a.side = p.side;
    a.bias_part = db ? (float*)((char*)ws + p.slab_bytes + (((size_t)Cout * sizeof(double) + 255) / 256) * 256) : nullptr;
    constexpr size_t lds = (size_t)(G4_RING * G4_VROW + 4 * G4_DROW) * sizeof(float);
    static_assert(lds >= (size_t)9 * 64 * 32 * sizeof(float), "epilogue staging fits");
    static_assert(lds <= 160 * 1024, "wgrad-wino4 LDS budget");
    static PesrDeviceOnce attr_once;
    attr_once([&] {
        (void)hipFuncSetAttribute((const void*)conv3x3_wgrad_wino4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    constexpr size_t ldsx = (size_t)2 * 6 * 3 * 1024 * sizeof(float);      // the variant's G^T staging (144 KiB) = its eight-slot ring + dM buffers
    static_assert(ldsx >= (size_t)(X4_RING * X4_VROW + 4 * X4_DROW) * sizeof(float) && ldsx <= 160 * 1024, "wgrad-wino4x LDS budget");
    static PesrDeviceOnce attr_once_x;
    attr_once_x([&] {
        (void)hipFuncSetAttribute((const void*)conv3x3_wgrad_wino4x_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv3x3_wgrad_wino4x_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv3x3_wgrad_wino4p_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    const int grid = p.split * p.co_tiles * p.ci_tiles;
    if (variant == 3) hipLaunchKernelGGL(conv3x3_wgrad_wino4p_kernel<true>, dim3(grid), dim3(P4_NT), ldsx, stream, a);
    else if (variant == 2) hipLaunchKernelGGL(conv3x3_wgrad_wino4x_kernel<true>, dim3(grid), dim3(X4_NT), ldsx, stream, a);
    else if (variant == 1) hipLaunchKernelGGL(conv3x3_wgrad_wino4x_kernel<false>, dim3(grid), dim3(X4_NT), ldsx, stream, a);
    else hipLaunchKernelGGL(conv3x3_wgrad_wino4_kernel, dim3(grid), dim3(G4_NT), lds, stream, a);
    int rc = pesr_launch_status();
    if (rc) return rc;
    // the partial blocks already are dw in tap order: the direct kernel's fixed-order reduce finishes the job
    return pesr_wgrad_reduce_launch((const float*)ws, dw, p.split, Cout, Cin, alpha, ps_in, (const float*)a.bias_part, p.split, db, accumulate, stream);
}
