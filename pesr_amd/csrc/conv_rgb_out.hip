// Forward of the C -> 3 conv (3x3, stride 1, zero padding 1): the Generator's last layer (reference model/pesr.py:38,
// `Conv(num_channels, 3)`), HBM-bound: it reads 1 KiB per output pixel at C = 256 and writes 12 bytes.
//
// The implicit-GEMM kernel pads the 3 output channels to an MFMA tile of 16 and runs 9 taps, so it is bound by MFMAs that
// multiply zeros (1.4 TB/s of input).  Here the whole stencil moves into the MFMA's N dimension: for every INPUT pixel
//     P[r][x'][n] = sum_c x[r][x'][c] * w[co][c][ky][kx],      n = ky*9 + kx*3 + co   (27 of 32 columns = two 16-wide MFMA tiles)
//     out[y][x][co] = bias[co] + sum_{ky,kx} P[y + ky - 1][x + kx - 1][ky*9 + kx*3 + co]
// i.e. two MFMAs per (16 pixels, 4 channels) instead of nine, and every input row is fetched ONCE, straight from global memory
// into the MFMA's A operand (lane = pixel, 16 bytes = 4 channels = 4 k-steps): no LDS staging of activations at all.  LDS
// holds the weights (32 KiB at C = 256, scattered there from the OIHW tensor at kernel start) and a ring of four P rows from
// which a finished output row gathers its nine terms.
//
// One workgroup (4 waves) walks a band of TH output rows of a 192-column strip (TH + 2 input rows); wave w owns the 16-pixel
// column tiles w, w+4, w+8.  Its loads run D - 1 16-channel chunks ahead (D = 8, or 4 when C is not a multiple of 128), across
// row boundaries: ~80 KB in flight per CU.  (6 or 12 waves with 2 / 1 column tiles each measured 3 % / 20 % slower.)  Images
// wider than 192 are cut into strips of 190 output columns whose 192 input columns overlap by one on each side.
// Measured at 16 x 192 x 192 x 256: 166 us = 3.6 TB/s of activations (the implicit-GEMM kernel: 450 us); without the MFMAs
// the same loop takes 159 us, i.e. the 64-byte-per-pixel access pattern of the A operand is the limit.
#include <mutex>
#include "common.h"
#include "launchers.h"

namespace {
constexpr int RO_TH = 12;         // output rows per workgroup (input rows read: TH + 2)
constexpr int RO_NW = 4;          // waves per workgroup (6 / 12 with 2 / 1 column tiles each measured 3 % / 20 % slower)
constexpr int RO_MT = 12 / RO_NW; // column tiles per wave
constexpr int RO_NT = RO_NW * 64; // threads
constexpr int RO_COLS = RO_NW * RO_MT * 16;   // 192 input columns per strip
constexpr int RO_PS = 28;         // floats per pixel of a P row (27 used; 28 keeps the D-tile writes bank-conflict free)
constexpr int RO_PROW = (RO_COLS + 2) * RO_PS;   // floats: [col -1 .. 192][28]

struct RgbOutArgs {
    const float* x;     // [N][H][W][C]
    const float* w;     // OIHW [3][C][3][3]
    const float* bias;  // [3] or null
    float* y;           // [N][H][W][3]
    int N, H, W, C;
    int strips, halo, outw;   // strips per row; 1 when strips overlap (W > 192), else 0; output columns per strip
    int bands;                // ceil(H / TH)
    int act; float slope;
    int wmode;                // 0: w is the conv's own OIHW [3][C][3][3]; 1: w is OIHW [C][3][3][3] of a 3 -> C conv whose INPUT GRADIENT
                              //    this launch computes (x = its dy): w'[co][c][ky][kx] = w[c][co][2-ky][2-kx]
};

template <int RO_D>               // chunk ring depth per column tile (RO_D - 1 loads in flight)
__global__ __launch_bounds__(RO_NT) void conv_rgb_out_kernel(const RgbOutArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int C16 = a.C >> 4;
    float* const wl = lds;                                  // [chunk C16][n 32][16 ch]
    float* const pring = lds + C16 * 512;                   // [4][RO_PROW]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;

    int b = blockIdx.x;
    if ((gridDim.x & 7) == 0) b = (b & 7) * (gridDim.x >> 3) + (b >> 3);   // neighbouring bands (shared halo rows) on one XCD / L2
    const int strip = b % a.strips;  b /= a.strips;
    const int band = b % a.bands;
    const int img = b / a.bands;
    const int y0 = band * RO_TH;
    const int xin0 = strip * a.outw - a.halo;               // image column of the strip's local column 0

    // ---- per-lane A offsets: column tile t of this wave, pixel i, k-slot g (bytes inside an input row; out of the image: 2^31)
    unsigned a_off[RO_MT];
#pragma unroll
    for (int t = 0; t < RO_MT; ++t) {
        const int col = xin0 + (wave + RO_NW * t) * 16 + i;
        a_off[t] = (col >= 0 && col < a.W) ? (unsigned)((col * a.C + 4 * g) * 4) : 0x80000000u;
    }
    const unsigned row_bytes = (unsigned)a.W * a.C * 4;
    const float* const b_lane = wl + i * 16 + 4 * g;        // + chunk * 512 (+ 256 for the second N tile)
    auto row_rsrc = [&](int r) -> __amdgpu_buffer_rsrc_t {  // image row r of this band's image (an invalid row: empty descriptor)
        const bool ok = r >= 0 && r < a.H;
        const float* const rowp = a.x + ((size_t)img * a.H + (ok ? r : 0)) * a.W * a.C;
        const unsigned long long pv = (unsigned long long)rowp;                              // provably wave-uniform base: no waterfall loops
        return __builtin_amdgcn_make_buffer_rsrc(
            (void*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(pv >> 32)) << 32) |
                    (unsigned)__builtin_amdgcn_readfirstlane((unsigned)pv)),     // (unsigned): the builtin returns int - no sign extension
            0, __builtin_amdgcn_readfirstlane(ok ? row_bytes : 0u), 0x00020000);
    };

    // Input rows ti = 0 .. TH+1 are image rows y0 - 1 + ti.  Even bands walk them downwards, odd bands upwards, so that a band and its
    // neighbour (same XCD, see above) read the two halo rows they share at the same time and the second read hits L2 - which takes the
    // SAME order inside the shared pair (round 6: with the pair crossed, the two reads of a row were one row step = ~6 MB of the XCD's
    // traffic apart and both went to HBM: 5.34 M fabric requests for 4.72 M lines).  An odd band therefore walks 12, 13, 11, 10 .. 2, 0, 1.
    const int rows_in = RO_TH + 2;
    const bool rev = band & 1;
    auto row_of_step = [&](int step) -> int {
        if (!rev) return step;
        if (step < 2) return rows_in - 2 + step;
        if (step >= rows_in - 2) return step - (rows_in - 2);
        return rows_in - 1 - step;
    };
    u32x4 fa[RO_D][RO_MT];
    bool primed = false;
    for (int step = 0; step < rows_in; ++step) {            // the first valid row's loads go out before the weights are staged
        const int r = y0 - 1 + row_of_step(step);
        if (r >= 0 && r < a.H) {
            const __amdgpu_buffer_rsrc_t rs = row_rsrc(r);
#pragma unroll
            for (int d = 0; d < RO_D - 1; ++d)
#pragma unroll
                for (int tt = 0; tt < RO_MT; ++tt) fa[d][tt] = __builtin_amdgcn_raw_buffer_load_b128(rs, a_off[tt], d * 64, 0);
            primed = true;
            break;
        }
    }

    // ---- weights: OIHW -> [chunk][n = ky*9 + kx*3 + co][c & 15], columns n >= 27 zero; P ring zeroed (its pad columns stay zero).
    // The 27 C floats are read in their own order, sixteen loads of a thread in flight at a time (round 6: as a gather loop of one load
    // and one LDS store per iteration the 32 dependent round trips of this prologue cost ~20 us of a 170 us launch).
    for (int e = tid; e < C16 * 80; e += RO_NT) {
        const int chunk = e / 80, rem = e - chunk * 80;
        wl[chunk * 512 + 27 * 16 + rem] = 0.f;
    }
    const int wtotal = 27 * a.C;
#pragma unroll 1
    for (int base = 0; base < wtotal; base += 16 * RO_NT) {
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int idx = base + k * RO_NT + tid;
            v[k] = idx < wtotal ? a.w[idx] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int idx = base + k * RO_NT + tid;
            if (idx < wtotal) {
                const int t9 = idx / 9, k9 = idx - t9 * 9;
                int c, co, ky = k9 / 3, kx = k9 - ky * 3;
                if (a.wmode) { c = t9 / 3; co = t9 - c * 3; ky = 2 - ky; kx = 2 - kx; }      // w is [C][3][3][3] of the transposed conv, taps flipped
                else { co = (t9 >= a.C) + (t9 >= 2 * a.C); c = t9 - co * a.C; }              // w is [3][C][3][3]
                wl[(c >> 4) * 512 + (ky * 9 + kx * 3 + co) * 16 + (c & 15)] = v[k];
            }
        }
    }
    for (int e = tid; e < 4 * RO_PROW; e += RO_NT) pring[e] = 0.f;
    __syncthreads();

#pragma unroll 1
    for (int step = 0; step < rows_in; ++step) {
        const int t = row_of_step(step);
        const int r = y0 - 1 + t;
        const bool valid = r >= 0 && r < a.H;
        if (rev && step == rows_in - 2) __syncthreads();    // row 0 takes the ring slot of row 4, which the previous step's gather still reads
        if (valid) {                                        // rows outside the image contribute zeros: skipped in the gather below
            const __amdgpu_buffer_rsrc_t rs = row_rsrc(r);
            const int r_next = y0 - 1 + row_of_step(step + 1);
            const bool next_valid = step + 1 < rows_in && r_next >= 0 && r_next < a.H;
            const __amdgpu_buffer_rsrc_t rs_next = row_rsrc(r_next);
            if (!primed) {                                  // first row of the band: fill the pipeline
#pragma unroll
                for (int d = 0; d < RO_D - 1; ++d)
#pragma unroll
                    for (int tt = 0; tt < RO_MT; ++tt) fa[d][tt] = __builtin_amdgcn_raw_buffer_load_b128(rs, a_off[tt], d * 64, 0);
                primed = true;
            }
            f32x4 acc[2][RO_MT];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int tt = 0; tt < RO_MT; ++tt) acc[h][tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
            for (int ch0 = 0; ch0 < C16; ch0 += RO_D)       // C16 % 4 == 0: the ring slots are compile-time constants
#pragma unroll
            for (int u = 0; u < RO_D; ++u) {
                const int ch = ch0 + u, tgt = ch + RO_D - 1, slot = u, tslot = (u + RO_D - 1) & (RO_D - 1);
                if (tgt < C16) {
#pragma unroll
                    for (int tt = 0; tt < RO_MT; ++tt) fa[tslot][tt] = __builtin_amdgcn_raw_buffer_load_b128(rs, a_off[tt], tgt * 64, 0);
                } else if (next_valid) {
#pragma unroll
                    for (int tt = 0; tt < RO_MT; ++tt) fa[tslot][tt] = __builtin_amdgcn_raw_buffer_load_b128(rs_next, a_off[tt], (tgt - C16) * 64, 0);
                }
                const f32x4 fb0 = *(const f32x4*)(b_lane + ch * 512), fb1 = *(const f32x4*)(b_lane + ch * 512 + 256);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int tt = 0; tt < RO_MT; ++tt) {
                        const float av = __builtin_bit_cast(f32x4, fa[slot][tt])[kk];
                        acc[0][tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, fb0[kk], acc[0][tt], 0, 0, 0);
                        acc[1][tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, fb1[kk], acc[1][tt], 0, 0, 0);
                    }
            }
            if (!next_valid) primed = false;
            // P row -> ring slot t & 3 (D tile: lane holds pixels 4g .. 4g+3 of the tile, column n = i of its N tile)
            float* const pr = pring + (t & 3) * RO_PROW + RO_PS;           // local column 0
#pragma unroll
            for (int tt = 0; tt < RO_MT; ++tt)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    float* q = pr + ((wave + RO_NW * tt) * 16 + 4 * g + jj) * RO_PS;
                    q[i] = acc[0][tt][jj];
                    if (i < 11) q[16 + i] = acc[1][tt][jj];
                }
        }
        __syncthreads();
        // output row j (image row y0 + j) needs the P rows of input rows j, j+1, j+2: complete two steps behind the walk (an odd band's
        // rows 1 and 0 both with its last step)
        const int nout = step < 2 ? 0 : !rev ? 1 : step < rows_in - 2 ? 1 : step == rows_in - 1 ? 2 : 0;
        for (int o = 0; o < nout; ++o) {
            const int j = rev ? t - o : t - 2, oy = y0 + j;
            if (j < 0 || j >= RO_TH || oy >= a.H) continue;
            float* const yrow = a.y + ((size_t)img * a.H + oy) * a.W * 3;
            for (int u = tid; u < a.outw * 3; u += RO_NT) {
                const int lx = u / 3, co = u - lx * 3;              // output column strip * outw + lx = local column lx + halo
                const int ox = strip * a.outw + lx;
                if (ox < a.W) {
                    float v = 0.f;
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        const int rr = oy - 1 + ky;                 // = y0 - 1 + (j + ky)
                        if (rr >= 0 && rr < a.H) {
                            const float* q = pring + ((j + ky) & 3) * RO_PROW + (lx + a.halo) * RO_PS + ky * 9 + co;   // column x-1
                            v += (q[0] + q[RO_PS + 3]) + q[2 * RO_PS + 6];
                        }
                    }
                    if (a.bias) v += a.bias[co];
                    if (a.act == PESR_ACT_RELU) v = v > 0.f ? v : 0.f;
                    else if (a.act == PESR_ACT_LRELU) v = v > 0.f ? v : v * a.slope;
                    yrow[(size_t)ox * 3 + co] = v;
                }
            }
        }
    }
}
}  // namespace

static int rgb_out_launch(const float* x, const float* w, const float* bias, float* y, int N, int H, int W, int C, int act, float slope,
                          int wmode, hipStream_t stream) {
    if (N < 1 || H < 1 || W < 1 || C < 64 || C % 64 || C > 512) return PESR_EINVAL;   // the chunk loop is unrolled by the ring depth (>= 4)
    if ((size_t)W * C * 4 >= ((size_t)1 << 31)) return PESR_EINVAL;
    RgbOutArgs a{};
    a.x = x; a.w = w; a.bias = bias; a.y = y; a.N = N; a.H = H; a.W = W; a.C = C; a.act = act; a.slope = slope; a.wmode = wmode;
    if (W <= RO_COLS) { a.strips = 1; a.halo = 0; a.outw = RO_COLS; }
    else { a.halo = 1; a.outw = RO_COLS - 2; a.strips = pesr_cdiv(W, a.outw); }
    a.bands = pesr_cdiv(H, RO_TH);
    const size_t lds = ((size_t)(C / 16) * 512 + 4 * RO_PROW) * sizeof(float);
    static PesrDeviceOnce attr_once;
    attr_once([&] {
        (void)hipFuncSetAttribute((const void*)conv_rgb_out_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_rgb_out_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    const dim3 grid((unsigned)((size_t)N * a.bands * a.strips));
    if (C % 128 == 0) hipLaunchKernelGGL(conv_rgb_out_kernel<8>, grid, dim3(RO_NT), lds, stream, a);
    else hipLaunchKernelGGL(conv_rgb_out_kernel<4>, grid, dim3(RO_NT), lds, stream, a);
    return pesr_launch_status();
}

int pesr_conv_rgb_out_fwd_launch(const float* x, const float* w, const float* bias, float* y, int N, int H, int W, int C, int act,
                                 float slope, hipStream_t stream) {
    return rgb_out_launch(x, w, bias, y, N, H, W, C, act, slope, 0, stream);
}

// Input gradient of a 3 -> C conv = the C -> 3 conv of dy with the flipped, transposed kernel: the same HBM-bound kernel, reading
// the forward conv's OIHW [C][3][3][3] weights through the transposing index (no pack, no padded MFMAs).
int pesr_conv_rgb_in_dgrad_launch(const float* dy, const float* w, float* dx, int N, int H, int W, int C, hipStream_t stream) {
    return rgb_out_launch(dy, w, nullptr, dx, N, H, W, C, PESR_ACT_NONE, 0.f, 1, stream);
}
