// 3x3 conv (pad 1, stride 1) FROM a 3-channel tensor: reference `embed` (model/pesr.py:23), Discriminator
// features.0.0 (model/pesr.py:53), vgg19 features.0 (model/vgg.py:8) - and, with the weights read transposed and
// flipped, the input gradient of the C -> 3 conv `upsample.4` (model/basic.py:60), which is the same operation on dy.
// K = 27 only, so this is HBM-bound on writing the C-channel result (151 MB for C = 64 at 16 x 192 x 192, 604 MB for
// C = 256); the zero-padded MFMA path spends 5x the time on padding.
//
// A workgroup owns an 8 x 32 pixel tile: the 10 x 34 halo goes to LDS once as [pixel][R, G, B, 0]; one thread owns 4
// consecutive output channels (108 weights in VGPRs, loaded once per kernel) and walks the tile's pixels, taking each
// tap's RGB with ONE broadcast ds_read_b128 (all channel groups of a pixel read the same address).  A wave stores
// 64 consecutive float4 = whole pixels, fully coalesced.
// Accumulation order per output: bias + taps in (ky, kx, ci) order - a plain fmaf chain.
#include "common.h"
#include "launchers.h"

#define RGB_TH 8
#define RGB_TW 32

// WMODE 0: w is OIHW [C][3][3][3] of this conv.   WMODE 1: w is OIHW [3][C][3][3] of the conv whose input gradient this is.
// STATS (round 4; the Discriminator's features.0 in front of its BatchNorm, reference model/basic.py:26-30): the workgroup also leaves
// the per-channel sum and sum of squares of what it stored in part[blockIdx.x][2][C] - the layout of bn_reduce_kernel<0>'s partial
// rows, so bn_finalize_kernel takes them as they are and the statistics need no pass of their own over the 151 MB result.  A thread
// adds its (at most a few dozen) pixels in fp32, everything across threads and workgroups is added in double, in a fixed order.
template <int ACT, int WMODE, bool STATS = false>
__global__ __launch_bounds__(256) void conv_rgb_in_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ y, int N, int H,
                                                          int W, int C, float slope, int tiles_x, int tiles_y,
                                                          float* __restrict__ part = nullptr) {
    constexpr int HW_ = RGB_TW + 2, HH_ = RGB_TH + 2;
    __shared__ f32x4 halo[HH_ * HW_];
    const int C4 = C >> 2;
    const int cg = threadIdx.x % C4;                 // channel group (4 channels)
    const int pl = threadIdx.x / C4;                 // pixel lane within the block
    const int ppb = 256 / C4;                        // pixel lanes
    float wr[4][27];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int k = 0; k < 27; ++k) {
            const int ci = k % 3, t = k / 3, oc = cg * 4 + q;       // k = (ky*3+kx)*3 + ci
            wr[q][k] = WMODE == 0 ? w[(oc * 3 + ci) * 9 + t] : w[((size_t)ci * C + oc) * 9 + (8 - t)];
        }
    float br[4] = {0.f, 0.f, 0.f, 0.f};
    if (bias) {
#pragma unroll
        for (int q = 0; q < 4; ++q) br[q] = bias[cg * 4 + q];
    }
    f32x4 st1 = {0.f, 0.f, 0.f, 0.f}, st2 = {0.f, 0.f, 0.f, 0.f};
    const int ntiles = N * tiles_y * tiles_x;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
        const int x0 = tx * RGB_TW, y0 = ty * RGB_TH;
        const float* xi = x + (size_t)n * H * W * 3;
        __syncthreads();                             // the previous tile's readers are done
        constexpr int HIT = (HH_ * HW_ + 255) / 256;  // both halo entries of a thread are loaded before either is stored (as a loop: one
        f32x4 hv[HIT];                                // dependent round trip per entry)
#pragma unroll
        for (int k = 0; k < HIT; ++k) {
            const int e = threadIdx.x + k * 256;
            const int hy = e / HW_, hx = e - hy * HW_;
            const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
            hv[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (e < HH_ * HW_ && iy >= 0 && iy < H && ix >= 0 && ix < W) {
                const float* q = xi + ((size_t)iy * W + ix) * 3;
                hv[k].x = q[0]; hv[k].y = q[1]; hv[k].z = q[2];
            }
        }
#pragma unroll
        for (int k = 0; k < HIT; ++k) {
            const int e = threadIdx.x + k * 256;
            if (e < HH_ * HW_) halo[e] = hv[k];
        }
        __syncthreads();
        if (pl < ppb) {
            for (int p = pl; p < RGB_TH * RGB_TW; p += ppb) {
                const int py = p / RGB_TW, px = p - py * RGB_TW;
                const int oy = y0 + py, ox = x0 + px;
                if (oy >= H || ox >= W) continue;
                float in[27];
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const f32x4 v = halo[(py + t / 3) * HW_ + px + t % 3];
                    in[t * 3 + 0] = v.x; in[t * 3 + 1] = v.y; in[t * 3 + 2] = v.z;
                }
                f32x4 o;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float s = br[q];
#pragma unroll
                    for (int k = 0; k < 27; ++k) s = fmaf(in[k], wr[q][k], s);
                    if (ACT == PESR_ACT_RELU) s = s > 0.f ? s : 0.f;
                    else if (ACT == PESR_ACT_LRELU) s = s > 0.f ? s : s * slope;
                    o[q] = s;
                }
                *(f32x4*)(y + (((size_t)n * H + oy) * W + ox) * C + cg * 4) = o;
                if (STATS) { st1 += o; st2 += o * o; }
            }
        }
    }
    if (STATS) {
        __shared__ f32x4 sred[2][256];
        sred[0][threadIdx.x] = st1; sred[1][threadIdx.x] = st2;          // (a thread past the pixel lanes holds zeros)
        __syncthreads();
        if (threadIdx.x < C4) {
            f64x4 d1 = {0.0, 0.0, 0.0, 0.0}, d2 = {0.0, 0.0, 0.0, 0.0};
            for (int k = 0; k < 256 / C4; ++k) {
                d1 += __builtin_convertvector(sred[0][k * C4 + cg], f64x4);
                d2 += __builtin_convertvector(sred[1][k * C4 + cg], f64x4);
            }
            f32x4* p = (f32x4*)part + (size_t)blockIdx.x * 2 * C4;
            p[cg] = __builtin_convertvector(d1, f32x4); p[C4 + cg] = __builtin_convertvector(d2, f32x4);
        }
    }
}

template <int WMODE>
static int rgb_in_launch(const float* x, const float* w, const float* bias, float* y, int N, int H, int W, int C, int act, float slope,
                         hipStream_t stream) {
    if (C % 4 || C > 1024 || 256 % (C / 4)) return PESR_EINVAL;
    const int tiles_x = (W + RGB_TW - 1) / RGB_TW, tiles_y = (H + RGB_TH - 1) / RGB_TH;
    long grid = (long)N * tiles_x * tiles_y;
    if (grid > 256 * 8) grid = 256 * 8;
    const dim3 g((unsigned)grid), b(256);
    if (act == PESR_ACT_RELU) hipLaunchKernelGGL((conv_rgb_in_kernel<PESR_ACT_RELU, WMODE>), g, b, 0, stream, x, w, bias, y, N, H, W, C, slope, tiles_x, tiles_y);
    else if (act == PESR_ACT_LRELU) hipLaunchKernelGGL((conv_rgb_in_kernel<PESR_ACT_LRELU, WMODE>), g, b, 0, stream, x, w, bias, y, N, H, W, C, slope, tiles_x, tiles_y);
    else hipLaunchKernelGGL((conv_rgb_in_kernel<PESR_ACT_NONE, WMODE>), g, b, 0, stream, x, w, bias, y, N, H, W, C, slope, tiles_x, tiles_y);
    return pesr_launch_status();
}

int pesr_conv_rgb_in_launch(const float* x, const float* w, const float* bias, float* y, int N, int H, int W, int C, int act,
                            float slope, hipStream_t stream) {
    return rgb_in_launch<0>(x, w, bias, y, N, H, W, C, act, slope, stream);
}

// rows of per-workgroup BatchNorm partials the call below writes ([rows][2][C] floats), 0: shape not covered
int pesr_conv_rgb_in_stats_rows(int N, int H, int W, int C) {
    if (C % 4 || C > 1024 || 256 % (C / 4) || N < 1 || H < 1 || W < 1) return 0;
    const long grid = (long)N * ((W + RGB_TW - 1) / RGB_TW) * ((H + RGB_TH - 1) / RGB_TH);
    return (int)(grid > 256 * 8 ? 256 * 8 : grid);
}

// y = conv3x3(x[N][H][W][3], w) (no bias, no activation) + part[rows][2][C] = per-workgroup sums / sums of squares of y
int pesr_conv_rgb_in_stats_launch(const float* x, const float* w, float* y, float* part, int N, int H, int W, int C, hipStream_t stream) {
    const int rows = pesr_conv_rgb_in_stats_rows(N, H, W, C);
    if (!rows || !part) return PESR_EINVAL;
    const int tiles_x = (W + RGB_TW - 1) / RGB_TW, tiles_y = (H + RGB_TH - 1) / RGB_TH;
    hipLaunchKernelGGL((conv_rgb_in_kernel<PESR_ACT_NONE, 0, true>), dim3((unsigned)rows), dim3(256), 0, stream, x, w, (const float*)nullptr, y,
                       N, H, W, C, 0.f, tiles_x, tiles_y, part);
    return pesr_launch_status();
}

// dx[N][H][W][C] of y = conv3x3(x, w[3][C][3][3]) given dy[N][H][W][3]
int pesr_conv_rgb_out_dgrad_launch(const float* dy, const float* w, float* dx, int N, int H, int W, int C, hipStream_t stream) {
    return rgb_in_launch<1>(dy, w, nullptr, dx, N, H, W, C, PESR_ACT_NONE, 0.f, stream);
}
