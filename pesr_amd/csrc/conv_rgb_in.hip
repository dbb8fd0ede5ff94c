// Forward 3x3 conv (pad 1, stride 1) from a 3-channel input: reference `embed` (model/pesr.py:23),
// Discriminator features.0.0 (model/pesr.py:53) and vgg19 features.0 (model/vgg.py:8).  K = 27 only, so this
// is HBM-bound on writing the C-channel output (151 MB for C = 64 at 16 x 192 x 192); the zero-padded MFMA
// path spends 5x the time on padding.  One thread owns 4 consecutive output channels (108 weights in VGPRs,
// loaded once) and walks pixels; the 16 (C/4) threads of a pixel share its 27 input values through L1.
// Accumulation order per output: bias + taps in (ky, kx, ci) order - a plain fmaf chain.
#include "common.h"
#include "launchers.h"

template <int ACT>
__global__ __launch_bounds__(256) void conv_rgb_in_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ y, int N, int H,
                                                          int W, int C, float slope) {
    // w: OIHW [C][3][3][3]
    const int C4 = C >> 2;
    const int cg = threadIdx.x % C4;                 // channel group (4 channels)
    const int pl = threadIdx.x / C4;                 // pixel lane within the block
    const int ppb = 256 / C4;                        // pixels per block iteration
    float wr[4][27];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int k = 0; k < 27; ++k) {
            // k = (ky*3+kx)*3 + ci  <-  OIHW index ((co*3 + ci)*3 + ky)*3 + kx
            const int ci = k % 3, t = k / 3;
            wr[q][k] = w[((cg * 4 + q) * 3 + ci) * 9 + t];
        }
    float br[4] = {0.f, 0.f, 0.f, 0.f};
    if (bias) {
#pragma unroll
        for (int q = 0; q < 4; ++q) br[q] = bias[cg * 4 + q];
    }
    const long total = (long)N * H * W;
    for (long p = (long)blockIdx.x * ppb + pl; p < total; p += (long)gridDim.x * ppb) {
        if (pl >= ppb) break;
        const int xx = (int)(p % W);
        const int yy = (int)((p / W) % H);
        const long n = p / ((long)W * H);
        const float* xi = x + n * (long)H * W * 3;
        float in[27];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int iy = yy + ky - 1, ix = xx + kx - 1;
                const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
                const float* q = xi + ((long)(ok ? iy : 0) * W + (ok ? ix : 0)) * 3;
                const float v0 = q[0], v1 = q[1], v2 = q[2];
                in[(ky * 3 + kx) * 3 + 0] = ok ? v0 : 0.f;
                in[(ky * 3 + kx) * 3 + 1] = ok ? v1 : 0.f;
                in[(ky * 3 + kx) * 3 + 2] = ok ? v2 : 0.f;
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
        *(f32x4*)(y + p * C + cg * 4) = o;
    }
}

int pesr_conv_rgb_in_launch(const float* x, const float* w, const float* bias, float* y, int N, int H, int W, int C, int act,
                            float slope, hipStream_t stream) {
    if (C % 4 || C > 1024 || 256 % (C / 4)) return PESR_EINVAL;
    const int ppb = 256 / (C / 4);
    const long total = (long)N * H * W;
    long grid = (total + ppb - 1) / ppb;
    if (grid > 256 * 8) grid = 256 * 8;
    if (act == PESR_ACT_RELU) hipLaunchKernelGGL(conv_rgb_in_kernel<PESR_ACT_RELU>, dim3((unsigned)grid), dim3(256), 0, stream, x, w, bias, y, N, H, W, C, slope);
    else if (act == PESR_ACT_LRELU) hipLaunchKernelGGL(conv_rgb_in_kernel<PESR_ACT_LRELU>, dim3((unsigned)grid), dim3(256), 0, stream, x, w, bias, y, N, H, W, C, slope);
    else hipLaunchKernelGGL(conv_rgb_in_kernel<PESR_ACT_NONE>, dim3((unsigned)grid), dim3(256), 0, stream, x, w, bias, y, N, H, W, C, slope);
    return pesr_launch_status();
}
