// Image-space and feature-space losses of the GAN step, fused forward + gradient (gfx950, HBM-bound).
//   l1_tv : reference train.py:131,137-140,240-242 on sr/hr [N][H][W][3] (NHWC):
//             l1 = mean |sr - hr| ; tv = SUM |sr[..,x]-sr[..,x+1]| + SUM |sr[y]-sr[y+1]|   (a sum)
//             grad = a_l1 * sign(sr-hr)/numel + a_tv * d(tv)/d(sr)
//   mse   : reference train.py:134-136 F.mse_loss(vgg_sr, vgg_hr): mean (a-b)^2, grad_a = g * 2 (a-b)/numel
// Sums go through per-block fp32 partials and a fixed-order double finalize (deterministic).
#include "common.h"
#include "launchers.h"
#include "reduce_rows.h"

__device__ __forceinline__ float sgn(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

__global__ __launch_bounds__(256) void l1_tv_kernel(const float* __restrict__ sr, const float* __restrict__ hr, float* __restrict__ grad,
                                                    float* __restrict__ part, int N, int H, int W, float g_l1, float g_tv) {
    // one thread per pixel (3 channels); g_l1 = upstream * alpha_l1 / numel, g_tv = upstream * alpha_tv
    const long total = (long)N * H * W;
    float s_l1 = 0.f, s_tv = 0.f;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int x = (int)(e % W);
        const int y = (int)((e / W) % H);
        const float* p = sr + e * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = p[c];
            const float d = v - hr[e * 3 + c];
            s_l1 += fabsf(d);
            float g = g_l1 * sgn(d);
            float t = 0.f;
            if (x + 1 < W) { const float q = v - p[3 + c]; s_tv += fabsf(q); t += sgn(q); }
            if (x > 0) { t -= sgn(p[c - 3] - v); }
            if (y + 1 < H) { const float q = v - p[(long)W * 3 + c]; s_tv += fabsf(q); t += sgn(q); }
            if (y > 0) { t -= sgn(p[c - (long)W * 3] - v); }
            if (grad) grad[e * 3 + c] = g + g_tv * t;
        }
    }
    __shared__ float red[4][2];
    const float a = wave_sum(s_l1), b = wave_sum(s_tv);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = a; red[threadIdx.x >> 6][1] = b; }
    __syncthreads();
    if (threadIdx.x < 2) part[blockIdx.x * 2 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
// second stage (fixed-order row reduce, reduce_rows.h) and the scalar finish in one launch
__global__ __launch_bounds__(1024) void l1_tv_final_kernel(const float* __restrict__ part, int nb, float* __restrict__ out, double inv_numel) {
    __shared__ double red[16][64];
    const double t = reduce_rows_block(part, nb, 2, threadIdx.x & 63, (threadIdx.x & 63) < 2, red);
    if (threadIdx.x < 2) out[threadIdx.x] = (float)(threadIdx.x == 0 ? t * inv_numel : t);   // L1 mean, TV sum
}

__global__ __launch_bounds__(256) void mse_kernel(const f32x4* __restrict__ a, const f32x4* __restrict__ b, f32x4* __restrict__ grad,
                                                  float* __restrict__ part, long n4, float gscale) {
    float s = 0.f;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (long)gridDim.x * blockDim.x) {
        const f32x4 d = a[e] - b[e];
        s += d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w;
        if (grad) grad[e] = d * gscale;
    }
    __shared__ float red[4];
    const float w = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(1024) void mse_final_kernel(const float* __restrict__ part, int nb, float* __restrict__ out, double inv_numel) {
    __shared__ double red[16][64];
    const double t = reduce_rows_block(part, nb, 1, 0, (threadIdx.x & 63) == 0, red);
    if (threadIdx.x == 0) out[0] = (float)(t * inv_numel);
}

int pesr_loss_l1_tv_launch(const float* sr, const float* hr, float* grad, float* out2, int N, int H, int W, float g_l1, float g_tv,
                           void* ws, size_t ws_bytes, hipStream_t stream) {
    const int nb = 1024;
    if (!ws || ws_bytes < 64 + (size_t)nb * 2 * sizeof(float)) return PESR_EWORKSPACE;
    double* dsum = (double*)ws;
    float* part = (float*)((char*)ws + 64);
    hipLaunchKernelGGL(l1_tv_kernel, dim3(nb), dim3(256), 0, stream, sr, hr, grad, part, N, H, W, g_l1, g_tv);
    hipLaunchKernelGGL(l1_tv_final_kernel, dim3(1), dim3(1024), 0, stream, (const float*)part, nb, out2, 1.0 / ((double)N * H * W * 3));
    return pesr_launch_status();
}
int pesr_loss_mse_launch(const float* a, const float* b, float* grad, float* out1, long n, float gscale, void* ws, size_t ws_bytes,
                         hipStream_t stream) {
    if (n % 4) return PESR_EINVAL;
    const int nb = 512;
    if (!ws || ws_bytes < 64 + (size_t)nb * sizeof(float)) return PESR_EWORKSPACE;
    double* dsum = (double*)ws;
    float* part = (float*)((char*)ws + 64);
    hipLaunchKernelGGL(mse_kernel, dim3(nb), dim3(256), 0, stream, (const f32x4*)a, (const f32x4*)b, (f32x4*)grad, part, n / 4, gscale);
    hipLaunchKernelGGL(mse_final_kernel, dim3(1), dim3(1024), 0, stream, (const float*)part, nb, out1, 1.0 / (double)n);
    return pesr_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Validation PSNR on the device (reference utils.py:10-18,32-41): both images are clipped to 0..255 and rounded to
// uint8 values, converted to the BT.601 luma y = (65.738 r + 129.057 g + 25.064 b)/256 + 16 in DOUBLE (as numpy does),
// clipped and rounded again, and the squared differences are summed.  Every term is an integer-valued double, so the sum
// is exact whatever its order: the result is bit-identical to the host computation.
// Element (n = 0, c, y, x) of a logical NCHW tensor lives at c*sc + (y*W + x)*sp.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double luma_u8(const float* p, long sc) {
    const double r = rint(fmin(fmax((double)p[0], 0.0), 255.0));
    const double g = rint(fmin(fmax((double)p[sc], 0.0), 255.0));
    const double b = rint(fmin(fmax((double)p[2 * sc], 0.0), 255.0));
    const double y = ((r * (65.738 / 256) + g * (129.057 / 256)) + b * (25.064 / 256)) + 16.0;
    return rint(fmin(fmax(y, 0.0), 255.0));
}
__global__ __launch_bounds__(256) void psnr_y_kernel(const float* __restrict__ a, const float* __restrict__ b, double* __restrict__ part,
                                                     long P, long asc, long asp, long bsc, long bsp) {
    double s = 0.0;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (long)gridDim.x * blockDim.x) {
        const double d = luma_u8(a + p * asp, asc) - luma_u8(b + p * bsp, bsc);
        s += d * d;
    }
    __shared__ double red[4];
    const double w = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}
__global__ void psnr_final_kernel(const double* __restrict__ part, double* __restrict__ out, int nb, double inv_n) {
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int k = 0; k < nb; ++k) s += part[k];
        const double mse = s * inv_n;
        out[0] = mse;
        out[1] = 20.0 * log10(255.0 / sqrt(mse));
    }
}
int pesr_psnr_y_launch(const float* a, const float* b, double* out2, int H, int W, int a_nhwc, int b_nhwc, void* ws, size_t ws_bytes,
                       hipStream_t stream) {
    const int nb = 256;
    if (!ws || ws_bytes < (size_t)nb * sizeof(double)) return PESR_EWORKSPACE;
    const long P = (long)H * W;
    hipLaunchKernelGGL(psnr_y_kernel, dim3(nb), dim3(256), 0, stream, a, b, (double*)ws, P, a_nhwc ? 1L : P, a_nhwc ? 3L : 1L,
                       b_nhwc ? 1L : P, b_nhwc ? 3L : 1L);
    hipLaunchKernelGGL(psnr_final_kernel, dim3(1), dim3(64), 0, stream, (const double*)ws, out2, nb, 1.0 / (double)P);
    return pesr_launch_status();
}

// ------------------------------------------------------------------------------------------------
// GAN losses on the [B, 1] logits (reference train.py:132-133,210-213,244-253 and model/focal_loss.py:9-13): value and the
// gradients w.r.t. both logit vectors in ONE launch instead of ~25 scalar-sized ATen kernels per loss (sigmoid, log_sigmoid, sub,
// mul, mean, their backward ...).  B <= 1024, one block.
//   side 0 (discriminator): SGAN  BCE(r, 1) + BCE(f, 0)            RSGAN  BCE(r - f, 1)      RaSGAN [BCE(r - mean f, 1) + BCE(f - mean r, 0)] / 2
//   side 1 (generator):     SGAN  L(f, 1)                          RSGAN  L(f - r, 1)        RaSGAN [L(r - mean f, 0) + L(f - mean r, 1)] / 2
//   L = BCE-with-logits, or (focal = 1) the reference's FocalLoss with the torch-0.4 gradient through w = (1 - pt)^gamma AND the
//   BCE term: dL/dx = [dw/dx * bce + w * (p - t)] / B,  dw/dx = -gamma (1 - pt)^(gamma - 1) (2t - 1) p (1 - p).
// All means are over the B samples given (the data-parallel RaSGAN, whose batch means span the ranks, stays on the torch path).
// out[0] = scale * loss;  d_real / d_fake [B] = scale * dloss/dlogit (either may be NULL).
// ------------------------------------------------------------------------------------------------
namespace {
__device__ __forceinline__ float bce_logits(float x, float t) {              // as ATen: (1 - t) x + max(-x, 0) + log(exp(-m) + exp(-x - m))
    const float m = fmaxf(-x, 0.f);
    return (1.f - t) * x + m + logf(expf(-m) + expf(-x - m));
}
__device__ __forceinline__ void loss_term(float x, float t, int focal, float gamma, float* val, float* dx) {
    const float p = 1.f / (1.f + expf(-x));
    const float bce = bce_logits(x, t);
    if (!focal) { *val = bce; *dx = p - t; return; }
    const float pt = p * t + (1.f - p) * (1.f - t);
    const float w = powf(1.f - pt, gamma);
    const float dw = gamma == 0.f ? 0.f : -gamma * powf(1.f - pt, gamma - 1.f) * (2.f * t - 1.f) * p * (1.f - p);
    *val = w * bce;
    *dx = dw * bce + w * (p - t);
}
__device__ __forceinline__ double block_sum1024(double v, double* red) {     // fixed order over the 16 waves
    const double w = wave_sum_d(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = w;
    __syncthreads();
    double s = 0.0;
    for (int k = 0; k < 16; ++k) s += red[k];
    return s;
}
}  // namespace

__global__ __launch_bounds__(1024) void gan_loss_kernel(const float* __restrict__ pr, const float* __restrict__ pf, int B, int gan, int side,
                                                        int focal, float gamma, float scale, float* __restrict__ out,
                                                        float* __restrict__ d_real, float* __restrict__ d_fake) {
    __shared__ double red[16];
    const int i = threadIdx.x;
    const bool on = i < B;
    const float r = on ? pr[i] : 0.f, f = on ? pf[i] : 0.f;
    const float invB = 1.0f / (float)B;
    float val = 0.f, gr = 0.f, gf = 0.f;                   // per-sample loss term(s) and d(sum of terms)/d(r_i), d/d(f_i) before the 1/B
    const int fo = side == 1 ? focal : 0;                  // the discriminator side is plain BCE in the reference
    if (gan == 0) {                                        // SGAN
        if (side == 0) {
            float v1, g1, v2, g2;
            loss_term(r, 1.f, 0, 0.f, &v1, &g1); loss_term(f, 0.f, 0, 0.f, &v2, &g2);
            val = v1 + v2; gr = g1; gf = g2;
        } else { loss_term(f, 1.f, fo, gamma, &val, &gf); }
    } else if (gan == 1) {                                 // RSGAN
        float g;
        if (side == 0) { loss_term(r - f, 1.f, 0, 0.f, &val, &g); gr = g; gf = -g; }
        else { loss_term(f - r, 1.f, fo, gamma, &val, &g); gf = g; gr = -g; }
    }
    if (gan == 2) {                                        // RaSGAN: two passes (the means first)
        const float mr = (float)(block_sum1024(on ? (double)r : 0.0, red)) * invB;
        const float mf = (float)(block_sum1024(on ? (double)f : 0.0, red)) * invB;
        float va, ga, vb, gb;
        loss_term(r - mf, side == 0 ? 1.f : 0.f, fo, gamma, &va, &ga);
        loss_term(f - mr, side == 0 ? 0.f : 1.f, fo, gamma, &vb, &gb);
        if (!on) { va = vb = ga = gb = 0.f; }
        const float sa = (float)block_sum1024((double)ga, red), sb = (float)block_sum1024((double)gb, red);
        val = 0.5f * (va + vb);
        gr = 0.5f * (ga - sb * invB);                      // r_i also moves every b_j through mean r
        gf = 0.5f * (gb - sa * invB);
    }
    if (!on) val = 0.f;
    const double tot = block_sum1024((double)val, red);
    if (i == 0) out[0] = scale * (float)(tot * (double)invB);
    if (on) {
        if (d_real) d_real[i] = scale * gr * invB;
        if (d_fake) d_fake[i] = scale * gf * invB;
    }
}

int pesr_gan_loss_launch(const float* pred_real, const float* pred_fake, int B, int gan, int side, int focal, float gamma, float scale,
                         float* out, float* d_real, float* d_fake, hipStream_t stream) {
    if (B < 1 || B > 1024 || gan < 0 || gan > 2 || side < 0 || side > 1) return PESR_EINVAL;
    hipLaunchKernelGGL(gan_loss_kernel, dim3(1), dim3(1024), 0, stream, pred_real, pred_fake, B, gan, side, focal, gamma, scale, out, d_real,
                       d_fake);
    return pesr_launch_status();
}
