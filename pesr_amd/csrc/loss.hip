// Image-space and feature-space losses of the GAN step, fused forward + gradient (gfx950, HBM-bound).
//   l1_tv : reference train.py:131,137-140,240-242 on sr/hr [N][H][W][3] (NHWC):
//             l1 = mean |sr - hr| ; tv = SUM |sr[..,x]-sr[..,x+1]| + SUM |sr[y]-sr[y+1]|   (a sum)
//             grad = a_l1 * sign(sr-hr)/numel + a_tv * d(tv)/d(sr)
//   mse   : reference train.py:134-136 F.mse_loss(vgg_sr, vgg_hr): mean (a-b)^2, grad_a = g * 2 (a-b)/numel
// Sums go through per-block fp32 partials and a fixed-order double finalize (deterministic).
#include "common.h"
#include "launchers.h"

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
__global__ void l1_tv_final_kernel(const double* __restrict__ dsum, float* __restrict__ out, double inv_numel) {
    if (threadIdx.x < 2) out[threadIdx.x] = (float)(threadIdx.x == 0 ? dsum[0] * inv_numel : dsum[1]);   // L1 mean, TV sum
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
__global__ void mse_final_kernel(const double* __restrict__ dsum, float* __restrict__ out, double inv_numel) {
    if (threadIdx.x == 0) out[0] = (float)(dsum[0] * inv_numel);
}

int pesr_loss_l1_tv_launch(const float* sr, const float* hr, float* grad, float* out2, int N, int H, int W, float g_l1, float g_tv,
                           void* ws, size_t ws_bytes, hipStream_t stream) {
    const int nb = 1024;
    if (!ws || ws_bytes < 64 + (size_t)nb * 2 * sizeof(float)) return PESR_EWORKSPACE;
    double* dsum = (double*)ws;
    float* part = (float*)((char*)ws + 64);
    hipLaunchKernelGGL(l1_tv_kernel, dim3(nb), dim3(256), 0, stream, sr, hr, grad, part, N, H, W, g_l1, g_tv);
    const int rc = pesr_reduce_rows_launch(part, dsum, nb, 2, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(l1_tv_final_kernel, dim3(1), dim3(64), 0, stream, (const double*)dsum, out2, 1.0 / ((double)N * H * W * 3));
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
    const int rc = pesr_reduce_rows_launch(part, dsum, nb, 1, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(mse_final_kernel, dim3(1), dim3(64), 0, stream, (const double*)dsum, out1, 1.0 / (double)n);
    return pesr_launch_status();
}
