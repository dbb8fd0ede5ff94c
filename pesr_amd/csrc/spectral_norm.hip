// Spectral normalisation of a conv weight (torch.nn.utils.spectral_norm semantics, one power iteration), gfx950.
//
// Reference model/basic.py:25 wraps the Discriminator's convs in `spectral_norm(...)` when --spectral_norm true - an undefined
// name there (NameError); the evident intent is torch.nn.utils.spectral_norm, whose algorithm this file restates for
// the weight viewed as a matrix W [O][K] (K = Cin * 9, the OIHW tensor as it lies in memory):
//   training:  v <- normalize(W^T u),  u <- normalize(W v)   (in place, eps 1e-12: x / max(||x||, eps))
//   always:    sigma = u^T W v,   W_hat = W / sigma
//   backward:  dW = (G - <G, W_hat> u v^T) / sigma          (u, v constants, as in torch: they are clones made under no_grad)
// All reductions are fixed-order (per-block partials summed in index order), in double where terms cancel; no atomics.  The
// matrices are at most 512 x 4608 (9.4 MB): HBM-trivial, latency-bound - four / three small launches.
#include "common.h"
#include "launchers.h"

namespace {
constexpr int SN_NT = 256;

__device__ __forceinline__ double block_sum_d(double v, double* red) {      // fixed order: waves 0..3
    const double w = wave_sum_d(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = w;
    __syncthreads();
    return ((red[0] + red[1]) + red[2]) + red[3];
}

// t[j] = sum_i W[i][j] u[i]  (one thread per column j: coalesced rows); part[block] = sum_j t[j]^2 of the block's columns
__global__ __launch_bounds__(SN_NT) void sn_wtu_kernel(const float* __restrict__ W, const float* __restrict__ u, float* __restrict__ t,
                                                       double* __restrict__ part, int O, int K) {
    __shared__ double red[4];
    const int j = blockIdx.x * SN_NT + threadIdx.x;
    double s = 0.0;
    if (j < K) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int i = 0;
        for (; i + 4 <= O; i += 4) {
            a0 = fmaf(W[(size_t)i * K + j], u[i], a0); a1 = fmaf(W[(size_t)(i + 1) * K + j], u[i + 1], a1);
            a2 = fmaf(W[(size_t)(i + 2) * K + j], u[i + 2], a2); a3 = fmaf(W[(size_t)(i + 3) * K + j], u[i + 3], a3);
        }
        for (; i < O; ++i) a0 = fmaf(W[(size_t)i * K + j], u[i], a0);
        const float tj = (a0 + a1) + (a2 + a3);
        t[j] = tj;
        s = (double)tj * (double)tj;
    }
    const double b = block_sum_d(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = b;
}

// one block per row i: s[i] = sum_j W[i][j] vn[j] with vn = t / max(||t||, eps) (update) or vn = v (no update);
// the blocks also write the new v, each its share of the columns
__global__ __launch_bounds__(SN_NT) void sn_wv_kernel(const float* __restrict__ W, const float* __restrict__ t, const double* __restrict__ part,
                                                      int nparts, float* __restrict__ v, float* __restrict__ s, int O, int K, int update,
                                                      float eps) {
    __shared__ double red[4];
    float inv = 1.0f;
    if (update) {
        double n2 = 0.0;
        for (int k = 0; k < nparts; ++k) n2 += part[k];
        const float nrm = (float)sqrt(n2);
        inv = 1.0f / fmaxf(nrm, eps);
    }
    const float* src = update ? t : v;
    const float* row = W + (size_t)blockIdx.x * K;
    double acc = 0.0;
    for (int j = threadIdx.x; j < K; j += SN_NT) acc += (double)row[j] * (double)(src[j] * inv);
    const double tot = block_sum_d(acc, red);
    if (threadIdx.x == 0) s[blockIdx.x] = (float)tot;
    if (update) {   // v <- normalize(W^T u): block i writes columns [i * chunk, (i + 1) * chunk)
        const int chunk = (K + O - 1) / O;
        for (int j = blockIdx.x * chunk + threadIdx.x; j < K && j < (blockIdx.x + 1) * chunk; j += SN_NT) v[j] = t[j] * inv;
    }
}

// u <- normalize(s) (update), sigma = sum_i u[i] s[i]
__global__ __launch_bounds__(1024) void sn_finish_kernel(const float* __restrict__ s, float* __restrict__ u, float* __restrict__ sigma,
                                                         int O, int update, float eps) {
    __shared__ double red[16];
    __shared__ float inv_sh;
    const int i = threadIdx.x;
    if (update) {
        double q = i < O ? (double)s[i] * (double)s[i] : 0.0;
        q = wave_sum_d(q);
        if ((i & 63) == 0) red[i >> 6] = q;
        __syncthreads();
        if (i == 0) {
            double n2 = 0.0;
            for (int k = 0; k < 16; ++k) n2 += red[k];
            inv_sh = 1.0f / fmaxf((float)sqrt(n2), eps);
        }
        __syncthreads();
        if (i < O) u[i] = s[i] * inv_sh;
        __syncthreads();
    }
    double d = i < O ? (double)u[i] * (double)s[i] : 0.0;
    d = wave_sum_d(d);
    __syncthreads();
    if ((i & 63) == 0) red[i >> 6] = d;
    __syncthreads();
    if (i == 0) {
        double tot = 0.0;
        for (int k = 0; k < 16; ++k) tot += red[k];
        sigma[0] = (float)tot;
    }
}

__global__ __launch_bounds__(SN_NT) void sn_scale_kernel(const f32x4* __restrict__ W, const float* __restrict__ sigma, f32x4* __restrict__ out,
                                                         long n4) {
    const float sg = sigma[0];
    for (long e = (long)blockIdx.x * SN_NT + threadIdx.x; e < n4; e += (long)gridDim.x * SN_NT) {
        const f32x4 w = W[e];
        out[e] = (f32x4){w.x / sg, w.y / sg, w.z / sg, w.w / sg};
    }
}

// part[block] = sum over the block's elements of G * W_hat
__global__ __launch_bounds__(SN_NT) void sn_dot_kernel(const f32x4* __restrict__ G, const f32x4* __restrict__ Wh, double* __restrict__ part,
                                                       long n4) {
    __shared__ double red[4];
    double acc = 0.0;
    for (long e = (long)blockIdx.x * SN_NT + threadIdx.x; e < n4; e += (long)gridDim.x * SN_NT) {
        const f32x4 g = G[e], w = Wh[e];
        acc += ((double)g.x * w.x + (double)g.y * w.y) + ((double)g.z * w.z + (double)g.w * w.w);
    }
    const double b = block_sum_d(acc, red);
    if (threadIdx.x == 0) part[blockIdx.x] = b;
}

// dW[i][j] = (G[i][j] - c u[i] v[j]) / sigma, c = sum of the partials (fixed order)
__global__ __launch_bounds__(SN_NT) void sn_grad_kernel(const float* __restrict__ G, const float* __restrict__ u, const float* __restrict__ v,
                                                        const float* __restrict__ sigma, const double* __restrict__ part, int nparts,
                                                        float* __restrict__ dW, int O, int K, int accumulate) {
    double cd = 0.0;
    for (int k = 0; k < nparts; ++k) cd += part[k];
    const float c = (float)cd, sg = sigma[0];
    const long n = (long)O * K;
    for (long e = (long)blockIdx.x * SN_NT + threadIdx.x; e < n; e += (long)gridDim.x * SN_NT) {
        const int i = (int)(e / K), j = (int)(e - (long)i * K);
        const float val = (G[e] - c * u[i] * v[j]) / sg;
        dW[e] = accumulate ? dW[e] + val : val;
    }
}
}  // namespace

size_t pesr_spectral_norm_ws_bytes(int O, int K) {
    return (size_t)K * sizeof(float) + (size_t)O * sizeof(float) + 1024 * sizeof(double) + 512;
}

int pesr_spectral_norm_fwd_launch(const float* W, float* u, float* v, float* w_hat, float* sigma, int O, int K, int update, float eps,
                                  void* ws, size_t ws_bytes, hipStream_t stream) {
    if (O < 1 || O > 1024 || K < 4 || (((long)O * K) & 3)) return PESR_EINVAL;
    if (!ws || ws_bytes < pesr_spectral_norm_ws_bytes(O, K)) return PESR_EWORKSPACE;
    double* part = (double*)ws;                                         // [<= 1024]
    float* t = (float*)((char*)ws + 1024 * sizeof(double));             // [K]
    float* s = t + K;                                                   // [O]
    const int nb = (K + SN_NT - 1) / SN_NT;
    if (nb > 1024) return PESR_EINVAL;
    if (update) hipLaunchKernelGGL(sn_wtu_kernel, dim3(nb), dim3(SN_NT), 0, stream, W, (const float*)u, t, part, O, K);
    hipLaunchKernelGGL(sn_wv_kernel, dim3(O), dim3(SN_NT), 0, stream, W, (const float*)t, (const double*)part, nb, v, s, O, K, update, eps);
    hipLaunchKernelGGL(sn_finish_kernel, dim3(1), dim3(1024), 0, stream, (const float*)s, u, sigma, O, update, eps);
    const long n4 = (long)O * K / 4;
    const int grid = (int)((n4 + SN_NT - 1) / SN_NT < 2048 ? (n4 + SN_NT - 1) / SN_NT : 2048);
    hipLaunchKernelGGL(sn_scale_kernel, dim3(grid), dim3(SN_NT), 0, stream, (const f32x4*)W, (const float*)sigma, (f32x4*)w_hat, n4);
    return pesr_launch_status();
}

int pesr_spectral_norm_bwd_launch(const float* G, const float* w_hat, const float* u, const float* v, const float* sigma, float* dW, int O,
                                  int K, int accumulate, void* ws, size_t ws_bytes, hipStream_t stream) {
    if (O < 1 || O > 1024 || K < 4 || (((long)O * K) & 3)) return PESR_EINVAL;
    if (!ws || ws_bytes < pesr_spectral_norm_ws_bytes(O, K)) return PESR_EWORKSPACE;
    double* part = (double*)ws;
    const long n4 = (long)O * K / 4;
    const int nb = (int)((n4 + SN_NT - 1) / SN_NT < 512 ? (n4 + SN_NT - 1) / SN_NT : 512);
    hipLaunchKernelGGL(sn_dot_kernel, dim3(nb), dim3(SN_NT), 0, stream, (const f32x4*)G, (const f32x4*)w_hat, part, n4);
    const long n = (long)O * K;
    const int grid = (int)((n + SN_NT - 1) / SN_NT < 4096 ? (n + SN_NT - 1) / SN_NT : 4096);
    hipLaunchKernelGGL(sn_grad_kernel, dim3(grid), dim3(SN_NT), 0, stream, G, u, v, sigma, (const double*)part, nb, dW, O, K, accumulate);
    return pesr_launch_status();
}
