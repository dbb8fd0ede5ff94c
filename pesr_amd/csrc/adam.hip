// Fused Adam over one flat fp32 parameter buffer (reference train.py:124-125: optim.Adam, betas (0.9, 0.999),
// eps 1e-8, no weight decay, no amsgrad).  Follows torch's single-tensor update order:
//   m = lerp(m, g, 1-b1) ; v = b2*v + (1-b2)*g*g ; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
// HBM-bound: reads p, g, m, v and writes p, m, v once (28 B per parameter).
#include "common.h"
#include "launchers.h"

__device__ __forceinline__ void adam_update(f32x4& pp, f32x4& mm, f32x4& vv, f32x4 gg, float one_minus_b1, float b2, float one_minus_b2,
                                            float step_size, float bc2_sqrt, float eps, float gscale) {
    gg = gg * gscale;
    mm = mm + (gg - mm) * one_minus_b1;
    vv = vv * b2 + gg * gg * one_minus_b2;
    f32x4 den;
    den.x = sqrtf(vv.x) / bc2_sqrt + eps; den.y = sqrtf(vv.y) / bc2_sqrt + eps;
    den.z = sqrtf(vv.z) / bc2_sqrt + eps; den.w = sqrtf(vv.w) / bc2_sqrt + eps;
    pp.x -= step_size * (mm.x / den.x); pp.y -= step_size * (mm.y / den.y);
    pp.z -= step_size * (mm.z / den.z); pp.w -= step_size * (mm.w / den.w);
}
__global__ void adam_kernel(f32x4* __restrict__ p, const f32x4* __restrict__ g, f32x4* __restrict__ m, f32x4* __restrict__ v, long n4,
                            float one_minus_b1, float b2, float one_minus_b2, float step_size, float bc2_sqrt, float eps, float gscale) {
    const long stride = (long)gridDim.x * blockDim.x;
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; e + stride < n4; e += 2 * stride) {             // two elements per pass, the gradient (read once, dead afterwards) non-temporal:
        const long e1 = e + stride;                        // 226 -> 211 us for the Generator's 43 M parameters (round 6)
        f32x4 g0 = __builtin_nontemporal_load(g + e), g1 = __builtin_nontemporal_load(g + e1);
        f32x4 m0 = m[e], v0 = v[e], p0 = p[e], m1 = m[e1], v1 = v[e1], p1 = p[e1];
        adam_update(p0, m0, v0, g0, one_minus_b1, b2, one_minus_b2, step_size, bc2_sqrt, eps, gscale);
        adam_update(p1, m1, v1, g1, one_minus_b1, b2, one_minus_b2, step_size, bc2_sqrt, eps, gscale);
        p[e] = p0; m[e] = m0; v[e] = v0; p[e1] = p1; m[e1] = m1; v[e1] = v1;
    }
    for (; e < n4; e += stride) {
        f32x4 mm = m[e], vv = v[e], pp = p[e];
        adam_update(pp, mm, vv, g[e], one_minus_b1, b2, one_minus_b2, step_size, bc2_sqrt, eps, gscale);
        p[e] = pp; m[e] = mm; v[e] = vv;
    }
}

int pesr_adam_launch(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps, int step,
                     float gscale, hipStream_t stream) {
    if (n % 4 || step < 1) return PESR_EINVAL;
    const double bc1 = 1.0 - pow((double)b1, (double)step);
    const double bc2 = 1.0 - pow((double)b2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    const long n4 = n / 4;
    const int grid = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
    hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, stream, (f32x4*)p, (const f32x4*)g, (f32x4*)m, (f32x4*)v, n4, 1.0f - b1, b2,
                       1.0f - b2, step_size, bc2_sqrt, eps, gscale);
    return pesr_launch_status();
}

// ---- the same update with its step-dependent scalars read from device memory ------------------------------------------------
// For hipGraph replay (Trainer.capture_gan_step): a captured launch replays its kernel arguments, so the step count and the
// learning rate must not be among them.  state[0] = learning rate (written by the host between replays), state[1] = step count
// (as a float-exact integer < 2^24 would overflow in long runs: kept as two 32-bit halves in state[4..5]), state[2] = lr / bc1,
// state[3] = sqrt(bc2).  adam_tick_kernel advances the count and refreshes state[2..3] in double, like the host path.
__global__ void adam_tick_kernel(float* __restrict__ state, float b1, float b2) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    unsigned long long t = ((unsigned long long)__float_as_uint(state[5]) << 32) | __float_as_uint(state[4]);
    t += 1;
    state[4] = __uint_as_float((unsigned)(t & 0xffffffffu));
    state[5] = __uint_as_float((unsigned)(t >> 32));
    const double bc1 = 1.0 - pow((double)b1, (double)t);
    const double bc2 = 1.0 - pow((double)b2, (double)t);
    state[2] = (float)((double)state[0] / bc1);
    state[3] = (float)sqrt(bc2);
}
__global__ void adam_dev_kernel(f32x4* __restrict__ p, const f32x4* __restrict__ g, f32x4* __restrict__ m, f32x4* __restrict__ v, long n4,
                                float one_minus_b1, float b2, float one_minus_b2, const float* __restrict__ state, float eps, float gscale) {
    const float step_size = state[2], bc2_sqrt = state[3];
    const long stride = (long)gridDim.x * blockDim.x;
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; e + stride < n4; e += 2 * stride) {
        const long e1 = e + stride;
        f32x4 g0 = __builtin_nontemporal_load(g + e), g1 = __builtin_nontemporal_load(g + e1);
        f32x4 m0 = m[e], v0 = v[e], p0 = p[e], m1 = m[e1], v1 = v[e1], p1 = p[e1];
        adam_update(p0, m0, v0, g0, one_minus_b1, b2, one_minus_b2, step_size, bc2_sqrt, eps, gscale);
        adam_update(p1, m1, v1, g1, one_minus_b1, b2, one_minus_b2, step_size, bc2_sqrt, eps, gscale);
        p[e] = p0; m[e] = m0; v[e] = v0; p[e1] = p1; m[e1] = m1; v[e1] = v1;
    }
    for (; e < n4; e += stride) {
        f32x4 mm = m[e], vv = v[e], pp = p[e];
        adam_update(pp, mm, vv, g[e], one_minus_b1, b2, one_minus_b2, step_size, bc2_sqrt, eps, gscale);
        p[e] = pp; m[e] = mm; v[e] = vv;
    }
}

int pesr_adam_dev_launch(float* p, const float* g, float* m, float* v, long n, float* state, float b1, float b2, float eps, float gscale,
                         hipStream_t stream) {
    if (n % 4 || !state) return PESR_EINVAL;
    const long n4 = n / 4;
    const int grid = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(64), 0, stream, state, b1, b2);
    hipLaunchKernelGGL(adam_dev_kernel, dim3(grid), dim3(256), 0, stream, (f32x4*)p, (const f32x4*)g, (f32x4*)m, (f32x4*)v, n4, 1.0f - b1, b2,
                       1.0f - b2, (const float*)state, eps, gscale);
    return pesr_launch_status();
}
