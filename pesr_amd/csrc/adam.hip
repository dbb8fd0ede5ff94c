// Fused Adam over one flat fp32 parameter buffer (reference train.py:124-125: optim.Adam, betas (0.9, 0.999),
// eps 1e-8, no weight decay, no amsgrad).  Follows torch's single-tensor update order:
//   m = lerp(m, g, 1-b1) ; v = b2*v + (1-b2)*g*g ; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
// HBM-bound: reads p, g, m, v and writes p, m, v once (28 B per parameter).
#include "common.h"
#include "launchers.h"

__global__ void adam_kernel(f32x4* __restrict__ p, const f32x4* __restrict__ g, f32x4* __restrict__ m, f32x4* __restrict__ v, long n4,
                            float one_minus_b1, float b2, float one_minus_b2, float step_size, float bc2_sqrt, float eps, float gscale) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (long)gridDim.x * blockDim.x) {
        const f32x4 gg = g[e] * gscale;
        f32x4 mm = m[e], vv = v[e], pp = p[e];
        mm = mm + (gg - mm) * one_minus_b1;
        vv = vv * b2 + gg * gg * one_minus_b2;
        f32x4 den;
        den.x = sqrtf(vv.x) / bc2_sqrt + eps; den.y = sqrtf(vv.y) / bc2_sqrt + eps;
        den.z = sqrtf(vv.z) / bc2_sqrt + eps; den.w = sqrtf(vv.w) / bc2_sqrt + eps;
        pp.x -= step_size * (mm.x / den.x); pp.y -= step_size * (mm.y / den.y);
        pp.z -= step_size * (mm.z / den.z); pp.w -= step_size * (mm.w / den.w);
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
