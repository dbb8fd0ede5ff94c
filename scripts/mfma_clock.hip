// Measurement tool (not product code): sustained fp32-input MFMA issue rate and the shader clock under that load on gfx950.
//   hipcc -O3 --offload-arch=gfx950 scripts/mfma_clock.hip -o exp/mfma_clock && exp/mfma_clock
// 256 workgroups x 8 waves (2 per SIMD) issue NACC independent v_mfma_f32_16x16x4_f32 per loop iteration.  One extra
// wave in workgroup 0 spins on a fixed-length scalar s_nop loop and times it with s_memrealtime (100 MHz): the ratio to
// the same loop on an idle chip gives the clock ratio under load.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(576) void mfma_loop(const float* __restrict__ src, float* __restrict__ out, long long* __restrict__ tim,
                                                 int iters, int spin, int mfma_on) {
    const int wave = threadIdx.x >> 6;
    if (wave == 8) {   // timing wave: scalar loop only
        if (blockIdx.x != 0) return;
        const long long r0 = __builtin_amdgcn_s_memrealtime();
        const long long c0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < spin; ++i) {
            asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15");
        }
        const long long c1 = __builtin_amdgcn_s_memtime();
        const long long r1 = __builtin_amdgcn_s_memrealtime();
        if ((threadIdx.x & 63) == 0) { tim[0] = r1 - r0; tim[1] = c1 - c0; }
        return;
    }
    if (!mfma_on) return;
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a[4], b[9];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = src[(threadIdx.x + i * 512) & 4095];
#pragma unroll
    for (int i = 0; i < 9; ++i) b[i] = src[(threadIdx.x * 3 + i * 512 + 7) & 4095];
    const long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i & 3], b[i % 9], acc[i], 0, 0, 0);
        asm volatile("" ::: "memory");
    }
    const long long r1 = __builtin_amdgcn_s_memrealtime();
    f32x4 s = acc[0];
#pragma unroll
    for (int i = 1; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * 512 + threadIdx.x] = s.x + s.y + s.z + s.w;
    if (threadIdx.x == 0 && blockIdx.x == 1) tim[2] = r1 - r0;
}

int main() {
    float *src, *out; long long* tim;
    hipMalloc(&src, 4096 * 4); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&tim, 64);
    std::vector<float> h(4096);
    for (int mode = 0; mode < 2; ++mode) {          // 0: zeros (low toggle), 1: random operands
        for (int i = 0; i < 4096; ++i) h[i] = mode ? (float)((i * 2654435761u) >> 8) / 16777216.f - 0.5f : 0.f;
        hipMemcpy(src, h.data(), 4096 * 4, hipMemcpyHostToDevice);
        for (int on = 0; on < 2; ++on) {
            const int iters = 12000, spin = on ? 40000 : 40000;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 3; ++rep) {
                hipMemset(tim, 0, 64);
                hipEventRecord(e0);
                hipLaunchKernelGGL(mfma_loop<36>, dim3(256), dim3(576), 0, 0, src, out, tim, iters, spin, on);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                long long t[3]; hipMemcpy(t, tim, 24, hipMemcpyDeviceToHost);
                const double spin_us = t[0] / 100.0, spin_cyc = (double)spin * 8 * 16;
                const double mfma_us = t[2] / 100.0;
                const double tflops = on ? 256.0 * 8 * iters * 36 * 2048.0 / (mfma_us * 1e-6) / 1e12 : 0.0;
                printf("data=%s mfma=%d rep=%d kernel %.1f us | spin %.1f us (%.0f nominal cycles -> %.3f GHz-equivalent, memtime/realtime %.3f) | mfma loop %.1f us = %.1f TFLOP/s\n",
                       mode ? "random" : "zeros", on, rep, ms * 1e3, spin_us, spin_cyc, spin_cyc / spin_us / 1e3, (double)t[1] / (double)t[0], mfma_us, tflops);
            }
        }
    }
    return 0;
}
