// Sustained fp32 MFMA rate of the chip under this repo's issue pattern: NACC independent 16x16x4 accumulators per wave, operands
// in registers (random data), no memory traffic in the loop.  Prints TFLOP/s and the in-kernel clock (s_memtime / s_memrealtime).
//   hipcc -O3 --offload-arch=gfx950 scripts/mfma_peak_probe.hip -o exp/mfma_peak_probe && exp/mfma_peak_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(512) void probe(const float* in, float* out, int iters, unsigned long long* clk) {
    f32x4 acc[NACC];
    float a[6], b[9];
    for (int i = 0; i < 6; ++i) a[i] = in[threadIdx.x * 16 + i];
    for (int i = 0; i < 9; ++i) b[i] = in[threadIdx.x * 16 + 6 + i];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i % 6], b[i % 9], acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    f32x4 s = acc[0];
    for (int i = 1; i < NACC; ++i) s += acc[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y + s.z + s.w;
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int NACC>
static void run(const char* name, int threads, const float* in, float* out, unsigned long long* clk, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256;
    for (int w = 0; w < 200; ++w) hipLaunchKernelGGL(probe<NACC>, dim3(grid), dim3(threads), 0, 0, in, out, iters, clk);   // warm the clocks
    hipDeviceSynchronize();
    std::vector<float> ms;
    for (int rep = 0; rep < 9; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(probe<NACC>, dim3(grid), dim3(threads), 0, 0, in, out, iters, clk);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float t; hipEventElapsedTime(&t, e0, e1); ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    std::vector<unsigned long long> h(grid * 2);
    hipMemcpy(h.data(), clk, grid * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::vector<double> ghz;
    for (int i = 0; i < grid; ++i) ghz.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 0.1);
    std::sort(ghz.begin(), ghz.end());
    const double flops = (double)grid * (threads / 64) * iters * NACC * 2048.0;
    printf("%-34s %8.1f us  %6.1f TFLOP/s   in-kernel clock median %.3f GHz (min %.3f max %.3f)\n", name, ms[4] * 1e3, flops / (ms[4] * 1e-3) / 1e12,
           ghz[grid / 2], ghz[0], ghz[grid - 1]);
}

int main() {
    float *in, *out; unsigned long long* clk;
    hipMalloc(&in, 512 * 16 * sizeof(float)); hipMalloc(&out, 256 * 512 * sizeof(float)); hipMalloc(&clk, 256 * 2 * sizeof(unsigned long long));
    std::vector<float> h(512 * 16);
    srand(1);
    for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(in, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
    run<18>("18 acc, 8 waves/CU, 200 us", 512, in, out, clk, 576 * 1);
    run<18>("18 acc, 8 waves/CU, 2 ms", 512, in, out, clk, 576 * 10);
    run<27>("27 acc, 8 waves/CU, 200 us", 512, in, out, clk, 384 * 1);
    run<27>("27 acc, 8 waves/CU, 2 ms", 512, in, out, clk, 384 * 10);
    run<18>("18 acc, 4 waves/CU, 2 ms", 256, in, out, clk, 1152 * 10);
    // zero operands: the same loop without data-dependent power
    hipMemset(in, 0, 512 * 16 * sizeof(float));
    run<18>("18 acc, 8 waves/CU, 2 ms, zeros", 512, in, out, clk, 576 * 10);
    return 0;
}
