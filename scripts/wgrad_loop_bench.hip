// Measurement tool (not product code): the inner loop of conv3x3_wgrad_kernel<4,1,48,1> in isolation.
//   hipcc -O3 --offload-arch=gfx950 scripts/wgrad_loop_bench.hip -o exp/wgrad_loop_bench && exp/wgrad_loop_bench
// MODE 0: MFMAs only (operands fixed)       MODE 1: + 13 ds_read_b32 fragments per 36 MFMAs, compiler-scheduled
// MODE 2: as 1 with sched_barrier-pinned double buffering      MODE 3: A fragments as one ds_read_b128 (co = 4r + i)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int TWO = 48, TWX = 50, XS = 80, DS = 144, X_FLOATS = 3 * TWX * XS, BUF = X_FLOATS + TWO * DS;

template <int MODE>
__global__ __launch_bounds__(512) void loop_kernel(const float* __restrict__ src, float* __restrict__ out, int segs) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
    for (int i = tid; i < 2 * BUF; i += 512) lds[i] = src[i & 4095];
    __syncthreads();
    f32x4 acc[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[t][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int ci_tile = wave & 3, co_half = wave >> 2;
#pragma unroll 1
    for (int seg = 0; seg < segs; ++seg) {
        const float* buf = lds + (seg & 1) * BUF;
        const float* xb = buf + ci_tile * 16 + r;
        const float* db = buf + X_FLOATS + co_half * 64 + (MODE == 3 ? r * 4 : r);
        float av0[4], bv0[9], av1[4], bv1[9];
#define SB if (MODE >= 2) __builtin_amdgcn_sched_barrier(0);
#define RD(AV, BV, K4)                                                                              \
        {                                                                                           \
            const int px_ = (K4) * 4 + g;                                                           \
            if (MODE == 3) { const f32x4 a4 = *(const f32x4*)(db + px_ * DS); AV[0] = a4.x; AV[1] = a4.y; AV[2] = a4.z; AV[3] = a4.w; } \
            else { _Pragma("unroll") for (int i = 0; i < 4; ++i) AV[i] = db[px_ * DS + i * 16]; }   \
            _Pragma("unroll") for (int t = 0; t < 9; ++t) BV[t] = xb[((t / 3) * TWX + px_ + (t % 3)) * XS]; \
            SB                                                                                      \
        }
#define MM(AV, BV)                                                                                  \
        _Pragma("unroll") for (int t = 0; t < 9; ++t)                                               \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                           \
                acc[t][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV[i], BV[t], acc[t][i], 0, 0, 0); \
        SB
        RD(av0, bv0, 0)
        if (MODE == 0) RD(av1, bv1, 1)
#pragma unroll
        for (int k4 = 0; k4 < TWO / 4; k4 += 2) {
            if (MODE != 0) RD(av1, bv1, k4 + 1)
            MM(av0, bv0)
            if (MODE != 0 && k4 + 2 < TWO / 4) RD(av0, bv0, k4 + 2)
            MM(av1, bv1)
        }
        if (MODE == 0) asm volatile("" ::: "memory");
    }
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) s += acc[t][i];
    out[blockIdx.x * 512 + tid] = s.x + s.y + s.z + s.w;
}

template <int MODE>
static void run(const float* src, float* out) {
    const int segs = 24;
    auto k = loop_kernel<MODE>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 8; ++rep) {
        hipEventRecord(e0);
        for (int j = 0; j < 10; ++j) hipLaunchKernelGGL(k, dim3(256), dim3(512), 2 * BUF * 4, 0, src, out, segs);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double us = best * 100.0, flops = 256.0 * 8 * segs * 432 * 2048.0;
    printf("MODE %d: %.1f us per launch = %.1f TFLOP/s (%.1f %% of 157.3)\n", MODE, us, flops / us / 1e6, flops / us / 1e6 / 1.573);
}
int main() {
    float *src, *out; hipMalloc(&src, 4096 * 4); hipMalloc(&out, 256 * 512 * 4);
    float h[4096]; for (int i = 0; i < 4096; ++i) h[i] = (float)((i * 2654435761u) >> 8) / 16777216.f - 0.5f;
    hipMemcpy(src, h, sizeof h, hipMemcpyHostToDevice);
    run<0>(src, out); run<1>(src, out); run<2>(src, out); run<3>(src, out);
    return 0;
}
