// Does work on the other pipes cost fp32-MFMA time?  768-thread workgroups (3 waves per SIMD, one workgroup per CU), four independent
// 32x32x2 fp32 accumulators per wave (the y-nested weight gradient's k-step), operands in registers; per k-step (4 MFMAs) each wave also
// issues NV independent VALU instructions (v_fma_f32 on private registers, results kept alive), NL ds_read_b32 and NS s_nop.
// Prints us per launch and MFMA-pipe utilisation against the pure-MFMA build.
//   hipcc -O3 --offload-arch=gfx950 scripts/mfma_valu_probe.hip -o exp/mfma_valu_probe && exp/mfma_valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, int NL, int PK>
__global__ __launch_bounds__(768) void probe(const float* in, float* out, int iters) {
    __shared__ float lds[768 * 8];
    f32x16 acc[4];
    float a[4], b[4], v[8];
    for (int i = 0; i < 4; ++i) { a[i] = in[threadIdx.x * 16 + i]; b[i] = in[threadIdx.x * 16 + 4 + i]; }
    for (int i = 0; i < 8; ++i) { v[i] = in[threadIdx.x * 16 + 8 + i]; lds[threadIdx.x * 8 + i] = v[i]; }
    for (int t = 0; t < 4; ++t) for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
    __syncthreads();
    const float* lp = lds + threadIdx.x;
    float l[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NL; ++i) l[i & 7] += lp[(i & 7) * 768];            // ds_read_b32 + 1 VALU each (the add is counted in NV below)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[t], acc[t], 0, 0, 0);
            if (PK == 0) {
#pragma unroll
                for (int i = 0; i < NV; ++i) if (i % 4 == t) v[i & 7] = __builtin_fmaf(v[i & 7], 1.0000001f, 1e-9f);
            } else {
#pragma unroll
                for (int i = 0; i < NV; ++i) if (i % 4 == t) {
                    typedef float f2 __attribute__((ext_vector_type(2)));
                    f2 p = {v[(2 * i) & 7], v[(2 * i + 1) & 7]};
                    p = p * (f2){1.0000001f, 1.0000001f} + (f2){1e-9f, 1e-9f};
                    v[(2 * i) & 7] = p.x; v[(2 * i + 1) & 7] = p.y;
                }
            }
        }
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int j = 0; j < 16; ++j) s += acc[t][j];
    for (int i = 0; i < 8; ++i) s += v[i] + l[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV, int NL, int PK>
static double run(const char* name, const float* in, float* out, int iters, double base_us) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 100; ++w) hipLaunchKernelGGL((probe<NV, NL, PK>), dim3(256), dim3(768), 0, 0, in, out, iters);
    hipDeviceSynchronize();
    std::vector<float> ms;
    for (int rep = 0; rep < 9; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((probe<NV, NL, PK>), dim3(256), dim3(768), 0, 0, in, out, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float t; hipEventElapsedTime(&t, e0, e1); ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    const double us = ms[4] * 1e3;
    const double ideal = (double)iters * 12 * 64 / 2.39e3;        // 3 waves x 4 MFMAs x 64 cycles per k-step and SIMD, at 2.39 GHz
    printf("%-46s %8.1f us   pipe busy %.3f   %+7.1f cycles per k-step and SIMD over the MFMA-only build\n", name, us, ideal / us,
           base_us > 0 ? (us - base_us) * 2.39e3 / iters : 0.0);
    return us;
}

int main() {
    float *in, *out;
    hipMalloc(&in, 768 * 16 * sizeof(float)); hipMalloc(&out, 256 * 768 * sizeof(float));
    std::vector<float> h(768 * 16);
    for (auto& x : h) x = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(in, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
    const int iters = 288;                                        // one G-body launch: 48 segments x 6 k-steps
    const double b = run<0, 0, 0>("MFMA only (4 per k-step and wave)", in, out, iters, 0);
    run<4, 0, 0>("+ 4 v_fma_f32 per k-step and wave", in, out, iters, b);
    run<8, 0, 0>("+ 8 v_fma_f32", in, out, iters, b);
    run<16, 0, 0>("+ 16 v_fma_f32", in, out, iters, b);
    run<32, 0, 0>("+ 32 v_fma_f32", in, out, iters, b);
    run<8, 0, 1>("+ 8 v_pk_fma_f32", in, out, iters, b);
    run<16, 0, 1>("+ 16 v_pk_fma_f32", in, out, iters, b);
    run<6, 6, 0>("+ 6 ds_read_b32 + 6 v_add", in, out, iters, b);
    run<12, 12, 0>("+ 12 ds_read_b32 + 12 v_add", in, out, iters, b);
    return 0;
}
