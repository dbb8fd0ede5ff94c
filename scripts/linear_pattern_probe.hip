// HBM rate of ways to stream the Discriminator's Linear(73728, 1024) weight matrix W[1024][K] (302 MB) into MFMA operand lanes
// (lane (i = lane & 15, g = lane >> 4) holds 16 B of row n0 + 16 b + i at k + 4 g; loads only, XOR-folded), to see what holds
// linear_fwd_mfma_kernel at ~3 TB/s when a contiguous stream reaches ~6:
//   T0  the product's tiling: a wave owns 64 rows (NB = 4) x 576 k, two 16-k steps in flight (10 loads)
//   T1  T0 on a matrix whose row pitch is K + 64 floats (is it the 2^15 x 9 byte pitch - every row of a tile on the same channels?)
//   T2  a wave owns 16 rows (NB = 1) x 2304 k, eight 16-k steps in flight
//   T3  a wave owns 64 rows x 576 k, eight 16-k steps of ONE 16-row group in flight at a time (32 loads)
//   T4  the same bytes as one contiguous stream (16 KiB per wave and trip): the ceiling for this buffer size
//   hipcc -O3 --offload-arch=gfx950 scripts/linear_pattern_probe.hip -o /tmp/lin && /tmp/lin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr long K = 73728; constexpr int N = 1024;

template <int NB, int LU>
__global__ __launch_bounds__(256) void tiled(const float* W, unsigned* out, long pitch, long kchunk, int ksplit) {
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
    const int ngroups = N / (16 * NB), ng = wave % ngroups, ks = wave / ngroups;
    if (ks >= ksplit) return;
    const long k0 = ks * kchunk, k1 = k0 + kchunk;
    u32x4 acc = {0u, 0u, 0u, 0u};
    const float* wr[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) wr[b] = W + (size_t)(ng * 16 * NB + b * 16 + i) * pitch;
    for (long k = k0 + 4 * g; k < k1; k += 16 * LU) {
        u32x4 w[LU][NB];
#pragma unroll
        for (int u = 0; u < LU; ++u)
#pragma unroll
            for (int b = 0; b < NB; ++b) w[u][b] = __builtin_nontemporal_load((const u32x4*)(wr[b] + k + 16 * u));
#pragma unroll
        for (int u = 0; u < LU; ++u)
#pragma unroll
            for (int b = 0; b < NB; ++b) acc ^= w[u][b];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[threadIdx.x] = 1;
}
// T5: T0 plus what the real kernel does beside the W stream: XL = the x piece per 16-k step (16 rows of x, from L2), MF = the MFMAs
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int XL, int MF>
__global__ __launch_bounds__(256) void tiled_full(const float* W, const float* x, float* part, long kchunk, int ksplit) {
    constexpr int NB = 4, LU = 2;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
    const int ngroups = N / 64, ng = wave % ngroups, ks = wave / ngroups;
    if (ks >= ksplit) return;
    const long k0 = ks * kchunk, k1 = k0 + kchunk;
    f32x4 acc[NB];
    for (int b = 0; b < NB; ++b) acc[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* wr[NB];
    for (int b = 0; b < NB; ++b) wr[b] = W + (size_t)(ng * 64 + b * 16 + i) * K;
    const float* xr = x + (size_t)i * K;
    for (long k = k0 + 4 * g; k < k1; k += 16 * LU) {
        f32x4 a[LU], w[LU][NB];
#pragma unroll
        for (int u = 0; u < LU; ++u) {
            a[u] = XL ? *(const f32x4*)(xr + k + 16 * u) : (f32x4){1.f, 2.f, 3.f, 4.f};
#pragma unroll
            for (int b = 0; b < NB; ++b) w[u][b] = __builtin_nontemporal_load((const f32x4*)(wr[b] + k + 16 * u));
        }
#pragma unroll
        for (int u = 0; u < LU; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    if (MF) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][e], w[u][b][e], acc[b], 0, 0, 0);
                    else acc[b][e] += a[u][e] * w[u][b][e];
                }
    }
    for (int b = 0; b < NB; ++b)
        for (int jj = 0; jj < 4; ++jj) part[((size_t)ks * 16 + 4 * g + jj) * N + ng * 64 + b * 16 + i] = acc[b][jj];
}
// T3: 64 rows per wave, but one 16-row group at a time, LU steps deep
template <int LU>
__global__ __launch_bounds__(256) void tiled_rows(const float* W, unsigned* out, long pitch, long kchunk, int ksplit) {
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
    const int ngroups = N / 64, ng = wave % ngroups, ks = wave / ngroups;
    if (ks >= ksplit) return;
    const long k0 = ks * kchunk, k1 = k0 + kchunk;
    u32x4 acc = {0u, 0u, 0u, 0u};
    for (int b = 0; b < 4; ++b) {
        const float* wr = W + (size_t)(ng * 64 + b * 16 + i) * pitch;
        for (long k = k0 + 4 * g; k < k1; k += 16 * LU) {
            u32x4 w[LU];
#pragma unroll
            for (int u = 0; u < LU; ++u) w[u] = (k + 16 * u < k1 + 4 * g) ? __builtin_nontemporal_load((const u32x4*)(wr + k + 16 * u)) : (u32x4){0u, 0u, 0u, 0u};
#pragma unroll
            for (int u = 0; u < LU; ++u) acc ^= w[u];
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[threadIdx.x] = 1;
}
__global__ __launch_bounds__(256) void stream(const float* W, unsigned* out, long tiles) {
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    u32x4 acc = {0u, 0u, 0u, 0u};
    for (long t = wave; t < tiles; t += nwaves) {
        const float* base = W + t * 4096;
        u32x4 v[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) v[c] = __builtin_nontemporal_load((const u32x4*)(base + c * 256 + lane * 4));
#pragma unroll
        for (int c = 0; c < 16; ++c) acc ^= v[c];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[threadIdx.x] = 1;
}

template <typename F>
static void timeit(const char* name, F launch, char* flush, size_t flush_bytes) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) launch();
    hipDeviceSynchronize();
    std::vector<float> ms;
    for (int rep = 0; rep < 9; ++rep) {
        hipMemsetAsync(flush, rep, flush_bytes, 0);          // 512 MB through the caches: W comes from HBM
        hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float t; hipEventElapsedTime(&t, e0, e1); ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    printf("%-78s %8.1f us  %5.2f TB/s\n", name, ms[4] * 1e3, (double)N * K * 4 / (ms[4] * 1e-3) / 1e12);
}

int main() {
    float* W; float* Wp; unsigned* out; char* flush;
    const size_t flush_bytes = 512u << 20;
    hipMalloc(&W, (size_t)N * K * 4); hipMalloc(&Wp, (size_t)N * (K + 64) * 4); hipMalloc(&out, 1024); hipMalloc(&flush, flush_bytes);
    hipMemset(W, 1, (size_t)N * K * 4); hipMemset(Wp, 1, (size_t)N * (K + 64) * 4);
    timeit("T0 product tiling: 64 rows x 576 k per wave, 10 loads in flight", [&] { hipLaunchKernelGGL((tiled<4, 2>), dim3(512), dim3(256), 0, 0, W, out, K, 576L, 128); }, flush, flush_bytes);
    timeit("T1 the same on a row pitch of K + 64 floats", [&] { hipLaunchKernelGGL((tiled<4, 2>), dim3(512), dim3(256), 0, 0, Wp, out, K + 64, 576L, 128); }, flush, flush_bytes);
    timeit("T0b 64 rows x 576 k, 20 loads in flight (LU 4: 9 trips need kchunk % 64 == 0 -> 576 ok)", [&] { hipLaunchKernelGGL((tiled<4, 3>), dim3(512), dim3(256), 0, 0, W, out, K, 576L, 128); }, flush, flush_bytes);
    timeit("T2 16 rows x 2304 k per wave, 8 loads in flight", [&] { hipLaunchKernelGGL((tiled<1, 8>), dim3(512), dim3(256), 0, 0, W, out, K, 2304L, 32); }, flush, flush_bytes);
    timeit("T2b 16 rows x 1152 k per wave (4096 waves), 8 loads in flight", [&] { hipLaunchKernelGGL((tiled<1, 8>), dim3(1024), dim3(256), 0, 0, W, out, K, 1152L, 64); }, flush, flush_bytes);
    timeit("T3 64 rows x 576 k per wave, one 16-row group at a time, 9 loads in flight", [&] { hipLaunchKernelGGL((tiled_rows<9>), dim3(512), dim3(256), 0, 0, W, out, K, 576L, 128); }, flush, flush_bytes);
    timeit("T3p the same on the padded pitch", [&] { hipLaunchKernelGGL((tiled_rows<9>), dim3(512), dim3(256), 0, 0, Wp, out, K + 64, 576L, 128); }, flush, flush_bytes);
    float* x; float* part;
    hipMalloc(&x, (size_t)16 * K * 4); hipMalloc(&part, (size_t)128 * 16 * N * 4); hipMemset(x, 0, (size_t)16 * K * 4);
    timeit("T5a T0 + one VALU FMA per element (no x loads, no MFMA), partial stores", [&] { hipLaunchKernelGGL((tiled_full<0, 0>), dim3(512), dim3(256), 0, 0, W, x, part, 576L, 128); }, flush, flush_bytes);
    timeit("T5b T0 + x loads, VALU FMA", [&] { hipLaunchKernelGGL((tiled_full<1, 0>), dim3(512), dim3(256), 0, 0, W, x, part, 576L, 128); }, flush, flush_bytes);
    timeit("T5c T0 + MFMAs (x = constants)", [&] { hipLaunchKernelGGL((tiled_full<0, 1>), dim3(512), dim3(256), 0, 0, W, x, part, 576L, 128); }, flush, flush_bytes);
    timeit("T5d T0 + x loads + MFMAs (= the product kernel's loop)", [&] { hipLaunchKernelGGL((tiled_full<1, 1>), dim3(512), dim3(256), 0, 0, W, x, part, 576L, 128); }, flush, flush_bytes);
    timeit("T4 contiguous stream, 16 KiB per wave and trip, 2048 waves", [&] { hipLaunchKernelGGL(stream, dim3(512), dim3(256), 0, 0, W, out, (long)N * K / 4096); }, flush, flush_bytes);
    timeit("T4b contiguous stream, 8192 waves", [&] { hipLaunchKernelGGL(stream, dim3(2048), dim3(256), 0, 0, W, out, (long)N * K / 4096); }, flush, flush_bytes);
    return 0;
}
