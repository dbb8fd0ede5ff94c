// HBM rate of two ways to fetch a [pixels][256 ch] fp32 tensor into the lanes of a 16-pixel MFMA tile (the C -> 3 conv's A operand):
//   A  lane (pixel i = lane & 15, k-slot g = lane >> 4) loads 16 B at channel 16 * chunk + 4 g: one instruction = 16 pixels x 64 B
//   B  lane (pixel p = lane & 7,  c8 = lane >> 3)       loads 16 B at channel 32 * chunk + 4 c8: one instruction = 8 pixels x 128 B (whole lines)
// Same bytes, same number of instructions, 16 KiB in flight per wave (all loads of a tile issued before the first is used).
//   hipcc -O3 --offload-arch=gfx950 scripts/load_pattern_probe.hip -o /tmp/lpp && /tmp/lpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int PATTERN>
__global__ __launch_bounds__(256) void probe(const float* x, unsigned* out, long tiles) {
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    u32x4 acc = {0u, 0u, 0u, 0u};
    for (long t = wave; t < tiles; t += nwaves) {
        const float* base = x + t * 16 * 256;                  // 16 pixels x 256 channels = 16 KiB
        u32x4 v[16];
        if (PATTERN == 0) {
            const int i = lane & 15, g = lane >> 4;
#pragma unroll
            for (int ch = 0; ch < 16; ++ch) v[ch] = *(const u32x4*)(base + i * 256 + ch * 16 + 4 * g);
        } else {
            const int p = lane & 7, c8 = lane >> 3;
#pragma unroll
            for (int ch = 0; ch < 8; ++ch) {
                v[2 * ch] = *(const u32x4*)(base + p * 256 + ch * 32 + 4 * c8);
                v[2 * ch + 1] = *(const u32x4*)(base + (p + 8) * 256 + ch * 32 + 4 * c8);
            }
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) acc ^= v[k];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[threadIdx.x] = 1;
}

template <int PATTERN>
static void run(const char* name, const float* x, unsigned* out, long tiles, int wgs) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(probe<PATTERN>, dim3(wgs), dim3(256), 0, 0, x, out, tiles);
    hipDeviceSynchronize();
    std::vector<float> ms;
    for (int rep = 0; rep < 9; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(probe<PATTERN>, dim3(wgs), dim3(256), 0, 0, x, out, tiles);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float t; hipEventElapsedTime(&t, e0, e1); ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    const double bytes = (double)tiles * 16384;
    printf("%-58s %4d workgroups  %8.1f us  %6.2f TB/s\n", name, wgs, ms[4] * 1e3, bytes / (ms[4] * 1e-3) / 1e12);
}

int main() {
    const long tiles = 16L * 192 * 192 / 16 * 4;       // 4 x (16 x 192 x 192 pixels): 2.4 GB, far beyond the 256 MiB Infinity Cache
    float* x; unsigned* out;
    hipMalloc(&x, tiles * 16384); hipMalloc(&out, 1024);
    hipMemset(x, 1, tiles * 16384);
    for (int wgs : {512, 1024, 2048}) {
        run<0>("A: 16 pixels x 64 B per instruction (the product's pattern)", x, out, tiles, wgs);
        run<1>("B: 8 pixels x 128 B per instruction (whole lines)", x, out, tiles, wgs);
    }
    return 0;
}
