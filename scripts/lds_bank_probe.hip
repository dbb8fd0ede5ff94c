// Measures the LDS cycles of ONE ds_read_b128 wave-instruction for arbitrary per-lane byte addresses (gfx950).
//   hipcc -O3 --offload-arch=gfx950 scripts/lds_bank_probe.hip -o exp/lds_bank_probe ; exp/lds_bank_probe < patterns.txt
// stdin: one pattern per line: a name followed by 64 byte offsets.  Output: LDS cycles per read (4 waves x 8 reads in flight, 2000 rounds: pipe-bound).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void probe(const int* offs, long long* out, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384; i += 256) ((float*)lds)[i] = (float)i;
    __syncthreads();
    const unsigned a = (unsigned)offs[lane];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    long long t0 = 0, t1 = 0;
    for (int rep = 0; rep < 2; ++rep) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        for (int it = 0; it < 2000; ++it) {   // 4 waves x 8 reads in flight: the LDS pipe is the bottleneck
            f32x4 v0, v1, v2, v3, v4, v5, v6, v7;
            asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8\n\tds_read_b128 %2, %8\n\tds_read_b128 %3, %8\n\t"
                         "ds_read_b128 %4, %8\n\tds_read_b128 %5, %8\n\tds_read_b128 %6, %8\n\tds_read_b128 %7, %8\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7)
                         : "v"(a)
                         : "memory");
            acc += v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
        }
        __syncthreads();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    }
    if (threadIdx.x == 0) out[0] = t1 - t0;
    sink[threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}
int main() {
    int* d_offs; long long* d_out; float* d_sink;
    hipMalloc(&d_offs, 64 * sizeof(int)); hipMalloc(&d_out, 8); hipMalloc(&d_sink, 256 * sizeof(float));
    std::string line;
    while (std::getline(std::cin, line)) {
        std::istringstream is(line);
        std::string name; is >> name;
        std::vector<int> offs(64);
        bool ok = true;
        for (int i = 0; i < 64; ++i) if (!(is >> offs[i])) ok = false;
        if (!ok) continue;
        hipMemcpy(d_offs, offs.data(), 64 * sizeof(int), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(256), 65536, 0, d_offs, d_out, d_sink);
        long long dt = 0;
        hipMemcpy(&dt, d_out, 8, hipMemcpyDeviceToHost);
        printf("%-40s %.2f cycles/read\n", name.c_str(), (double)dt / (2000.0 * 8.0 * 4.0));
    }
    return 0;
}
