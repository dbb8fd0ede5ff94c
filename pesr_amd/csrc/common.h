// Shared device/host helpers for the pesr_amd HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

// Error codes returned across the C ABI (0 = ok, >0 = hipError_t, <0 = ours).
#define PESR_OK 0
#define PESR_EINVAL (-1)     // unsupported shape / bad argument
#define PESR_EWORKSPACE (-2) // workspace too small

#define PESR_API extern "C" __attribute__((visibility("default")))

static inline int pesr_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PESR_OK : (int)e;
}

// One-time kernel attributes (hipFuncAttributeMaxDynamicSharedMemorySize) belong to the function object of ONE device: a process
// that drives several GPUs (the reference's single-process nn.DataParallel, train.py:114-118) must set them once per device, not
// once per process.  static PesrDeviceOnce once; once([&] { hipFuncSetAttribute(...); });
#ifdef __cplusplus
#include <atomic>
#include <mutex>
struct PesrDeviceOnce {
    std::atomic<unsigned long long> done{0};
    std::mutex m;
    template <class F> void operator()(F&& f) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        const unsigned long long bit = 1ull << (dev & 63);
        if (done.load(std::memory_order_acquire) & bit) return;
        std::lock_guard<std::mutex> g(m);
        if (done.load(std::memory_order_relaxed) & bit) return;
        f();
        done.fetch_or(bit, std::memory_order_release);
    }
};
#endif

static inline int pesr_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// epilogue activation ids
#define PESR_ACT_NONE 0
#define PESR_ACT_RELU 1
#define PESR_ACT_LRELU 2

// ---- wave-level reductions (wave = 64 lanes on gfx950) --------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Column sum over `rows` partial rows in increasing order, in double - eight loads in flight at a time (as a plain loop hipcc emits one
// load, one s_waitcnt vmcnt(0) and one add per iteration: `rows` dependent round trips on a single thread that its whole block waits for).
__device__ __forceinline__ double pesr_colsum_rows(const float* __restrict__ part, int rows, size_t stride) {
    double s = 0.0;
    int k = 0;
    for (; k + 8 <= rows; k += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(k + u) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += (double)v[u];
    }
    for (; k < rows; ++k) s += (double)part[(size_t)k * stride];
    return s;
}


// ---- BatchNorm sums out of a conv kernel's epilogue (round 6; SURVEY K10, reference model/basic.py:26-30) ------------------------
// mode 1 (a conv in front of a BatchNorm): the workgroup leaves the per-channel sum and sum of squares of what it stored in
//   part[row0 + pixel tile][2][C] - bn_reduce_kernel<0>'s row layout, so bn_finalize_kernel takes the rows as they are.
// mode 2 (the input gradient that PRODUCES the gradient of a BatchNorm + LeakyReLU output y = lrelu(gamma * xhat(z) + beta)):
//   the stored value is g' = v * lrelu'(gamma * xhat + beta) and the rows hold sum g' and sum g' * xhat (bn_reduce_kernel<1>'s
//   sums): the BatchNorm backward then only needs its apply pass.
// No atomics: one row per pixel tile, the n-tiles' channels side by side; a thread adds its few dozen pixels in fp32, everything
// across threads in double, in a fixed order.
// The struct is the LAST by-value parameter of the conv kernels and is read through a laundered kernarg pointer AFTER the main
// loop (pesr_bn_epi): left to itself hipcc loads every kernel argument in the prologue and keeps it in SGPRs through the loop,
// and the 9 extra registers pushed three direct-conv configurations over the SGPR budget (spills to scratch).
struct BnEpi {
    int mode;
    float slope;
    float* part;
    const float* z;           // mode 2: the BatchNorm's input, same shape as the kernel's output
    const float* mi;          // mode 2: mean_invstd [2][C]
    const float* gamma;
    const float* beta;
};
__device__ __forceinline__ const BnEpi* pesr_bn_epi(unsigned kernarg_offset) {
    const char* ka = (const char*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));      // opaque from here on: nothing read through it can be hoisted above this point
    return (const BnEpi*)(ka + kernarg_offset);
}
