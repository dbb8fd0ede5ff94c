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
