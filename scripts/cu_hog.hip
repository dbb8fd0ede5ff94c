// Rehearsal of RCCL-vs-compute CU contention on ONE GPU (scripts/cu_contention.py): `k` workgroups that each sit on a CU for
// `us` microseconds.  lds_bytes > 80 KiB makes two of them (and any of this repo's ~150 KiB conv workgroups) unable to share
// a CU - the case of a communication kernel whose workgroup cannot co-reside; a small lds_bytes lets a conv workgroup move
// in beside it if registers allow.  Not product code.
#include <hip/hip_runtime.h>
extern "C" __global__ void __launch_bounds__(256) cu_hog_kernel(long long ticks, unsigned* where) {
    extern __shared__ char lds[];
    const long long t0 = wall_clock64();
    if (threadIdx.x == 0) {
        lds[0] = 1;
        unsigned xcc, cu;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(cu));
        where[blockIdx.x] = (xcc & 15) << 16 | (cu & 0xffff);
    }
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
extern "C" int cu_hog(int k, int lds_bytes, int us, unsigned* where, hipStream_t stream) {
    static bool set = false;
    if (!set) { hipFuncSetAttribute((const void*)cu_hog_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set = true; }
    hipLaunchKernelGGL(cu_hog_kernel, dim3(k), dim3(256), lds_bytes, stream, (long long)us * 100, where);   // wall_clock64: 100 MHz
    return (int)hipGetLastError();
}
