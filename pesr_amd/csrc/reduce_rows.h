// Device-side form of reduce.hip's fixed-order second stage, for kernels that finish a "per-block partials -> final" pattern
// in the SAME launch that consumes the sums (BatchNorm finalize, loss scalars): one 1024-thread block = 64 columns x 16 row
// lanes; every thread adds its rows in increasing order, the 16 lanes are combined through LDS in lane order.  Identical
// arithmetic, order and result to reduce_rows_kernel.  All 1024 threads must call it; the sum is returned to row lane 0.
#pragma once
#include "common.h"

__device__ __forceinline__ double reduce_rows_block(const float* __restrict__ part, int nb, int ncols, int col, bool valid,
                                                    double (*red)[64]) {
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    double s = 0.0;
    if (valid) {
        int k = rl;
        for (; k + 48 < nb; k += 64) {   // 4 independent loads in flight
            const float a = part[(size_t)k * ncols + col], b = part[(size_t)(k + 16) * ncols + col];
            const float c = part[(size_t)(k + 32) * ncols + col], d = part[(size_t)(k + 48) * ncols + col];
            s += (double)a; s += (double)b; s += (double)c; s += (double)d;
        }
        for (; k < nb; k += 16) s += (double)part[(size_t)k * ncols + col];
    }
    __syncthreads();                      // (red may still be read from a previous call)
    red[rl][cl] = s;
    __syncthreads();
    double t = 0.0;
    if (rl == 0) {
        t = red[0][cl];
#pragma unroll
        for (int j = 1; j < 16; ++j) t += red[j][cl];
    }
    return t;
}
