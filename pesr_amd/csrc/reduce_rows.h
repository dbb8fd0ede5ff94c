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
        for (; k + 240 < nb; k += 256) { // 16 independent loads in flight (thousands of rows: the C <-> 3 weight gradients' partials)
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = part[(size_t)(k + 16 * u) * ncols + col];
#pragma unroll
            for (int u = 0; u < 16; ++u) s += (double)v[u];
        }
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

// Two columns at once (BatchNorm's sum / sum-of-squares pairs): the same per-thread order of additions for each column as two
// calls of reduce_rows_block, but the loads of both columns are in flight together and the lanes meet once - these launches are
// latency chains of a few microseconds (nb <= 512 rows), two of them back to back were 7 - 9 us.
__device__ __forceinline__ void reduce_rows_block2(const float* __restrict__ part, int nb, int ncols, int col0, int col1, bool valid,
                                                   double (*red0)[64], double (*red1)[64], double* out0, double* out1) {
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    double s = 0.0, u = 0.0;
    if (valid) {
        int k = rl;
        for (; k + 48 < nb; k += 64) {   // 8 independent loads in flight
            const float* p0 = part + (size_t)k * ncols;
            const float a = p0[col0], b = p0[(size_t)16 * ncols + col0], c = p0[(size_t)32 * ncols + col0], d = p0[(size_t)48 * ncols + col0];
            const float e = p0[col1], f = p0[(size_t)16 * ncols + col1], g = p0[(size_t)32 * ncols + col1], h = p0[(size_t)48 * ncols + col1];
            s += (double)a; s += (double)b; s += (double)c; s += (double)d;
            u += (double)e; u += (double)f; u += (double)g; u += (double)h;
        }
        for (; k < nb; k += 16) { s += (double)part[(size_t)k * ncols + col0]; u += (double)part[(size_t)k * ncols + col1]; }
    }
    __syncthreads();
    red0[rl][cl] = s; red1[rl][cl] = u;
    __syncthreads();
    double t0 = 0.0, t1 = 0.0;
    if (rl == 0) {
        t0 = red0[0][cl]; t1 = red1[0][cl];
#pragma unroll
        for (int j = 1; j < 16; ++j) { t0 += red0[j][cl]; t1 += red1[j][cl]; }
    }
    *out0 = t0; *out1 = t1;
}
