// Fixed-order second-stage reduction shared by every "per-block partials -> final" pattern in this library
// (bias gradients, BatchNorm statistics, loss sums, RGB weight gradients):
//   dsum[col] = sum_{k < nb} part[k * ncols + col]      accumulated in double
// One 1024-thread block owns 64 columns x 16 row lanes; each thread adds its rows in increasing order, the
// 16 lanes are combined through LDS in lane order - the result does not depend on scheduling (bitwise
// reproducible), and no atomics are used.
#include "common.h"
#include "launchers.h"
#include "reduce_rows.h"

__global__ __launch_bounds__(1024) void reduce_rows_kernel(const float* __restrict__ part, double* __restrict__ dsum, int nb,
                                                           int ncols) {
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + cl;
    __shared__ double red[16][64];
    const double t = reduce_rows_block(part, nb, ncols, col, col < ncols, red);
    if (rl == 0 && col < ncols) dsum[col] = t;
}

int pesr_reduce_rows_launch(const float* part, double* dsum, int nb, int ncols, hipStream_t stream) {
    hipLaunchKernelGGL(reduce_rows_kernel, dim3((ncols + 63) / 64), dim3(1024), 0, stream, part, dsum, nb, ncols);
    return pesr_launch_status();
}
