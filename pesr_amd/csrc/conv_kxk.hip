// Generic k x k conv (odd k other than 3, padding k/2, stride s), NHWC fp32 - the COMPATIBILITY path of the reference's
// `Conv(in, out, kernel_size, stride, bias)` (model/basic.py:4-7), which accepts any kernel size although the reference's
// own networks only ever use 3 (the MFMA / Winograd kernels).  Plain VALU kernels, one output element per thread, fixed
// summation order (deterministic); written for correctness and completeness, not tuned: nothing on the train step's path
// reaches them.
//   fwd  : y[n][oy][ox][co]  = b[co] + sum_{ky,kx,ci} x[n][oy*s+ky-p][ox*s+kx-p][ci] * w[co][ci][ky][kx]
//   dgrad: dx[n][iy][ix][ci] = sum_{ky,kx,co} dy[n][oy][ox][co] * w[co][ci][ky][kx]    (oy*s + ky - p == iy, ox likewise)
//   wgrad: dw[co][ci][ky][kx] = sum_{n,oy,ox} dy[n][oy][ox][co] * x[n][oy*s+ky-p][ox*s+kx-p][ci];  db[co] = sum dy
#include "common.h"
#include "launchers.h"

namespace {
__global__ __launch_bounds__(256) void convk_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                        float* __restrict__ y, int N, int H, int W, int Cin, int Cout, int k, int s,
                                                        int OH, int OW) {
    const long total = (long)N * OH * OW * Cout;
    const int p = k / 2;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int co = (int)(e % Cout);
        long r = e / Cout;
        const int ox = (int)(r % OW); r /= OW;
        const int oy = (int)(r % OH);
        const int n = (int)(r / OH);
        float acc = b ? b[co] : 0.f;
        for (int ky = 0; ky < k; ++ky) {
            const int iy = oy * s + ky - p;
            if (iy < 0 || iy >= H) continue;
            for (int kx = 0; kx < k; ++kx) {
                const int ix = ox * s + kx - p;
                if (ix < 0 || ix >= W) continue;
                const float* xp = x + (((size_t)n * H + iy) * W + ix) * Cin;
                const float* wp = w + ((size_t)co * Cin * k + ky) * k + kx;           // + ci * k * k
                for (int ci = 0; ci < Cin; ++ci) acc = fmaf(xp[ci], wp[(size_t)ci * k * k], acc);
            }
        }
        y[e] = acc;
    }
}

__global__ __launch_bounds__(256) void convk_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx,
                                                          int N, int H, int W, int Cin, int Cout, int k, int s, int OH, int OW) {
    const long total = (long)N * H * W * Cin;
    const int p = k / 2;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int ci = (int)(e % Cin);
        long r = e / Cin;
        const int ix = (int)(r % W); r /= W;
        const int iy = (int)(r % H);
        const int n = (int)(r / H);
        float acc = 0.f;
        for (int ky = 0; ky < k; ++ky) {
            const int ty = iy + p - ky;
            if (ty < 0 || ty % s) continue;
            const int oy = ty / s;
            if (oy >= OH) continue;
            for (int kx = 0; kx < k; ++kx) {
                const int tx = ix + p - kx;
                if (tx < 0 || tx % s) continue;
                const int ox = tx / s;
                if (ox >= OW) continue;
                const float* gp = dy + (((size_t)n * OH + oy) * OW + ox) * Cout;
                const float* wp = w + ((size_t)ci * k + ky) * k + kx;                  // + co * Cin * k * k
                for (int co = 0; co < Cout; ++co) acc = fmaf(gp[co], wp[(size_t)co * Cin * k * k], acc);
            }
        }
        dx[e] = acc;
    }
}

// one block per (co, tap): threads stride over ci (coalesced x reads), every thread walks all output pixels in order
__global__ __launch_bounds__(256) void convk_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dw,
                                                          int N, int H, int W, int Cin, int Cout, int k, int s, int OH, int OW) {
    const int co = blockIdx.x / (k * k), t = blockIdx.x % (k * k);
    const int ky = t / k, kx = t % k, p = k / 2;
    for (int ci = threadIdx.x; ci < Cin; ci += 256) {
        float acc = 0.f;
        for (int n = 0; n < N; ++n)
            for (int oy = 0; oy < OH; ++oy) {
                const int iy = oy * s + ky - p;
                if (iy < 0 || iy >= H) continue;
                for (int ox = 0; ox < OW; ++ox) {
                    const int ix = ox * s + kx - p;
                    if (ix < 0 || ix >= W) continue;
                    acc = fmaf(dy[(((size_t)n * OH + oy) * OW + ox) * Cout + co], x[(((size_t)n * H + iy) * W + ix) * Cin + ci], acc);
                }
            }
        dw[(((size_t)co * Cin + ci) * k + ky) * k + kx] = acc;
    }
}

__global__ __launch_bounds__(256) void convk_bgrad_kernel(const float* __restrict__ dy, float* __restrict__ db, long P, int Cout) {
    __shared__ double red[4];
    const int co = blockIdx.x;
    double acc = 0.0;
    for (long q = threadIdx.x; q < P; q += 256) acc += (double)dy[q * Cout + co];
    const double wsum = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = wsum;
    __syncthreads();
    if (threadIdx.x == 0) db[co] = (float)(((red[0] + red[1]) + red[2]) + red[3]);
}
}  // namespace

static bool convk_ok(int N, int H, int W, int Cin, int Cout, int k, int s) {
    return N >= 1 && H >= 1 && W >= 1 && Cin >= 1 && Cout >= 1 && k >= 1 && k <= 11 && (k & 1) && s >= 1 && s <= 4;
}

int pesr_conv_kxk_fwd_launch(const float* x, const float* w, const float* b, float* y, int N, int H, int W, int Cin, int Cout, int k, int s,
                             hipStream_t stream) {
    if (!convk_ok(N, H, W, Cin, Cout, k, s)) return PESR_EINVAL;
    const int OH = (H - 1) / s + 1, OW = (W - 1) / s + 1;
    const long total = (long)N * OH * OW * Cout;
    const int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    hipLaunchKernelGGL(convk_fwd_kernel, dim3(grid), dim3(256), 0, stream, x, w, b, y, N, H, W, Cin, Cout, k, s, OH, OW);
    return pesr_launch_status();
}

int pesr_conv_kxk_dgrad_launch(const float* dy, const float* w, float* dx, int N, int H, int W, int Cin, int Cout, int k, int s,
                               hipStream_t stream) {
    if (!convk_ok(N, H, W, Cin, Cout, k, s)) return PESR_EINVAL;
    const int OH = (H - 1) / s + 1, OW = (W - 1) / s + 1;
    const long total = (long)N * H * W * Cin;
    const int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    hipLaunchKernelGGL(convk_dgrad_kernel, dim3(grid), dim3(256), 0, stream, dy, w, dx, N, H, W, Cin, Cout, k, s, OH, OW);
    return pesr_launch_status();
}

int pesr_conv_kxk_wgrad_launch(const float* x, const float* dy, float* dw, float* db, int N, int H, int W, int Cin, int Cout, int k, int s,
                               hipStream_t stream) {
    if (!convk_ok(N, H, W, Cin, Cout, k, s)) return PESR_EINVAL;
    const int OH = (H - 1) / s + 1, OW = (W - 1) / s + 1;
    hipLaunchKernelGGL(convk_wgrad_kernel, dim3(Cout * k * k), dim3(256), 0, stream, x, dy, dw, N, H, W, Cin, Cout, k, s, OH, OW);
    if (db) hipLaunchKernelGGL(convk_bgrad_kernel, dim3(Cout), dim3(256), 0, stream, dy, db, (long)N * OH * OW, Cout);
    return pesr_launch_status();
}
