// BatchNorm2d (training mode, batch statistics) + LeakyReLU for the Discriminator's BasicBlock
// (reference model/basic.py:29-30, model/pesr.py:47: eps 1e-5, momentum 0.1, slope 0.2), NHWC fp32.
//   stats   : per-channel sum and sum of squares  -> per-block partials (no atomics); every thread accumulates in DOUBLE:
//             dbeta = sum(dz) and the mean are sums of mixed-sign terms that mostly cancel, and an fp32 running sum over
//             ~100 rows loses the bits the result lives in (these kernels are HBM-bound, the fp64 adds are free)
//   finalize: mean, biased var, invstd in double over the partials (fixed order); running stats
//             (unbiased var, momentum) and num_batches_tracked updated as nn.BatchNorm2d does
//   apply   : y = lrelu(gamma * (x - mean) * invstd + beta)         (optionally written NCHW for the
//             flatten in reference model/pesr.py:79)
//   backward: pass 1 reduces sum(dz) and sum(dz * xhat) with dz = dy * lrelu'(z); pass 2 writes
//             dx = gamma * invstd * (dz - mean(dz) - xhat * mean(dz * xhat))
#include "common.h"
#include "launchers.h"
#include "reduce_rows.h"

// partial sums over rows [r0, r1) of a [M][C] array of f(x): which = 0: {x, x^2}; which = 1: {dz, dz*xhat}
template <int WHICH>
__global__ __launch_bounds__(256) void bn_reduce_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                        const float* __restrict__ mean_invstd, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ part, long M, int C,
                                                        long rows_per_block, float slope, long dy_sn, long dy_sc, long dy_sp, long HW) {
    const int C4 = C >> 2;
    const int cw = C4 < 256 ? C4 : 256;
    const int rl = 256 / cw;
    const int tc = threadIdx.x % cw, tr = threadIdx.x / cw;
    const long r0 = (long)blockIdx.x * rows_per_block;
    long r1 = r0 + rows_per_block; if (r1 > M) r1 = M;
    __shared__ f64x4 red[2][256];
    for (int c0 = 0; c0 < C4; c0 += cw) {
        f64x4 s0 = {0.0, 0.0, 0.0, 0.0}, s1 = {0.0, 0.0, 0.0, 0.0};
        const int c4 = c0 + tc;
        if (tr < rl && c4 < C4) {
            f32x4 mu = {0.f, 0.f, 0.f, 0.f}, is = mu, ga = mu, be = mu;
            if (WHICH == 1) {
                mu = ((const f32x4*)mean_invstd)[c4]; is = ((const f32x4*)(mean_invstd + C))[c4];
                ga = ((const f32x4*)gamma)[c4]; be = ((const f32x4*)beta)[c4];
            }
            for (long rr = r0 + tr; rr < r1; rr += rl) {
                const f32x4 v = ((const f32x4*)x)[rr * C4 + c4];
                if (WHICH == 0) {
                    const f64x4 vd = __builtin_convertvector(v, f64x4);
                    s0 += vd; s1 += vd * vd;
                } else {
                    f32x4 g;
                    if (dy_sc == 1) g = ((const f32x4*)dy)[rr * C4 + c4];
                    else {  // dy stored NCHW (flattened classifier input): gather 4 channels
                        const long n = rr / HW, p = rr - n * HW;
                        const float* q = dy + n * dy_sn + p * dy_sp + (long)(c4 * 4) * dy_sc;
                        g = (f32x4){q[0], q[dy_sc], q[2 * dy_sc], q[3 * dy_sc]};
                    }
                    const f32x4 xh = (v - mu) * is;
                    const f32x4 z = ga * xh + be;
                    f32x4 dz;
                    dz.x = z.x > 0.f ? g.x : g.x * slope; dz.y = z.y > 0.f ? g.y : g.y * slope;
                    dz.z = z.z > 0.f ? g.z : g.z * slope; dz.w = z.w > 0.f ? g.w : g.w * slope;
                    const f64x4 dzd = __builtin_convertvector(dz, f64x4);
                    s0 += dzd; s1 += dzd * __builtin_convertvector(xh, f64x4);
                }
            }
        }
        red[0][threadIdx.x] = s0; red[1][threadIdx.x] = s1;
        __syncthreads();
        if (tr == 0 && c4 < C4) {
            for (int k = 1; k < rl; ++k) { s0 += red[0][k * cw + tc]; s1 += red[1][k * cw + tc]; }
            f32x4* p = (f32x4*)part + (size_t)blockIdx.x * 2 * C4;
            p[c4] = __builtin_convertvector(s0, f32x4); p[C4 + c4] = __builtin_convertvector(s1, f32x4);
        }
        __syncthreads();
    }
}

// The same fixed-order two-column row reduce for MANY rows (round 6: one row per pixel tile of a conv kernel whose epilogue leaves the
// sums - up to a few thousand): 1024 threads = CL channel lanes x 1024 / CL row lanes; a thread adds rows rl, rl + RL, ... in increasing
// order (four loads in flight), the row lanes are combined through LDS in lane order.  CL = 64 is reduce_rows_block2's shape (and
// arithmetic); CL = 16 gives a layer of 64 channels four blocks of 64 row lanes instead of one block of 16.
template <int CL>
__device__ __forceinline__ void bn_rows2(const float* __restrict__ part, int nb, int ncols, int col0, int col1, bool valid, double* red,
                                         double* out0, double* out1) {
    constexpr int RL = 1024 / CL;
    const int cl = threadIdx.x % CL, rl = threadIdx.x / CL;
    double s = 0.0, u = 0.0;
    if (valid) {
        int k = rl;
        for (; k + 7 * RL < nb; k += 8 * RL) {      // 16 independent loads in flight; the additions stay in row order
            const float* p0 = part + (size_t)k * ncols;
            float a[8], b[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { a[j] = p0[(size_t)j * RL * ncols + col0]; b[j] = p0[(size_t)j * RL * ncols + col1]; }
#pragma unroll
            for (int j = 0; j < 8; ++j) { s += (double)a[j]; u += (double)b[j]; }
        }
        for (; k < nb; k += RL) { s += (double)part[(size_t)k * ncols + col0]; u += (double)part[(size_t)k * ncols + col1]; }
    }
    __syncthreads();
    red[rl * CL + cl] = s; red[1024 + rl * CL + cl] = u;
    __syncthreads();
    double t0 = 0.0, t1 = 0.0;
    if (rl == 0) {
        t0 = red[cl]; t1 = red[1024 + cl];
        for (int j = 1; j < RL; ++j) { t0 += red[j * CL + cl]; t1 += red[1024 + j * CL + cl]; }
    }
    *out0 = t0; *out1 = t1;
}

// second stage of the statistics (fixed-order row reduce of the per-block partials) and the finalize step in
// ONE launch: block = CL channels; the small layers' BatchNorm is launch-bound, every launch saved is ~4.5 us
template <int CL>
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* __restrict__ part, int nb, int C, long M, float eps,
                                                           float momentum, float* __restrict__ mean_invstd,
                                                           float* __restrict__ running_mean, float* __restrict__ running_var,
                                                           long long* __restrict__ num_batches) {
    __shared__ double red[2048];
    const int c = blockIdx.x * CL + (threadIdx.x % CL);
    double s, ss;
    bn_rows2<CL>(part, nb, 2 * C, c, C + c, c < C, red, &s, &ss);
    if (blockIdx.x == 0 && threadIdx.x == 0 && num_batches) *num_batches += 1;
    if (c >= C || threadIdx.x / CL != 0) return;
    const double mean = s / (double)M;
    double var = ss / (double)M - mean * mean;
    if (var < 0.0) var = 0.0;
    mean_invstd[c] = (float)mean;
    mean_invstd[C + c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unb = M > 1 ? var * (double)M / (double)(M - 1) : var;
        running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * mean);
        running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unb);
    }
}

// second stage for backward, same fusion: sums[0][c] = sum dz, sums[1][c] = sum dz*xhat (+ dbeta, dgamma)
template <int CL>
__global__ __launch_bounds__(1024) void bn_bwd_sums_kernel(const float* __restrict__ part, int nb, int C, float* __restrict__ sums,
                                                           float* __restrict__ dbeta, float* __restrict__ dgamma, int accumulate) {
    __shared__ double red[2048];
    const int c = blockIdx.x * CL + (threadIdx.x % CL);
    double s0, s1;
    bn_rows2<CL>(part, nb, 2 * C, c, C + c, c < C, red, &s0, &s1);
    if (c >= C || threadIdx.x / CL != 0) return;
    const float a = (float)s0, b = (float)s1;
    sums[c] = a; sums[C + c] = b;
    if (dbeta) dbeta[c] = accumulate ? dbeta[c] + a : a;        // dbeta = sum dz, dgamma = sum dz*xhat
    if (dgamma) dgamma[c] = accumulate ? dgamma[c] + b : b;
}

__global__ void bn_apply_kernel(const f32x4* __restrict__ x, const float* __restrict__ mean_invstd, const float* __restrict__ gamma,
                                const float* __restrict__ beta, float* __restrict__ y, long M, int C, float slope, long HW,
                                long ysn, long ysc, long ysp) {
    const int C4 = C >> 2;
    const long total = M * C4;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(e % C4);
        const long rr = e / C4;
        const f32x4 mu = ((const f32x4*)mean_invstd)[c4], is = ((const f32x4*)(mean_invstd + C))[c4];
        const f32x4 ga = ((const f32x4*)gamma)[c4], be = ((const f32x4*)beta)[c4];
        f32x4 z = ga * ((x[e] - mu) * is) + be;
        z.x = z.x > 0.f ? z.x : z.x * slope; z.y = z.y > 0.f ? z.y : z.y * slope;
        z.z = z.z > 0.f ? z.z : z.z * slope; z.w = z.w > 0.f ? z.w : z.w * slope;
        if (ysc == 1) ((f32x4*)y)[e] = z;
        else {
            const long n = rr / HW, p = rr - n * HW;
            float* q = y + n * ysn + p * ysp + (long)(c4 * 4) * ysc;
            q[0] = z.x; q[ysc] = z.y; q[2 * ysc] = z.z; q[3 * ysc] = z.w;
        }
    }
}

// premasked: dy already is dz = dy * lrelu'(z) (written by the conv kernel that produced it, common.h BnEpi mode 2)
__global__ void bn_bwd_apply_kernel(const f32x4* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ mean_invstd,
                                    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ sums,
                                    f32x4* __restrict__ dx, long M, int C, float slope, long HW, long dy_sn, long dy_sc, long dy_sp,
                                    int premasked) {
    const int C4 = C >> 2;
    const long total = M * C4;
    const float invM = 1.0f / (float)M;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(e % C4);
        const long rr = e / C4;
        const f32x4 mu = ((const f32x4*)mean_invstd)[c4], is = ((const f32x4*)(mean_invstd + C))[c4];
        const f32x4 ga = ((const f32x4*)gamma)[c4], be = ((const f32x4*)beta)[c4];
        const f32x4 sdz = ((const f32x4*)sums)[c4] * invM, sdzx = ((const f32x4*)(sums + C))[c4] * invM;
        f32x4 g;
        if (dy_sc == 1) g = ((const f32x4*)dy)[e];
        else {
            const long n = rr / HW, p = rr - n * HW;
            const float* q = dy + n * dy_sn + p * dy_sp + (long)(c4 * 4) * dy_sc;
            g = (f32x4){q[0], q[dy_sc], q[2 * dy_sc], q[3 * dy_sc]};
        }
        const f32x4 xh = (x[e] - mu) * is;
        const f32x4 z = ga * xh + be;
        f32x4 dz = g;
        if (!premasked) {
            dz.x = z.x > 0.f ? g.x : g.x * slope; dz.y = z.y > 0.f ? g.y : g.y * slope;
            dz.z = z.z > 0.f ? g.z : g.z * slope; dz.w = z.w > 0.f ? g.w : g.w * slope;
        }
        dx[e] = ga * is * (dz - sdz - xh * sdzx);
    }
}

// [N][C][HW] -> [N][HW][C] through a 32 x 33 LDS tile (round 6): the gradient of the Discriminator's last block arrives NCHW-contiguous
// (it is the flattened classifier input, reference model/pesr.py:79); read in place, a channel-group lane of the backward kernels
// gathers its four channels HW floats apart - 64 cache lines per wave-load (bn_reduce<1> 43 us on 4.7 MB).  4 us for the copy instead.
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int HW) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z, c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;            // 32 x 8
    const float* s = src + (size_t)n * C * HW;
    float* d = dst + (size_t)n * C * HW;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = c0 + ty + j * 8, p = p0 + tx;
        if (c < C && p < HW) tile[ty + j * 8][tx] = s[(size_t)c * HW + p];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int p = p0 + ty + j * 8, c = c0 + tx;
        if (c < C && p < HW) d[(size_t)p * C + c] = tile[tx][ty + j * 8];
    }
}

namespace {
// many rows (a conv epilogue's pixel tiles, the 3 -> C kernel's 2048 workgroups): 16-channel blocks of 64 row lanes; few rows (bn_reduce's
// <= 512 blocks): the 64-channel blocks of 16 row lanes these kernels always had (same order of additions as before round 6)
static void bn_finalize_launch(const float* part, int rows, int C, long M, float eps, float momentum, float* mean_invstd, float* running_mean,
                               float* running_var, long long* num_batches, hipStream_t stream) {
    if (rows > 512)
        hipLaunchKernelGGL(bn_finalize_kernel<16>, dim3((C + 15) / 16), dim3(1024), 0, stream, part, rows, C, M, eps, momentum, mean_invstd,
                           running_mean, running_var, num_batches);
    else
        hipLaunchKernelGGL(bn_finalize_kernel<64>, dim3((C + 63) / 64), dim3(1024), 0, stream, part, rows, C, M, eps, momentum, mean_invstd,
                           running_mean, running_var, num_batches);
}
static void bn_bwd_sums_launch(const float* part, int rows, int C, float* sums, float* dbeta, float* dgamma, int accumulate, hipStream_t stream) {
    if (rows > 512)
        hipLaunchKernelGGL(bn_bwd_sums_kernel<16>, dim3((C + 15) / 16), dim3(1024), 0, stream, part, rows, C, sums, dbeta, dgamma, accumulate);
    else
        hipLaunchKernelGGL(bn_bwd_sums_kernel<64>, dim3((C + 63) / 64), dim3(1024), 0, stream, part, rows, C, sums, dbeta, dgamma, accumulate);
}
static void bn_grid(long M, long* nb, long* rpb) {
    long b = (M + 63) / 64; if (b > 512) b = 512; if (b < 1) b = 1;
    *rpb = (M + b - 1) / b;
    *nb = (M + *rpb - 1) / *rpb;
}
}  // namespace

static size_t bn_ws_base(long M, int C) {
    long nb, rpb; bn_grid(M, &nb, &rpb);
    return ((size_t)nb * 2 * C * sizeof(float) + 2 * (size_t)C * sizeof(float) + 2 * (size_t)C * sizeof(double) + 512 + 255) / 256 * 256;
}
// + room for an NHWC copy of an NCHW gradient (small tensors only: the layer in front of the classifier)
size_t pesr_bn_ws_bytes(long M, int C) {
    const size_t copy = (size_t)M * C * sizeof(float);
    return bn_ws_base(M, C) + (copy <= ((size_t)64 << 20) ? copy : 0);
}

// forward: x [M][C] (M = N*H*W) -> y; saves mean_invstd [2][C]
int pesr_bn_lrelu_fwd_launch(const float* x, const float* gamma, const float* beta, float* y, float* mean_invstd,
                             float* running_mean, float* running_var, long long* num_batches, long M, int C, long HW, float eps,
                             float momentum, float slope, int y_nchw, void* ws, size_t ws_bytes, hipStream_t stream) {
    if (C % 4) return PESR_EINVAL;
    long nb, rpb; bn_grid(M, &nb, &rpb);
    const size_t dsum_bytes = ((size_t)2 * C * sizeof(double) + 255) / 256 * 256;
    if (!ws || ws_bytes < dsum_bytes + (size_t)nb * 2 * C * sizeof(float)) return PESR_EWORKSPACE;
    double* dsum = (double*)ws;
    float* part = (float*)((char*)ws + dsum_bytes);
    hipLaunchKernelGGL(bn_reduce_kernel<0>, dim3((unsigned)nb), dim3(256), 0, stream, x, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, part, M, C, rpb, slope, 0L, 1L, 0L, HW);
    bn_finalize_launch(part, (int)nb, C, M, eps, momentum, mean_invstd, running_mean, running_var, num_batches, stream);
    const long total = M * (C / 4);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    long ysn = HW * C, ysc = 1, ysp = C;
    if (y_nchw) { ysn = HW * C; ysc = HW; ysp = 1; }
    hipLaunchKernelGGL(bn_apply_kernel, dim3(grid), dim3(256), 0, stream, (const f32x4*)x, (const float*)mean_invstd, gamma, beta, y, M, C,
                       slope, HW, ysn, ysc, ysp);
    return pesr_launch_status();
}

// conv 3 -> C (no bias) -> BatchNorm (training) -> LeakyReLU with the statistics taken from the conv kernel's epilogue (SURVEY K10's
// first half, for the layer where it pays most: the Discriminator's features.0, 16 x 192 x 192 x 64 = 151 MB): three launches - conv +
// per-workgroup partial sums, finalize, apply - instead of four, and one pass over z less.  z (saved for backward) and y are both
// written.  Workspace: pesr_conv_rgb_bn_ws_bytes.
size_t pesr_conv_rgb_bn_ws_bytes(int N, int H, int W, int C) {
    const int rows = pesr_conv_rgb_in_stats_rows(N, H, W, C);
    return rows ? (size_t)rows * 2 * C * sizeof(float) + 512 : 0;
}

int pesr_conv_rgb_bn_lrelu_fwd_launch(const float* x, const float* w, float* z, const float* gamma, const float* beta, float* y,
                                      float* mean_invstd, float* running_mean, float* running_var, long long* num_batches, int N, int H,
                                      int W, int C, float eps, float momentum, float slope, int y_nchw, void* ws, size_t ws_bytes,
                                      hipStream_t stream) {
    const int rows = pesr_conv_rgb_in_stats_rows(N, H, W, C);
    if (!rows) return PESR_EINVAL;
    if (!ws || ws_bytes < (size_t)rows * 2 * C * sizeof(float)) return PESR_EWORKSPACE;
    float* part = (float*)ws;
    int rc = pesr_conv_rgb_in_stats_launch(x, w, z, part, N, H, W, C, stream);
    if (rc) return rc;
    const long M = (long)N * H * W, HW = (long)H * W;
    bn_finalize_launch(part, rows, C, M, eps, momentum, mean_invstd, running_mean, running_var, num_batches, stream);
    const long total = M * (C / 4);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    long ysn = HW * C, ysc = 1, ysp = C;
    if (y_nchw) { ysn = HW * C; ysc = HW; ysp = 1; }
    hipLaunchKernelGGL(bn_apply_kernel, dim3(grid), dim3(256), 0, stream, (const f32x4*)z, (const float*)mean_invstd, gamma, beta, y, M, C,
                       slope, HW, ysn, ysc, ysp);
    return pesr_launch_status();
}

// backward: dy is the gradient w.r.t. the LeakyReLU output (NHWC, or NCHW when dy_nchw); dgamma/dbeta may be NULL
int pesr_bn_lrelu_bwd_launch(const float* x, const float* dy, const float* gamma, const float* beta, const float* mean_invstd,
                             float* dx, float* dgamma, float* dbeta, long M, int C, long HW, float slope, int dy_nchw, int accumulate,
                             void* ws, size_t ws_bytes, hipStream_t stream) {
    if (C % 4) return PESR_EINVAL;
    long nb, rpb; bn_grid(M, &nb, &rpb);
    const size_t part_bytes = (size_t)nb * 2 * C * sizeof(float);
    const size_t dsum_bytes = ((size_t)2 * C * sizeof(double) + 255) / 256 * 256;
    if (!ws || ws_bytes < dsum_bytes + part_bytes + 2 * (size_t)C * sizeof(float)) return PESR_EWORKSPACE;
    double* dsum = (double*)ws;
    float* part = (float*)((char*)ws + dsum_bytes);
    float* sums = (float*)((char*)ws + dsum_bytes + part_bytes);
    long sn = HW * C, sc = 1, sp = C;
    if (dy_nchw) {
        const size_t base = bn_ws_base(M, C), copy = (size_t)M * C * sizeof(float);
        if (ws_bytes >= base + copy && HW > 1) {      // transpose once, then the coalesced NHWC paths
            float* dyt = (float*)((char*)ws + base);
            hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((unsigned)((HW + 31) / 32), (unsigned)((C + 31) / 32), (unsigned)(M / HW)), dim3(256), 0, stream,
                               dy, dyt, C, (int)HW);
            dy = dyt;
        } else {
            sn = HW * C; sc = HW; sp = 1;
        }
    }
    hipLaunchKernelGGL(bn_reduce_kernel<1>, dim3((unsigned)nb), dim3(256), 0, stream, x, dy, mean_invstd, gamma, beta, part, M, C, rpb, slope,
                       sn, sc, sp, HW);
    bn_bwd_sums_launch(part, (int)nb, C, sums, dbeta, dgamma, accumulate, stream);
    const long total = M * (C / 4);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid), dim3(256), 0, stream, (const f32x4*)x, dy, mean_invstd, gamma, beta, (const float*)sums,
                       (f32x4*)dx, M, C, slope, HW, sn, sc, sp, 0);
    return pesr_launch_status();
}

// ---- round 6: the two halves of the BatchNorm whose sums come out of a conv kernel's epilogue (common.h BnEpi) -----------------------
// forward: part[rows][2][C] (sum, sum of squares per pixel tile) -> mean / invstd / running statistics; the apply pass is
// pesr_bn_lrelu_apply_launch.  backward: part[rows][2][C] (sum g', sum g' * xhat; g' = the already masked gradient the conv kernel
// stored) -> dgamma / dbeta and dz = gamma * invstd * (g' - mean(g') - xhat * mean(g' * xhat)).
int pesr_bn_finalize_launch(const float* part, int rows, int C, long M, float eps, float momentum, float* mean_invstd, float* running_mean,
                            float* running_var, long long* num_batches, hipStream_t stream) {
    if (C % 4 || rows < 1 || !part) return PESR_EINVAL;
    bn_finalize_launch(part, rows, C, M, eps, momentum, mean_invstd, running_mean, running_var, num_batches, stream);
    return pesr_launch_status();
}

int pesr_bn_lrelu_bwd_fused_launch(const float* z, const float* gmasked, const float* part, int rows, const float* gamma, const float* beta,
                                   const float* mean_invstd, float* dz, float* dgamma, float* dbeta, long M, int C, long HW, int accumulate,
                                   void* ws, size_t ws_bytes, hipStream_t stream) {
    if (C % 4 || rows < 1 || !part) return PESR_EINVAL;
    if (!ws || ws_bytes < 2 * (size_t)C * sizeof(float)) return PESR_EWORKSPACE;
    float* sums = (float*)ws;
    bn_bwd_sums_launch(part, rows, C, sums, dbeta, dgamma, accumulate, stream);
    const long total = M * (C / 4);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid), dim3(256), 0, stream, (const f32x4*)z, gmasked, mean_invstd, gamma, beta, (const float*)sums,
                       (f32x4*)dz, M, C, 0.f, HW, HW * C, 1L, (long)C, 1);
    return pesr_launch_status();
}

// ---- eval-mode BatchNorm (running statistics; nn.BatchNorm2d in .eval(), a constructor branch of reference model/basic.py:29) ----
// mean_invstd [2][C] is GIVEN (mean = running_mean, invstd = 1/sqrt(running_var + eps), computed by the caller).
int pesr_bn_lrelu_apply_launch(const float* x, const float* gamma, const float* beta, const float* mean_invstd, float* y, long M,
                               int C, long HW, float slope, int y_nchw, hipStream_t stream) {
    if (C % 4) return PESR_EINVAL;
    const long total = M * (C / 4);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    long ysn = HW * C, ysc = 1, ysp = C;
    if (y_nchw) { ysn = HW * C; ysc = HW; ysp = 1; }
    hipLaunchKernelGGL(bn_apply_kernel, dim3(grid), dim3(256), 0, stream, (const f32x4*)x, mean_invstd, gamma, beta, y, M, C, slope, HW,
                       ysn, ysc, ysp);
    return pesr_launch_status();
}

// backward with fixed statistics: dx = gamma * invstd * dz, dgamma = sum dz * xhat, dbeta = sum dz  (dz = dy * lrelu'(z));
// the batch-statistics terms of the training-mode formula vanish, i.e. bn_bwd_apply_kernel with zero sums.
int pesr_bn_lrelu_bwd_eval_launch(const float* x, const float* dy, const float* gamma, const float* beta, const float* mean_invstd,
                                  float* dx, float* dgamma, float* dbeta, long M, int C, long HW, float slope, int dy_nchw, void* ws,
                                  size_t ws_bytes, hipStream_t stream) {
    if (C % 4) return PESR_EINVAL;
    long nb, rpb; bn_grid(M, &nb, &rpb);
    const size_t part_bytes = (size_t)nb * 2 * C * sizeof(float);
    const size_t dsum_bytes = ((size_t)2 * C * sizeof(double) + 255) / 256 * 256;
    if (!ws || ws_bytes < dsum_bytes + part_bytes + 2 * (size_t)C * sizeof(float)) return PESR_EWORKSPACE;
    double* dsum = (double*)ws;
    float* part = (float*)((char*)ws + dsum_bytes);
    float* sums = (float*)((char*)ws + dsum_bytes + part_bytes);
    long sn = HW * C, sc = 1, sp = C;
    if (dy_nchw) { sn = HW * C; sc = HW; sp = 1; }
    if (dgamma || dbeta) {
        hipLaunchKernelGGL(bn_reduce_kernel<1>, dim3((unsigned)nb), dim3(256), 0, stream, x, dy, mean_invstd, gamma, beta, part, M, C, rpb,
                           slope, sn, sc, sp, HW);
        bn_bwd_sums_launch(part, (int)nb, C, sums, dbeta, dgamma, 0, stream);
    }
    hipError_t e = hipMemsetAsync(sums, 0, 2 * (size_t)C * sizeof(float), stream);
    if (e != hipSuccess) return (int)e;
    const long total = M * (C / 4);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid), dim3(256), 0, stream, (const f32x4*)x, dy, mean_invstd, gamma, beta, (const float*)sums,
                       (f32x4*)dx, M, C, slope, HW, sn, sc, sp, 0);
    return pesr_launch_status();
}

// ---- backward OF the training-mode BatchNorm backward (gradient penalty, reference train.py:216-226: the penalty is a function
// of dD/dx, so its gradient needs d(dz)/d(du, z, gamma) of  dz = gamma * s * (du - mean(du) - xhat * mean(du * xhat)) ) -------------
// With g = dL/d(dz) and the channel means m1 = mean(du), m2 = mean(du xhat), G1 = mean(g), Gx = mean(g xhat),
// Ga = mean(g (du - m1)) (derivation checked against autograd in fp64, tests/test_entrypoints_cpu.py):
//     L_du    = gamma s (g - G1 - xhat Gx)                      (the map du -> dz is self-adjoint)
//     L_z     = -gamma s^2 [ xhat (Ga - 3 m2 Gx) + m2 (g - G1) + Gx (du - m1) ]
//     L_gamma = s (sum(g du) - m1 sum(g) - m2 sum(g xhat))
// pass 1: five per-channel sums {du, du xhat, g, g xhat, g du}, double accumulators, per-block partials; pass 2: elementwise.
__global__ __launch_bounds__(256) void bn_bwd2_reduce_kernel(const float* __restrict__ z, const float* __restrict__ du,
                                                             const float* __restrict__ g, const float* __restrict__ mean_invstd,
                                                             float* __restrict__ part, long M, int C, long rows_per_block) {
    const int C4 = C >> 2;
    const int cw = C4 < 256 ? C4 : 256;
    const int rl = 256 / cw;
    const int tc = threadIdx.x % cw, tr = threadIdx.x / cw;
    const long r0 = (long)blockIdx.x * rows_per_block;
    long r1 = r0 + rows_per_block; if (r1 > M) r1 = M;
    __shared__ f64x4 red[256];
    for (int c0 = 0; c0 < C4; c0 += cw) {
        f64x4 s[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) s[k] = (f64x4){0.0, 0.0, 0.0, 0.0};
        const int c4 = c0 + tc;
        if (tr < rl && c4 < C4) {
            const f32x4 mu = ((const f32x4*)mean_invstd)[c4], is = ((const f32x4*)(mean_invstd + C))[c4];
            for (long rr = r0 + tr; rr < r1; rr += rl) {
                const f32x4 zv = ((const f32x4*)z)[rr * C4 + c4], dv = ((const f32x4*)du)[rr * C4 + c4], gv = ((const f32x4*)g)[rr * C4 + c4];
                const f64x4 xh = __builtin_convertvector((zv - mu) * is, f64x4);
                const f64x4 dd = __builtin_convertvector(dv, f64x4), gd = __builtin_convertvector(gv, f64x4);
                s[0] += dd; s[1] += dd * xh; s[2] += gd; s[3] += gd * xh; s[4] += gd * dd;
            }
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) {      // one LDS round per quantity keeps the buffer at 8 KiB
            red[threadIdx.x] = s[k];
            __syncthreads();
            if (tr == 0 && c4 < C4) {
                f64x4 t = s[k];
                for (int q = 1; q < rl; ++q) t += red[q * cw + tc];
                ((f32x4*)part)[((size_t)blockIdx.x * 5 + k) * C4 + c4] = __builtin_convertvector(t, f32x4);
            }
            __syncthreads();
        }
    }
}

__global__ void bn_bwd2_apply_kernel(const f32x4* __restrict__ z, const f32x4* __restrict__ du, const f32x4* __restrict__ g,
                                     const float* __restrict__ mean_invstd, const float* __restrict__ gamma,
                                     const double* __restrict__ dsum, f32x4* __restrict__ l_du, f32x4* __restrict__ l_z,
                                     float* __restrict__ l_gamma, long M, int C) {
    const int C4 = C >> 2;
    const long total = M * C4;
    const double invM = 1.0 / (double)M;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(e % C4);
        const f32x4 mu = ((const f32x4*)mean_invstd)[c4], is = ((const f32x4*)(mean_invstd + C))[c4], ga = ((const f32x4*)gamma)[c4];
        f32x4 m1, m2, G1, Gx, Ga;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = c4 * 4 + k;
            const double a = dsum[c] * invM, b = dsum[C + c] * invM, cc = dsum[2 * C + c] * invM, d = dsum[3 * C + c] * invM;
            m1[k] = (float)a; m2[k] = (float)b; G1[k] = (float)cc; Gx[k] = (float)d;
            Ga[k] = (float)(dsum[4 * C + c] * invM - cc * a);
        }
        const f32x4 xh = (z[e] - mu) * is, dv = du[e], gv = g[e];
        if (l_du) l_du[e] = ga * is * (gv - G1 - xh * Gx);
        if (l_z) l_z[e] = (0.f - ga) * is * is * (xh * (Ga - 3.0f * m2 * Gx) + m2 * (gv - G1) + Gx * (dv - m1));
        if (l_gamma && e < C4) {      // the first C4 threads also emit L_gamma for their channels
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = c4 * 4 + k;
                const double a = dsum[c] * invM, b = dsum[C + c] * invM;
                l_gamma[c] = (float)((double)mean_invstd[C + c] * (dsum[4 * C + c] - a * dsum[2 * C + c] - b * dsum[3 * C + c]));
            }
        }
    }
}

size_t pesr_bn_bwd_bwd_ws_bytes(long M, int C) {
    long nb, rpb; bn_grid(M, &nb, &rpb);
    return (size_t)nb * 5 * C * sizeof(float) + 5 * (size_t)C * sizeof(double) + 512;
}

int pesr_bn_bwd_bwd_launch(const float* z, const float* du, const float* g, const float* gamma, const float* mean_invstd, float* l_du,
                           float* l_z, float* l_gamma, long M, int C, void* ws, size_t ws_bytes, hipStream_t stream) {
    if (C % 4) return PESR_EINVAL;
    long nb, rpb; bn_grid(M, &nb, &rpb);
    const size_t dsum_bytes = ((size_t)5 * C * sizeof(double) + 255) / 256 * 256;
    if (!ws || ws_bytes < dsum_bytes + (size_t)nb * 5 * C * sizeof(float)) return PESR_EWORKSPACE;
    double* dsum = (double*)ws;
    float* part = (float*)((char*)ws + dsum_bytes);
    hipLaunchKernelGGL(bn_bwd2_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, stream, z, du, g, mean_invstd, part, M, C, rpb);
    int rc0 = pesr_reduce_rows_launch(part, dsum, (int)nb, 5 * C, stream);
    if (rc0) return rc0;
    const long total = M * (C / 4);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(bn_bwd2_apply_kernel, dim3(grid), dim3(256), 0, stream, (const f32x4*)z, (const f32x4*)du, (const f32x4*)g, mean_invstd,
                       gamma, (const double*)dsum, (f32x4*)l_du, (f32x4*)l_z, l_gamma, M, C);
    return pesr_launch_status();
}
