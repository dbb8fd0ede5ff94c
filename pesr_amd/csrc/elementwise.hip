// HBM-bound helpers of the PESR train step (gfx950): MeanShift 1x1 conv, standalone PixelShuffle,
// ReLU gradient masking, 2x2 max-pool.  All tensors NHWC fp32; 16-byte vector accesses where the channel
// count allows.
#include "common.h"
#include "launchers.h"

// ------------------------------------------------------------------------------------------------
// MeanShift: y[p][i] = sum_j w[i][j] x[p][j] + b[i]   (reference model/basic.py:9-17, a trainable 1x1 conv)
// Input/output element strides are explicit so the NCHW <-> NHWC change of the 3-channel boundary
// tensors is folded into this kernel: element (n, c, y, x) lives at n*sn + c*sc + (y*W + x)*sp.
// ------------------------------------------------------------------------------------------------
__global__ void meanshift_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                     float* __restrict__ y, int N, long HW, long xsn, long xsc, long xsp, long ysn, long ysc,
                                     long ysp) {
    const float w00 = w[0], w01 = w[1], w02 = w[2], w10 = w[3], w11 = w[4], w12 = w[5], w20 = w[6], w21 = w[7], w22 = w[8];
    const float b0 = b[0], b1 = b[1], b2 = b[2];
    const long total = (long)N * HW;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long n = e / HW, p = e - n * HW;
        const float* xi = x + n * xsn + p * xsp;
        const float x0 = xi[0], x1 = xi[xsc], x2 = xi[2 * xsc];
        float* yo = y + n * ysn + p * ysp;
        // same accumulation order as a direct conv: bias + sum_j w*x
        yo[0] = fmaf(w02, x2, fmaf(w01, x1, fmaf(w00, x0, b0)));
        yo[ysc] = fmaf(w12, x2, fmaf(w11, x1, fmaf(w10, x0, b1)));
        yo[2 * ysc] = fmaf(w22, x2, fmaf(w21, x1, fmaf(w20, x0, b2)));
    }
}

// backward: dx[p][j] = sum_i w[i][j] dy[p][i];  dw[i][j] = sum_p dy[p][i] x[p][j];  db[i] = sum_p dy[p][i]
// dy, dx are NHWC [P][3]; x has explicit strides.  part: [blocks][12] partial sums.
__global__ __launch_bounds__(256) void meanshift_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ w, float* __restrict__ dx,
                                                            float* __restrict__ part, int N, long HW, long xsn, long xsc,
                                                            long xsp) {
    const float w00 = w[0], w01 = w[1], w02 = w[2], w10 = w[3], w11 = w[4], w12 = w[5], w20 = w[6], w21 = w[7], w22 = w[8];
    // double accumulators: dw = sum over all pixels of dy * x with x in 0..255 is a sum of large mixed-sign terms that mostly
    // cancel; an fp32 running sum per thread measured 19 x the error of the CPU reference's fp32 result on this tensor
    double s[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) s[k] = 0.0;
    const long total = (long)N * HW;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long n = e / HW, p = e - n * HW;
        const float g0 = dy[e * 3], g1 = dy[e * 3 + 1], g2 = dy[e * 3 + 2];
        if (dx) {
            dx[e * 3] = fmaf(w20, g2, fmaf(w10, g1, w00 * g0));
            dx[e * 3 + 1] = fmaf(w21, g2, fmaf(w11, g1, w01 * g0));
            dx[e * 3 + 2] = fmaf(w22, g2, fmaf(w12, g1, w02 * g0));
        }
        const float* xi = x + n * xsn + p * xsp;
        const float x0 = xi[0], x1 = xi[xsc], x2 = xi[2 * xsc];
        s[0] += (double)(g0 * x0); s[1] += (double)(g0 * x1); s[2] += (double)(g0 * x2);
        s[3] += (double)(g1 * x0); s[4] += (double)(g1 * x1); s[5] += (double)(g1 * x2);
        s[6] += (double)(g2 * x0); s[7] += (double)(g2 * x1); s[8] += (double)(g2 * x2);
        s[9] += (double)g0; s[10] += (double)g1; s[11] += (double)g2;
    }
    __shared__ double red[4][12];
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const double v = wave_sum_d(s[k]);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 12) part[blockIdx.x * 12 + threadIdx.x] = (float)(red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}
__global__ void meanshift_bwd_final_kernel(const double* __restrict__ dsum, float* __restrict__ dw, float* __restrict__ db) {
    if (threadIdx.x < 12) {
        const double s = dsum[threadIdx.x];
        if (threadIdx.x < 9) dw[threadIdx.x] = (float)s; else db[threadIdx.x - 9] = (float)s;
    }
}

int pesr_meanshift_fwd_launch(const float* x, const float* w, const float* b, float* y, int N, int H, int W, long xsn, long xsc,
                              long xsp, long ysn, long ysc, long ysp, hipStream_t stream) {
    const long total = (long)N * H * W;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(meanshift_fwd_kernel, dim3(grid), dim3(256), 0, stream, x, w, b, y, N, (long)H * W, xsn, xsc, xsp, ysn, ysc, ysp);
    return pesr_launch_status();
}
int pesr_meanshift_bwd_launch(const float* dy, const float* x, const float* w, float* dx, float* dw, float* db, int N, int H,
                              int W, long xsn, long xsc, long xsp, void* ws, size_t ws_bytes, hipStream_t stream) {
    const int nb = 1024;
    if (!ws || ws_bytes < 128 + (size_t)nb * 12 * sizeof(float)) return PESR_EWORKSPACE;
    double* dsum = (double*)ws;                 // 12 doubles (padded to 128 B), then the partials
    float* part = (float*)((char*)ws + 128);
    hipLaunchKernelGGL(meanshift_bwd_kernel, dim3(nb), dim3(256), 0, stream, dy, x, w, dx, part, N, (long)H * W, xsn, xsc, xsp);
    const int rc = pesr_reduce_rows_launch(part, dsum, nb, 12, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(meanshift_bwd_final_kernel, dim3(1), dim3(64), 0, stream, (const double*)dsum, dw, db);
    return pesr_launch_status();
}

// ------------------------------------------------------------------------------------------------
// PixelShuffle(2) standalone (reference model/basic.py:57,59): out[n][2h+i][2w+j][c] = in[n][h][w][4c+2i+j].
// Pure indexing -> bit-exact.  The conv kernels fuse this; these exist for the API and as their check.
// ------------------------------------------------------------------------------------------------
__global__ void pixel_shuffle_kernel(const float* __restrict__ in, float* __restrict__ out, int N, int H, int W, int C, int inverse) {
    // C = channels of the shuffled (large) tensor; the small tensor has 4C
    const long total = (long)N * H * W * 4 * C;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        // e indexes the shuffled tensor [N][2H][2W][C]
        const int c = (int)(e % C);
        long rest = e / C;
        const int X = (int)(rest % (2 * W)); rest /= (2 * W);
        const int Y = (int)(rest % (2 * H));
        const int n = (int)(rest / (2 * H));
        const long small = ((((long)n * H + (Y >> 1)) * W + (X >> 1)) * 4 * C) + 4 * c + 2 * (Y & 1) + (X & 1);
        if (inverse) out[small] = in[e]; else out[e] = in[small];
    }
}
int pesr_pixel_shuffle_launch(const float* in, float* out, int N, int H, int W, int C, int inverse, hipStream_t stream) {
    const long total = (long)N * H * W * 4 * C;
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(pixel_shuffle_kernel, dim3(grid), dim3(256), 0, stream, in, out, N, H, W, C, inverse);
    return pesr_launch_status();
}

// ------------------------------------------------------------------------------------------------
// out = (ref > 0) ? alpha*g : slope*alpha*g   (+ add) : (Leaky)ReLU backward with optional scale / fan-in add
// ------------------------------------------------------------------------------------------------
__global__ void relu_mask_kernel(const f32x4* __restrict__ g, const f32x4* __restrict__ ref, const f32x4* __restrict__ add,
                                 f32x4* __restrict__ out, long n4, float alpha, float slope) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (long)gridDim.x * blockDim.x) {
        f32x4 v = g[e] * alpha;
        if (ref) {
            const f32x4 r = ref[e];
            v.x = r.x > 0.f ? v.x : v.x * slope; v.y = r.y > 0.f ? v.y : v.y * slope; v.z = r.z > 0.f ? v.z : v.z * slope; v.w = r.w > 0.f ? v.w : v.w * slope;
        }
        if (add) v += add[e];
        out[e] = v;
    }
}
int pesr_relu_mask_launch(const float* g, const float* ref, const float* add, float* out, long n, float alpha, float slope, hipStream_t stream) {
    if (n % 4) return PESR_EINVAL;
    const long n4 = n / 4;
    const int grid = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
    hipLaunchKernelGGL(relu_mask_kernel, dim3(grid), dim3(256), 0, stream, (const f32x4*)g, (const f32x4*)ref, (const f32x4*)add, (f32x4*)out, n4, alpha, slope);
    return pesr_launch_status();
}

// ------------------------------------------------------------------------------------------------
// 2x2/2 max-pool (torchvision vgg19 features 4, 9, 18, 27; reference model/vgg.py:8-10), NHWC, C % 4 == 0.
// Backward recomputes the arg-max from the saved input (first maximum in (dy, dx) scan order, as ATen)
// and, with relu_in = 1, also applies the preceding ReLU's mask (gradient only where the input > 0).
// ------------------------------------------------------------------------------------------------
__global__ void maxpool_fwd_kernel(const f32x4* __restrict__ x, f32x4* __restrict__ y, int N, int H, int W, int C4) {
    const int OH = H >> 1, OW = W >> 1;
    const long total = (long)N * OH * OW * C4;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)(e % C4);
        long rest = e / C4;
        const int ox = (int)(rest % OW); rest /= OW;
        const int oy = (int)(rest % OH);
        const int n = (int)(rest / OH);
        const f32x4* p = x + (((long)n * H + 2 * oy) * W + 2 * ox) * C4 + c;
        const f32x4 a = p[0], b = p[C4], cc = p[(long)W * C4], d = p[(long)W * C4 + C4];
        f32x4 m;
        m.x = fmaxf(fmaxf(a.x, b.x), fmaxf(cc.x, d.x)); m.y = fmaxf(fmaxf(a.y, b.y), fmaxf(cc.y, d.y));
        m.z = fmaxf(fmaxf(a.z, b.z), fmaxf(cc.z, d.z)); m.w = fmaxf(fmaxf(a.w, b.w), fmaxf(cc.w, d.w));
        y[e] = m;
    }
}
// which of (a, b, c, d) is the first maximum (ATen scan order); -1 if relu_in and the maximum is not > 0
__device__ __forceinline__ int argmax4(float a, float b, float c, float d, int relu_in) {
    int k = 0; float m = a;
    if (b > m) { m = b; k = 1; }
    if (c > m) { m = c; k = 2; }
    if (d > m) { m = d; k = 3; }
    return (relu_in && !(m > 0.f)) ? -1 : k;
}
__global__ void maxpool_bwd_kernel(const f32x4* __restrict__ x, const f32x4* __restrict__ dy, f32x4* __restrict__ dx, int N, int H,
                                   int W, int C4, int relu_in) {
    const int OH = H >> 1, OW = W >> 1;
    const long total = (long)N * OH * OW * C4;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)(e % C4);
        long rest = e / C4;
        const int ox = (int)(rest % OW); rest /= OW;
        const int oy = (int)(rest % OH);
        const int n = (int)(rest / OH);
        const long base = (((long)n * H + 2 * oy) * W + 2 * ox) * C4 + c;
        const f32x4 a = x[base], b = x[base + C4], cc = x[base + (long)W * C4], d = x[base + (long)W * C4 + C4];
        const f32x4 g = dy[e];
        f32x4 oa, ob, oc, od;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = argmax4(a[j], b[j], cc[j], d[j], relu_in);
            oa[j] = k == 0 ? g[j] : 0.f; ob[j] = k == 1 ? g[j] : 0.f; oc[j] = k == 2 ? g[j] : 0.f; od[j] = k == 3 ? g[j] : 0.f;
        }
        dx[base] = oa; dx[base + C4] = ob; dx[base + (long)W * C4] = oc; dx[base + (long)W * C4 + C4] = od;
    }
}
int pesr_maxpool2x2_fwd_launch(const float* x, float* y, int N, int H, int W, int C, hipStream_t stream) {
    if (C % 4 || H % 2 || W % 2) return PESR_EINVAL;
    const long total = (long)N * (H / 2) * (W / 2) * (C / 4);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid), dim3(256), 0, stream, (const f32x4*)x, (f32x4*)y, N, H, W, C / 4);
    return pesr_launch_status();
}
int pesr_maxpool2x2_bwd_launch(const float* x, const float* dy, float* dx, int N, int H, int W, int C, int relu_in, hipStream_t stream) {
    if (C % 4 || H % 2 || W % 2) return PESR_EINVAL;
    const long total = (long)N * (H / 2) * (W / 2) * (C / 4);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid), dim3(256), 0, stream, (const f32x4*)x, (const f32x4*)dy, (f32x4*)dx, N, H, W, C / 4, relu_in);
    return pesr_launch_status();
}
