// Skinny-batch Linear for the Discriminator's classifier (reference model/pesr.py:69-74: Linear(73728, 1024)
// -> LeakyReLU(0.2) -> Linear(1024, 1); ATen addmm / mm in forward and backward).  M (batch) <= 32.
// All three passes are HBM-bound on the 302 MB weight matrix, which each streams exactly once:
//   fwd  : y[m][n]  = act(sum_k x[m][k] W[n][k] + b[n])     split-K partials + fixed-order finalize
//   dgrad: dx[m][k] = sum_n dy[m][n] W[n][k]                 one launch, the N split over a workgroup's waves
//   wgrad: dW[n][k] = sum_m dy[m][n] x[m][k],  db[n] = sum_m dy[m][n]
#include <mutex>
#include "common.h"
#include "launchers.h"

#define LIN_MAXM 32

// ---- forward -----------------------------------------------------------------------------------
// One wave: NR consecutive output features, one K slice; lanes stride over K with 16-byte loads, so every load instruction of
// the wave reads ONE KiB of ONE weight row (lane-linear): the access pattern the memory system serves best - the MFMA form
// below reads 16 rows x 64 bytes per instruction and stays at 2.4 TB/s.  The x slice of the iteration (M rows x 1 KiB) comes
// from L1 / L2: M / NR times the weight bytes, shared by the waves of a block (same K slice, neighbouring feature groups).
// Used for 16 < M <= 32 (NR = 2); M <= 16 runs on linear_fwd_lds_kernel below.
template <int MB, int NR>
__global__ __launch_bounds__(256, 2) void linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                         float* __restrict__ part, int M, int N, long K, int ksplit, long kchunk) {
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const int ngroups = (N + NR - 1) / NR;
    const int ng = wave % ngroups, ks = wave / ngroups;
    if (ks >= ksplit) return;
    const int n0 = ng * NR;
    const long k0 = ks * kchunk;
    long k1 = k0 + kchunk; if (k1 > K) k1 = K;
    float acc[NR][MB];
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int m = 0; m < MB; ++m) acc[r][m] = 0.f;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    for (long k = k0 + lane * 4; k < k1; k += 256) {
        f32x4 w[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r)
            w[r] = (n0 + r < N) ? __builtin_nontemporal_load((const f32x4*)(W + (size_t)(n0 + r) * K + k)) : zero;
#pragma unroll
        for (int mb = 0; mb < MB; mb += 8) {          // the x rows in batches of 8 loads
            f32x4 xv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) xv[j] = (mb + j < M) ? *(const f32x4*)(x + (size_t)(mb + j) * K + k) : zero;
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int r = 0; r < NR; ++r)
                    acc[r][mb + j] = fmaf(w[r].w, xv[j].w, fmaf(w[r].z, xv[j].z, fmaf(w[r].y, xv[j].y, fmaf(w[r].x, xv[j].x, acc[r][mb + j]))));
        }
    }
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            const float s = wave_sum(acc[r][m]);
            if (lane == 0 && m < M && n0 + r < N) part[((size_t)ks * M + m) * N + n0 + r] = s;
        }
}
// M <= 16, the layer that matters (Linear(73728, 1024) at batch 16): a block's 4 waves take 8 features each (32 in all) of ONE K
// slice; per 256-k step the block stages the x piece (M rows x 1 KiB) through LDS once - each wave fetches four of the rows,
// double-buffered, one barrier per step - so x costs 0.5 x the weight bytes of vector-memory traffic instead of 2 x, and a lane
// holds only one x piece at a time next to its 8 weight pieces and 128 accumulators (no spills at 2 waves per SIMD).
template <int MB>
__global__ __launch_bounds__(256, 2) void linear_fwd_lds_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                                float* __restrict__ part, int M, int N, long K, int ksplit, long kchunk) {
    constexpr int NR = 8, RW = MB / 4;                      // x rows fetched per wave and step
    __shared__ f32x4 xs[2][MB][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nblk = (N + 4 * NR - 1) / (4 * NR);
    const int nb = blockIdx.x % nblk, ks = blockIdx.x / nblk;
    const int n0 = nb * 4 * NR + wave * NR;
    const long k0 = ks * kchunk;
    long k1 = k0 + kchunk; if (k1 > K) k1 = K;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    float acc[NR][MB];
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int m = 0; m < MB; ++m) acc[r][m] = 0.f;
    f32x4 xn[RW];
    auto load_x = [&](long k) {
#pragma unroll
        for (int j = 0; j < RW; ++j) { const int m = wave * RW + j; xn[j] = (m < M && k < k1) ? *(const f32x4*)(x + (size_t)m * K + k) : zero; }
    };
    auto store_x = [&](int buf) {
#pragma unroll
        for (int j = 0; j < RW; ++j) xs[buf][wave * RW + j][lane] = xn[j];
    };
    load_x(k0 + lane * 4);
    store_x(0);
    int cur = 0;
    for (long kb = k0; kb < k1; kb += 256, cur ^= 1) {      // uniform trip count for the whole block
        const long k = kb + lane * 4;
        f32x4 w[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r)
            w[r] = (n0 + r < N && k < k1) ? __builtin_nontemporal_load((const f32x4*)(W + (size_t)(n0 + r) * K + k)) : zero;
        const bool more = kb + 256 < k1;
        if (more) load_x(k + 256);
        __syncthreads();                                    // xs[cur] complete; everyone is done reading xs[cur ^ 1]
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            const f32x4 xv = xs[cur][m][lane];
#pragma unroll
            for (int r = 0; r < NR; ++r)
                acc[r][m] = fmaf(w[r].w, xv.w, fmaf(w[r].z, xv.z, fmaf(w[r].y, xv.y, fmaf(w[r].x, xv.x, acc[r][m]))));
        }
        if (more) store_x(cur ^ 1);
    }
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            const float s = wave_sum(acc[r][m]);
            if (lane == 0 && m < M && n0 + r < N) part[((size_t)ks * M + m) * N + n0 + r] = s;
        }
}
__global__ void linear_fwd_final_kernel(const float* __restrict__ part, const float* __restrict__ b, float* __restrict__ y, int M,
                                        int N, int ksplit, int act, float slope) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= M * N) return;
    float s = 0.f;
    for (int k = 0; k < ksplit; ++k) s += part[(size_t)k * M * N + e];
    if (b) s += b[e % N];
    if (act == PESR_ACT_LRELU) s = s > 0.f ? s : s * slope;
    else if (act == PESR_ACT_RELU) s = s > 0.f ? s : 0.f;
    y[e] = s;
}

// ---- dgrad -------------------------------------------------------------------------------------
// dx[m][k] = sum_n dy[m][n] W[n][k] in ONE launch, the weight matrix streamed once and nothing else of size written:
// a workgroup owns 256 consecutive k (one 16-byte piece per lane: every load instruction of a wave reads ONE KiB of ONE
// weight row) for ALL rows; its 8 waves split the N rows 8 ways (LD_U rows' loads in flight per wave), keep M x 4 sums per
// lane and meet in LDS at the end - waves 4..7 park their sums, waves 0..3 add their own, then the four partial rows are
// summed in wave order: a fixed order, bit-reproducible.  64 KiB of LDS, ~110 VGPRs: two workgroups share a CU, so the
// K / 256 = 288 workgroups of the 73728-column layer are all resident at once (no second round) with 72 KiB of loads in
// flight per CU.  Round 2's form (thread = 4 k, grid over 16 N-slices, M x K partials per slice + a finalize launch) moved
// 302 + 150 MB per call.  dy values are wave-uniform (scalar loads).
#ifndef LD_U
#define LD_U 8
#endif
template <int MB>
__global__ __launch_bounds__(512, (MB <= 16 ? 4 : 2)) void linear_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ W,
                                                              float* __restrict__ dx, int M, int N, long K) {
    extern __shared__ __attribute__((aligned(16))) float lds_red[];        // [4][MB][256] floats
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long k = (long)blockIdx.x * 256 + lane * 4;
    const bool kok = k < K;
    const long kc = kok ? k : 0;
    const int nchunk = (N + 7) / 8;
    const int n0 = wave * nchunk;
    int n1 = n0 + nchunk; if (n1 > N) n1 = N;
    f32x4 acc[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int U = LD_U;
    int n = n0;
    for (; n + U <= n1; n += U) {
        f32x4 w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) w[u] = __builtin_nontemporal_load((const f32x4*)(W + (size_t)(n + u) * K + kc));
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int m = 0; m < MB; ++m)
                if (m < M) acc[m] += w[u] * dy[(size_t)m * N + n + u];
    }
    for (; n < n1; ++n) {
        const f32x4 w = *(const f32x4*)(W + (size_t)n * K + kc);
#pragma unroll
        for (int m = 0; m < MB; ++m)
            if (m < M) acc[m] += w * dy[(size_t)m * N + n];
    }
    f32x4* red = (f32x4*)lds_red + (size_t)(wave & 3) * MB * 64 + lane;      // [wave & 3][m][lane]
    if (wave >= 4) {
#pragma unroll
        for (int m = 0; m < MB; ++m) red[m * 64] = acc[m];
    }
    __syncthreads();
    if (wave < 4) {
#pragma unroll
        for (int m = 0; m < MB; ++m) red[m * 64] = acc[m] + red[m * 64];     // wave w + wave w + 4
    }
    __syncthreads();
    // 512 threads finish MB x 64 pieces: ((p0 + p1) + p2) + p3
    const f32x4* all = (const f32x4*)lds_red;
    for (int e = threadIdx.x; e < MB * 64; e += 512) {
        const int m = e >> 6, l = e & 63;
        const long kk = (long)blockIdx.x * 256 + l * 4;
        if (m < M && kk < K) {
            const f32x4 s = ((all[e] + all[MB * 64 + e]) + all[2 * MB * 64 + e]) + all[3 * MB * 64 + e];
            *(f32x4*)(dx + (size_t)m * K + kk) = s;
        }
    }
}

// ---- wgrad -------------------------------------------------------------------------------------
// thread: 4 consecutive k with x[0..M)[k4] held in registers; loops over an N slice writing dW rows.
template <int MB>
__global__ __launch_bounds__(256) void linear_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           float* __restrict__ dW, int M, int N, long K, int nchunk, int accumulate) {
    const long k = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    const int n0 = blockIdx.y * nchunk;
    int n1 = n0 + nchunk; if (n1 > N) n1 = N;
    if (k >= K) return;
    f32x4 xv[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) xv[m] = m < M ? *(const f32x4*)(x + (size_t)m * K + k) : (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int n = n0; n < n1; ++n) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < MB; ++m)
            if (m < M) s += xv[m] * dy[(size_t)m * N + n];
        if (accumulate) s += __builtin_nontemporal_load((const f32x4*)(dW + (size_t)n * K + k));   // dW += ... : a second use of the layer in one backward
        __builtin_nontemporal_store(s, (f32x4*)(dW + (size_t)n * K + k));
    }
}
__global__ void linear_bgrad_kernel(const float* __restrict__ dy, float* __restrict__ db, int M, int N, int accumulate) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float s = 0.f;
    for (int m = 0; m < M; ++m) s += dy[(size_t)m * N + n];
    db[n] = accumulate ? db[n] + s : s;
}

namespace {
struct LinPlan { int ksplit; long kchunk; int nr; };
static void lin_plan(int M, int N, long K, LinPlan* p) {
    // forward: K slices in multiples of 256 (one KiB per lane-linear load); M <= 16: blocks of 32 features, ~512 of them (two
    // waves per SIMD on every CU); else waves = ceil(N / 2) * ksplit ~ 4096
    p->nr = M <= 16 ? 8 : 2;
    const int ngroups = (N + p->nr - 1) / p->nr;
    int ks = (M <= 16 ? 2048 : 4096) / ngroups; if (ks < 1) ks = 1;
    long kc = ((K + ks - 1) / ks + 255) / 256 * 256;
    if (kc < 256) kc = 256;
    p->kchunk = kc; p->ksplit = (int)((K + kc - 1) / kc);
}
}  // namespace

size_t pesr_linear_ws_bytes(int M, int N, long K) {
    LinPlan p; lin_plan(M, N, K, &p);
    return (size_t)p.ksplit * M * N * sizeof(float) + 256;
}

int pesr_linear_fwd_launch(const float* x, const float* W, const float* b, float* y, int M, int N, long K, int act, float slope,
                           void* ws, size_t ws_bytes, hipStream_t stream) {
    if (M > LIN_MAXM || M < 1 || K % 4) return PESR_EINVAL;
    LinPlan p; lin_plan(M, N, K, &p);
    if (!ws || ws_bytes < (size_t)p.ksplit * M * N * sizeof(float)) return PESR_EWORKSPACE;
    const long waves = (long)((N + p.nr - 1) / p.nr) * p.ksplit;
    if (M <= 16)
        hipLaunchKernelGGL(linear_fwd_lds_kernel<16>, dim3((unsigned)(((N + 31) / 32) * p.ksplit)), dim3(256), 0, stream, x, W, (float*)ws, M, N, K,
                           p.ksplit, p.kchunk);
    else
        hipLaunchKernelGGL((linear_fwd_kernel<32, 2>), dim3((int)((waves + 3) / 4)), dim3(256), 0, stream, x, W, (float*)ws, M, N, K, p.ksplit, p.kchunk);
    hipLaunchKernelGGL(linear_fwd_final_kernel, dim3((M * N + 255) / 256), dim3(256), 0, stream, (const float*)ws, b, y, M, N, p.ksplit, act, slope);
    return pesr_launch_status();
}

int pesr_linear_dgrad_launch(const float* dy, const float* W, float* dx, int M, int N, long K, void* ws, size_t ws_bytes,
                             hipStream_t stream) {
    if (M > LIN_MAXM || M < 1 || K % 4) return PESR_EINVAL;
    (void)ws; (void)ws_bytes;                       // (the one-launch kernel needs no workspace; kept in the ABI)
    const unsigned grid = (unsigned)((K + 255) / 256);
    if (M <= 16) {
        constexpr size_t lds = (size_t)4 * 16 * 256 * sizeof(float);
        hipLaunchKernelGGL(linear_dgrad_kernel<16>, dim3(grid), dim3(512), lds, stream, dy, W, dx, M, N, K);
    } else {
        constexpr size_t lds = (size_t)4 * 32 * 256 * sizeof(float);
        static std::once_flag attr_once;
        std::call_once(attr_once, [&] {
            (void)hipFuncSetAttribute((const void*)linear_dgrad_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        });
        hipLaunchKernelGGL(linear_dgrad_kernel<32>, dim3(grid), dim3(512), lds, stream, dy, W, dx, M, N, K);
    }
    return pesr_launch_status();
}

int pesr_linear_wgrad_launch(const float* dy, const float* x, float* dW, float* db, int M, int N, long K, int accumulate,
                             hipStream_t stream) {
    if (M > LIN_MAXM || M < 1 || K % 4) return PESR_EINVAL;
    int nchunk = 64; if (nchunk > N) nchunk = N;
    const dim3 grid((unsigned)((K / 4 + 255) / 256), (unsigned)((N + nchunk - 1) / nchunk));
    if (M <= 16) hipLaunchKernelGGL(linear_wgrad_kernel<16>, grid, dim3(256), 0, stream, dy, x, dW, M, N, K, nchunk, accumulate);
    else hipLaunchKernelGGL(linear_wgrad_kernel<32>, grid, dim3(256), 0, stream, dy, x, dW, M, N, K, nchunk, accumulate);
    if (db) hipLaunchKernelGGL(linear_bgrad_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, dy, db, M, N, accumulate);
    return pesr_launch_status();
}
