// Skinny-batch Linear for the Discriminator's classifier (reference model/pesr.py:69-74: Linear(73728, 1024)
// -> LeakyReLU(0.2) -> Linear(1024, 1); ATen addmm / mm in forward and backward).  M (batch) <= 32.
// All three passes are HBM-bound on the 302 MB weight matrix, which each streams exactly once:
//   fwd  : y[m][n]  = act(sum_k x[m][k] W[n][k] + b[n])     split-K partials (MFMA: the batch is one or two 16-row tiles) + fixed-order finalize
//   dgrad: dx[m][k] = sum_n dy[m][n] W[n][k]                 split-N partials + fixed-order finalize
//   wgrad: dW[n][k] = sum_m dy[m][n] x[m][k],  db[n] = sum_m dy[m][n]
#include <mutex>
#include "common.h"
#include "launchers.h"

#define LIN_MAXM 32

// ---- forward -----------------------------------------------------------------------------------
// M <= 16: the batch IS an MFMA dimension.  One wave owns NB x 16 output features and one K slice; per 16-k step every
// lane loads ONE 16-byte piece of x (row lane%16, k-slot lane/16) and NB pieces of W, and element e of the pieces feeds
// MFMA k-step e (k-slot g of step e stands for k = k0 + 4g + e).  The x slice is thus read once per NB*16 features
// (75 MB of L2 traffic for the 73728 -> 1024 layer instead of 600 MB with one feature row per lane group), and the
// 302 MB weight matrix streams exactly once.  D tile: row m = 4*(lane/16) + reg, col n = lane%16.
template <int NB>
__global__ __launch_bounds__(256) void linear_fwd_mfma_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                              float* __restrict__ part, int M, int N, long K, int ksplit, long kchunk) {
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const int i = lane & 15, g = lane >> 4;
    const int ngroups = (N + 16 * NB - 1) / (16 * NB);
    const int ng = wave % ngroups, ks = wave / ngroups;
    if (ks >= ksplit) return;
    const int n0 = ng * 16 * NB;
    const long k0 = ks * kchunk;
    long k1 = k0 + kchunk; if (k1 > K) k1 = K;
    f32x4 acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const float* xr = x + (size_t)(i < M ? i : 0) * K;
    const float* wr[NB];
    bool wok[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) { const int n = n0 + b * 16 + i; wok[b] = n < N; wr[b] = W + (size_t)(wok[b] ? n : 0) * K; }
    const bool xok = i < M;
    // (Round 6: x repacked k-major - a wave's x piece of a 16-k step as ONE contiguous KiB, [K/16][16][16], by a 5.9 us pack launch -
    // measured 84.6 us for this kernel against 83 - 85 on the plain layout: no gain; with NO x loads at all it takes ~72.  Not adopted.)
    // LU 16-k steps per trip, all their loads issued before the first MFMA (the two 64-byte halves of every 128-byte line of W
    // are then in flight together): LU * (NB + 1) KiB per wave
    constexpr int LU = 2;
    for (long k = k0 + 4 * g; k < k1 + 4 * g; k += 16 * LU) {   // uniform trip count; a lane's piece may lie past k1
        f32x4 a[LU], w[LU][NB];
#pragma unroll
        for (int u = 0; u < LU; ++u) {
            const long ku = k + 16 * u;
            const bool in = ku < k1;
            a[u] = (xok && in) ? *(const f32x4*)(xr + ku) : zero;
#pragma unroll
            for (int b = 0; b < NB; ++b) w[u][b] = (wok[b] && in) ? __builtin_nontemporal_load((const f32x4*)(wr[b] + ku)) : zero;
        }
#pragma unroll
        for (int u = 0; u < LU; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int b = 0; b < NB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][e], w[u][b][e], acc[b], 0, 0, 0);
    }
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int m = 4 * g + jj, n = n0 + b * 16 + i;
            if (m < M && n < N) part[((size_t)ks * M + m) * N + n] = acc[b][jj];
        }
}
// Round 6: the same product with W fetched in the pattern that streams fastest.  The kernel above takes W as 16 rows x 64 B per load
// instruction (the MFMA's B fragment: lane % 16 = row) - the fabric then sees a mix of 64- and 128-byte requests (profiles/
// r06_linear_tcc_summary.csv: 3.57 M requests for 2.36 M lines, none of 32 B) and the stream runs at 3.6 TB/s, while the input gradient's
// 4 rows x 256 B per instruction produce 128-byte requests only and reach 4.9.  Here a wave fetches its 64-row x 64-k tile of W as sixteen
// 4 x 256 B loads, passes it through a wave-private 16 KiB LDS image (16-byte chunks XOR-swizzled by the row: the writes fill whole rows,
// the 8 lanes of a fragment-read cycle hit 8 different chunks) and reads the B fragments from there; x as before.  No barrier: the
// image belongs to one wave, whose LDS operations execute in order.
template <int NB, int MT>   // MT: 16-row tiles of x (M <= 16 MT): two calls of a layer batched into one pass over W (round 6, the Discriminator's
__global__ __launch_bounds__(256) void linear_fwd_mfma_lds_kernel(const float* __restrict__ x, const float* __restrict__ W,   // classifier on [hr; sr])
                                                                  float* __restrict__ part, int M, int N, long K, int ksplit, long kchunk) {
    extern __shared__ __attribute__((aligned(16))) char lin_smem[];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + wv;
    const int i = lane & 15, g = lane >> 4;
    const int ngroups = (N + 16 * NB - 1) / (16 * NB);
    const int ng = (int)(wave % ngroups), ks = (int)(wave / ngroups);
    if (ks >= ksplit) return;
    const int n0 = ng * 16 * NB;
    const long k0 = ks * kchunk;
    long k1 = k0 + kchunk; if (k1 > K) k1 = K;
    char* const tile = lin_smem + wv * (16 * NB * 256);     // [16 NB rows][16 chunks of 16 B], chunk c of row r at slot c ^ (r & 15)
    f32x4 acc[MT][NB];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[t][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const float* xr[MT];
    bool xok[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) { xok[t] = 16 * t + i < M; xr[t] = x + (size_t)(xok[t] ? 16 * t + i : 0) * K; }
    const bool rows_in = n0 + 16 * NB <= N;
    for (long k = k0; k < k1; k += 64) {
        const bool all_in = rows_in && k + 64 <= k1;          // wave-uniform: the whole 64 x 64 tile and the x pieces exist
        f32x4 w[4 * NB], a[MT][4];
        if (all_in) {
#pragma unroll
            for (int j = 0; j < 4 * NB; ++j) w[j] = __builtin_nontemporal_load((const f32x4*)(W + (size_t)(n0 + 4 * j + g) * K + k + 4 * i));
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_) a[t][s_] = xok[t] ? *(const f32x4*)(xr[t] + k + 16 * s_ + 4 * g) : zero;
        } else {
#pragma unroll
            for (int j = 0; j < 4 * NB; ++j) {
                const int n = n0 + 4 * j + g;
                w[j] = (n < N && k + 4 * i < k1) ? *(const f32x4*)(W + (size_t)n * K + k + 4 * i) : zero;
            }
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_) a[t][s_] = (xok[t] && k + 16 * s_ + 4 * g < k1) ? *(const f32x4*)(xr[t] + k + 16 * s_ + 4 * g) : zero;
        }
#pragma unroll
        for (int j = 0; j < 4 * NB; ++j) {
            const int r = 4 * j + g;
            *(f32x4*)(tile + r * 256 + ((i ^ (r & 15)) << 4)) = w[j];
        }
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
            f32x4 bf[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int r = 16 * b + i;                       // (r & 15) == i
                bf[b] = *(const f32x4*)(tile + r * 256 + (((4 * s_ + g) ^ i) << 4));
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int t = 0; t < MT; ++t)
#pragma unroll
                    for (int b = 0; b < NB; ++b) acc[t][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][s_][e], bf[b][e], acc[t][b], 0, 0, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int m = 16 * t + 4 * g + jj, n = n0 + b * 16 + i;
                if (m < M && n < N) part[((size_t)ks * M + m) * N + n] = acc[t][b][jj];
            }
}
__global__ void linear_fwd_final_kernel(const float* __restrict__ part, const float* __restrict__ b, float* __restrict__ y, int M,
                                        int N, int ksplit, int act, float slope) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= M * N) return;
    // the partials are added in slice order (fixed), but loaded sixteen at a time: as a rolled loop its 128 dependent
    // load -> add steps took 32 us for 16 K outputs
    float s = 0.f;
    int k = 0;
    for (; k + 16 <= ksplit; k += 16) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = part[(size_t)(k + u) * M * N + e];
#pragma unroll
        for (int u = 0; u < 16; ++u) s += v[u];
    }
    for (; k < ksplit; ++k) s += part[(size_t)k * M * N + e];
    if (b) s += b[e % N];
    if (act == PESR_ACT_LRELU) s = s > 0.f ? s : s * slope;
    else if (act == PESR_ACT_RELU) s = s > 0.f ? s : 0.f;
    y[e] = s;
}

// ---- dgrad -------------------------------------------------------------------------------------
// thread: 4 consecutive k, all M rows; loops over an N slice; dy values are wave-uniform (scalar loads).
constexpr int LD_BT = 256;        // threads per block of the input-gradient kernel
constexpr int LD_U = 4;           // weight rows in flight per step (8: 134 / 124 us, 16: 146 / 149 against 111 / 101; profiles/r03_linear_sweep.txt)
template <int MB>
__global__ __launch_bounds__(LD_BT) void linear_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ W,
                                                             float* __restrict__ part, int M, int N, long K, int nchunk) {
    const long k = ((long)blockIdx.x * LD_BT + threadIdx.x) * 4;
    const int ns = blockIdx.y;
    const int n0 = ns * nchunk;
    int n1 = n0 + nchunk; if (n1 > N) n1 = N;
    if (k >= K) return;
    f32x4 acc[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // LD_U weight rows per step, all their 16-byte loads issued before the first FMA.  Round-3 sweep on one box (back-to-back /
    // after a 512 MB flush): LD_U 4: 111 / 101 us, 8: 134 / 124, 16: 146 / 149 (round 2 had measured 8 as 106 on its box); 64- /
    // 512- / 1024-thread blocks slower.  A one-launch form (a workgroup owns 256 k for ALL rows, its waves split N and meet in
    // LDS: no partial slabs, no finalize launch, 302 instead of 457 MB moved) measured 131 - 234 us in four variants: with 4.5 - 9
    // waves per CU it keeps too few loads in flight; this split-N form runs 16 waves per CU.
    constexpr int U = LD_U;
    int n = n0;
    for (; n + U <= n1; n += U) {
        f32x4 w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) w[u] = __builtin_nontemporal_load((const f32x4*)(W + (size_t)(n + u) * K + k));
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int m = 0; m < MB; ++m)
                if (m < M) acc[m] += w[u] * dy[(size_t)m * N + n + u];
    }
    for (; n < n1; ++n) {
        const f32x4 w = *(const f32x4*)(W + (size_t)n * K + k);
#pragma unroll
        for (int m = 0; m < MB; ++m)
            if (m < M) acc[m] += w * dy[(size_t)m * N + n];
    }
#pragma unroll
    for (int m = 0; m < MB; ++m)
        if (m < M) *(f32x4*)(part + ((size_t)ns * M + m) * K + k) = acc[m];
}
// M <= 16 (round 6): the input gradient on the matrix pipe.  One wave owns 64 consecutive k and an N slice.  Per 16 weight rows: every lane
// loads ONE 16-byte piece of dy - row i = lane % 16, columns n + 4 g .. + 3 (g = lane / 16) - and FOUR pieces of W - row n + 4 g + t
// (t = 0 .. 3), columns k0 + 4 i .. + 3: 16 lanes x 16 B = 256 contiguous bytes of each of four rows per load instruction.  MFMA (t, e):
// A = dy piece element t (k-slot g stands for weight row n + 4 g + t), B = element e of W piece t (column k0 + 4 i + e), accumulator e:
// D_e[m][i] = dx[m][k0 + 4 i + e].  A lane ends with dx[m = 4 g + jj][k0 + 4 i .. + 3] = {acc[0][jj] .. acc[3][jj]}: one 16-byte store
// per jj.  Against the VALU form above: no 64 FMAs per 16 bytes of W (the kernel was bound by them: 105 - 110 us = 0.45 of the HBM rate,
// W alone streams in ~60), and an N split of 4 instead of 16 (19 MB of partial slabs instead of 75).
template <int MT>             // 16-row tiles of dy (M <= 16 MT)
__global__ __launch_bounds__(256) void linear_dgrad_mfma_kernel(const float* __restrict__ dy, const float* __restrict__ W,
                                                                float* __restrict__ part, int M, int N, long K, int nchunk, long ktiles) {
    const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
    const long kt = wave % ktiles;
    const int ns = (int)(wave / ktiles);
    const int n0 = ns * nchunk;
    if (n0 >= N) return;
    int n1 = n0 + nchunk; if (n1 > N) n1 = N;
    const long k0 = kt * 64;
    const long kl = k0 + 4 * i;                              // this lane's four columns
    const bool kok = kl < K;                                 // (K % 4 == 0)
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[mt][e] = zero;
    const float* dyr[MT];
    bool mok[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) { mok[mt] = 16 * mt + i < M; dyr[mt] = dy + (size_t)(mok[mt] ? 16 * mt + i : 0) * N; }
    constexpr int U = 2;                                     // 16-row groups per trip: 8 KiB of W in flight per wave
    int n = n0;
    for (; n + 16 * U <= n1; n += 16 * U) {
        f32x4 d[U][MT], w[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int nb = n + 16 * u + 4 * g;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) d[u][mt] = mok[mt] ? *(const f32x4*)(dyr[mt] + nb) : zero;
#pragma unroll
            for (int t = 0; t < 4; ++t) w[u][t] = kok ? __builtin_nontemporal_load((const f32x4*)(W + (size_t)(nb + t) * K + kl)) : zero;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[mt][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(d[u][mt][t], w[u][t][e], acc[mt][e], 0, 0, 0);
    }
    for (; n < n1; n += 16) {                                // ragged end of the slice: rows past n1 contribute zeros
        const int nb = n + 4 * g;
        f32x4 d[MT], w[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bool rok = nb + t < n1;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) d[mt][t] = (mok[mt] && rok) ? dyr[mt][nb + t] : 0.f;
            w[t] = (kok && rok) ? *(const f32x4*)(W + (size_t)(nb + t) * K + kl) : zero;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(d[mt][t], w[t][e], acc[mt][e], 0, 0, 0);
    }
    if (kok) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int m = 16 * mt + 4 * g + jj;
                if (m < M) *(f32x4*)(part + ((size_t)ns * M + m) * K + kl) = (f32x4){acc[mt][0][jj], acc[mt][1][jj], acc[mt][2][jj], acc[mt][3][jj]};
            }
    }
}
// finalize: dx = sum over N-slices; the LeakyReLU derivative of the layer below is applied by its own backward
__global__ void linear_dgrad_final_kernel(const f32x4* __restrict__ part, f32x4* __restrict__ dx, long MK4, int nsplit) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < MK4; e += (long)gridDim.x * blockDim.x) {
        f32x4 s = part[e];
        for (int k = 1; k < nsplit; ++k) s += part[(size_t)k * MK4 + e];
        dx[e] = s;
    }
}

// ---- wgrad -------------------------------------------------------------------------------------
// thread: 4 consecutive k with x[0..M)[k4] held in registers; loops over an N slice writing dW rows.
template <int MB>
__global__ __launch_bounds__(256) void linear_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           float* __restrict__ dW, int M, int N, long K, int nchunk, int accumulate) {
    // Round 6: the block's dy tile [M][nchunk <= 64] goes to LDS once (transposed to [n][MB]: a row of it is the MB multipliers of one
    // output row, read as broadcast ds_read_b128s) - as wave-uniform global reads inside the n loop every output row waited for 16
    // dependent scalar loads before its store could issue.
    __shared__ __attribute__((aligned(16))) float dyt[64 * MB];
    const long k = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    const int n0 = blockIdx.y * nchunk;
    int n1 = n0 + nchunk; if (n1 > N) n1 = N;
    float dv[64 * MB / 256];                               // all of a thread's dy values are in flight before the first is stored
#pragma unroll
    for (int j = 0; j < 64 * MB / 256; ++j) {
        const int e = threadIdx.x + j * 256, nn = e / MB, m = e - nn * MB;
        dv[j] = (m < M && n0 + nn < n1) ? dy[(size_t)m * N + n0 + nn] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 64 * MB / 256; ++j) dyt[threadIdx.x + j * 256] = dv[j];
    __syncthreads();
    if (k >= K) return;
    f32x4 xv[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) xv[m] = m < M ? *(const f32x4*)(x + (size_t)m * K + k) : (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int n = n0; n < n1; ++n) {
        const f32x4* const d4 = (const f32x4*)(dyt + (n - n0) * MB);
        // Rows are added in groups of sixteen, each group a chain from zero, the groups then to each other: the result for 32 rows is bit
        // for bit what two calls of 16 rows (the second with `accumulate`) leave - the Discriminator's classifier on [hr; sr] in one pass
        // gives the gradients its two calls gave.  (Rows m >= M hold zeros in both operands.)
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int h = 0; h < MB / 16; ++h) {
            f32x4 sh = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 4 * h; q < 4 * h + 4; ++q) {
                const f32x4 d = d4[q];
                sh += xv[4 * q] * d.x; sh += xv[4 * q + 1] * d.y; sh += xv[4 * q + 2] * d.z; sh += xv[4 * q + 3] * d.w;
            }
            s = h == 0 ? sh : sh + s;             // (the second call computed its chain and added what was there)
        }
        // (non-temporal: the 302 MB stream is written once and not re-read by this kernel - 88 -> 78 us)
        if (accumulate) s += __builtin_nontemporal_load((const f32x4*)(dW + (size_t)n * K + k));   // dW += ... : a second use of the layer in one backward
        __builtin_nontemporal_store(s, (f32x4*)(dW + (size_t)n * K + k));
    }
}
__global__ void linear_bgrad_kernel(const float* __restrict__ dy, float* __restrict__ db, int M, int N, int accumulate) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float s = 0.f;                                          // groups of sixteen rows, as in the weight gradient above: under RSGAN the two
    for (int m0 = 0; m0 < M; m0 += 16) {                     // halves of [hr; sr] are exact negatives of each other and classifier.2.bias's
        float sh = 0.f;                                      // gradient is exactly zero, as in the reference
        for (int m = m0; m < M && m < m0 + 16; ++m) sh += dy[(size_t)m * N + n];
        s = m0 == 0 ? sh : s + sh;
    }
    db[n] = accumulate ? db[n] + s : s;
}

namespace {
struct LinPlan { int ksplit; long kchunk; int nsplit, nchunk; };
static void lin_plan(int M, int N, long K, LinPlan* p) {
    // forward, MFMA kernels (one or two 16-row tiles of x; a tiny N with M > 16 is launched as two 16-row calls): waves = ceil(N/64) * ksplit
    // ~ 2048 (more waves or deeper unrolling measured slower), K slices in multiples of 64
    if (M <= LIN_MAXM) {
        const int ngroups = (N + 63) / 64;
        int ks = 2048 / ngroups; if (ks < 1) ks = 1;
        long kc = ((K + ks - 1) / ks + 63) / 64 * 64;         // (multiples of 64: a trip of the LDS-staged kernel)
        if (kc < 64) kc = 64;
        p->kchunk = kc; p->ksplit = (int)((K + kc - 1) / kc);
    }
    // dgrad: one-wave blocks, ceil(K/256) * nsplit ~ 1024 of them (every N slice writes an M x K partial: keep them few)
    const long kb = (K + 4 * LD_BT - 1) / (4 * LD_BT);
    constexpr int LD_NS_TARGET = 1024, LD_NS_MAX = 16;    // (the 16 x 4 point of the sweep in profiles/r03_linear_sweep.txt)
    int ns = (int)((LD_NS_TARGET * 256 / LD_BT + kb - 1) / kb); if (ns < 1) ns = 1; if (ns > N) ns = N; if (ns > LD_NS_MAX) ns = LD_NS_MAX;
    p->nchunk = (N + ns - 1) / ns; p->nsplit = (N + p->nchunk - 1) / p->nchunk;
    if (M <= LIN_MAXM) {
        // MFMA input gradient: waves = ceil(K / 64) * nsplit ~ 4608 (18 per CU), slices in multiples of 32 rows (the loop's trip)
        const long ktiles = (K + 63) / 64;
        int ns2 = (int)((4608 + ktiles - 1) / ktiles); if (ns2 < 1) ns2 = 1; if (ns2 > 16) ns2 = 16;
        int nc = ((N + ns2 - 1) / ns2 + 31) / 32 * 32; if (nc < 32) nc = 32;
        p->nchunk = nc; p->nsplit = (N + nc - 1) / nc;
    }
}
}  // namespace

size_t pesr_linear_ws_bytes(int M, int N, long K) {
    LinPlan p; lin_plan(M, N, K, &p);
    const size_t a = (size_t)p.ksplit * M * N * sizeof(float);
    const size_t b = (size_t)p.nsplit * M * K * sizeof(float);
    return a > b ? a : b;
}

int pesr_linear_fwd_launch(const float* x, const float* W, const float* b, float* y, int M, int N, long K, int act, float slope,
                           void* ws, size_t ws_bytes, hipStream_t stream) {
    if (M > LIN_MAXM || M < 1 || K % 4) return PESR_EINVAL;
    LinPlan p; lin_plan(M, N, K, &p);
    if (!ws || ws_bytes < (size_t)p.ksplit * M * N * sizeof(float)) return PESR_EWORKSPACE;
    {
        const long waves = (long)((N + 63) / 64) * p.ksplit;
        const dim3 grid((unsigned)((waves + 3) / 4));
        if (N >= 64) {   // W through the wave-private LDS image (whole 128-byte lines per fetch); tiny N: the direct form
            static PesrDeviceOnce attr_once;
            attr_once([&] {
                (void)hipFuncSetAttribute((const void*)linear_fwd_mfma_lds_kernel<4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
                (void)hipFuncSetAttribute((const void*)linear_fwd_mfma_lds_kernel<4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
            });
            if (M <= 16) hipLaunchKernelGGL((linear_fwd_mfma_lds_kernel<4, 1>), grid, dim3(256), 64 * 1024, stream, x, W, (float*)ws, M, N, K, p.ksplit, p.kchunk);
            else hipLaunchKernelGGL((linear_fwd_mfma_lds_kernel<4, 2>), grid, dim3(256), 64 * 1024, stream, x, W, (float*)ws, M, N, K, p.ksplit, p.kchunk);
        } else if (M <= 16) {
            hipLaunchKernelGGL(linear_fwd_mfma_kernel<4>, grid, dim3(256), 0, stream, x, W, (float*)ws, M, N, K, p.ksplit, p.kchunk);
        } else {         // tiny N (classifier.2): the 16-row kernel twice - per row the sums its single calls make
            int rc = pesr_linear_fwd_launch(x, W, b, y, 16, N, K, act, slope, ws, ws_bytes, stream);
            if (rc) return rc;
            return pesr_linear_fwd_launch(x + (size_t)16 * K, W, b, y + (size_t)16 * N, M - 16, N, K, act, slope, ws, ws_bytes, stream);
        }
    }
    hipLaunchKernelGGL(linear_fwd_final_kernel, dim3((M * N + 63) / 64), dim3(64), 0, stream, (const float*)ws, b, y, M, N, p.ksplit, act, slope);
    return pesr_launch_status();
}

int pesr_linear_dgrad_launch(const float* dy, const float* W, float* dx, int M, int N, long K, void* ws, size_t ws_bytes,
                             hipStream_t stream) {
    if (M > LIN_MAXM || M < 1 || K % 4) return PESR_EINVAL;
    LinPlan p; lin_plan(M, N, K, &p);
    if (!ws || ws_bytes < (size_t)p.nsplit * M * K * sizeof(float)) return PESR_EWORKSPACE;
    const dim3 grid((unsigned)((K / 4 + LD_BT - 1) / LD_BT), (unsigned)p.nsplit);
    if (N % 4 == 0) {
        const long ktiles = (K + 63) / 64, waves = ktiles * p.nsplit;
        if (M <= 16) hipLaunchKernelGGL(linear_dgrad_mfma_kernel<1>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, stream, dy, W, (float*)ws, M, N, K, p.nchunk, ktiles);
        else hipLaunchKernelGGL(linear_dgrad_mfma_kernel<2>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, stream, dy, W, (float*)ws, M, N, K, p.nchunk, ktiles);
    } else if (M <= 16) hipLaunchKernelGGL(linear_dgrad_kernel<16>, grid, dim3(LD_BT), 0, stream, dy, W, (float*)ws, M, N, K, p.nchunk);
    else hipLaunchKernelGGL(linear_dgrad_kernel<32>, grid, dim3(LD_BT), 0, stream, dy, W, (float*)ws, M, N, K, p.nchunk);
    const long MK4 = (long)M * K / 4;
    hipLaunchKernelGGL(linear_dgrad_final_kernel, dim3((unsigned)((MK4 + 255) / 256 < 4096 ? (MK4 + 255) / 256 : 4096)), dim3(256), 0, stream,
                       (const f32x4*)ws, (f32x4*)dx, MK4, p.nsplit);
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
