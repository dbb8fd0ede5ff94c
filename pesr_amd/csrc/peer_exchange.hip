// Gradient exchange that holds no compute unit while it waits or moves bytes (round 5): the replacement of nn.DataParallel's
// reduce_add (reference train.py:114-118) as a reduce-scatter + all-gather over PEER MEMORY.  Every rank maps the other ranks'
// flat gradient buffers and flag words through IPC handles; an all-reduce of a slice is then, enqueued on ONE stream of the rank:
//     tell every peer "my gradients of this epoch are final"          hipStreamWriteValue32 into the peers' flag words
//     for every peer (one side stream each): wait for its word, copy   hipStreamWaitValue32 (the command processor polls: no wave is
//       MY 1/N of its buffer                                           resident), hipMemcpyAsync out of the mapping (copy engines
//                                                                      between devices)
//     sum the N copies of my 1/N in rank order                         peer_reduce_kernel: the one kernel, over 1/N of the bytes
//     tell every peer "my 1/N is reduced"; for every peer: wait, copy ITS reduced 1/N into my buffer
//     tell every peer "I have read your buffer"; wait for the same from everybody (their reads of MY buffer are over: it may be
//     written again)
// The sum of a slice is formed by exactly one rank, in the fixed order rank 0 .. N-1, and copied: replicas hold bit-identical
// gradients.  RCCL's all-reduce keeps workgroups resident for the whole transfer, and ONE held compute unit costs every
// 256-workgroup conv kernel a second round (x 1.82, profiles/r03_cu_contention.txt).  The flag words count epochs upwards (waits are
// ">="), so nothing is ever reset.  Primitives verified on this runtime with two processes on one GPU: scripts/ipc_probe.py.
#include <cstring>
#include "common.h"

namespace {
constexpr int PEER_MAX = 16;

struct PeerArgs {           // mirrors PesrPeerArgs of include/pesr_hip.h
    int rank, world;
    unsigned epoch;
    unsigned pad_;
    float* mine;                      // my tensor (a slice of the flat gradient buffer), numel floats
    float* peer[PEER_MAX];            // the same slice in every rank's buffer as mapped HERE (peer[rank] == mine)
    unsigned* my_flags;               // [3][PEER_MAX] words the peers write: READY, REDUCED, DONE
    unsigned* peer_flags[PEER_MAX];   // every rank's flag block as mapped here
    float* scratch;                   // (world - 1) x slice floats, this rank's own memory
    size_t numel;
    void* ctx;                        // pesr_peer_ctx_create(): one side stream per peer (the copies of a phase run on all links at once)
};

struct PeerCtx {
    hipStream_t side[PEER_MAX];
    hipEvent_t fork, join[PEER_MAX];
    hipEvent_t release;               // hipEventReleaseToSystem: recorded in front of every flag the PEERS' copies wait for (round 6)
    int n;
};

// occupies every wave slot of the GPU for `ticks` of the 100 MHz wall clock (pesr_peer_copy_probe): a copy that needs a compute unit
// (a blit kernel) cannot start before it ends, a copy engine does not care
__global__ __launch_bounds__(1024) void peer_hog_kernel(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}

__global__ __launch_bounds__(256) void peer_reduce_kernel(float* __restrict__ mine, const float* __restrict__ scratch, int world, int rank,
                                                          size_t n4, size_t slice4) {
    // out = ((x_0 + x_1) + x_2) + ... in rank order; x_rank is my own slice, x_q (q != rank) the copy in scratch slot q - (q > rank)
    f32x4* const m = (f32x4*)mine;
    const f32x4* const s = (const f32x4*)scratch;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        f32x4 acc = rank == 0 ? m[i] : s[i];
        for (int q = 1; q < world; ++q) acc += q == rank ? m[i] : s[(size_t)(q - (q > rank ? 1 : 0)) * slice4 + i];
        m[i] = acc;
    }
    // The sums are read next by the PEERS' copy engines, over xGMI, out of this GPU's memory: they must have left the L2 by the
    // time the kernel is complete (two ranks time-sharing one GPU - the only place this protocol has run - share the caches and
    // cannot show a missing write-back).  The host side also records an event (system-scope release) in front of the flag.
    __threadfence_system();
}
}  // namespace

PESR_API int pesr_peer_alloc(size_t bytes, void** ptr, unsigned char* handle64) {
    if (!ptr || !handle64 || !bytes) return PESR_EINVAL;
    hipError_t e = hipMalloc(ptr, bytes);
    if (e != hipSuccess) return (int)e;
    e = hipMemset(*ptr, 0, bytes);
    if (e != hipSuccess) return (int)e;
    // hipMemset of device memory returns before the zeroes are written (it is only ORDERED on the null stream, which the peers'
    // streams know nothing about): a peer that opens the handle and writes its first flag word could be overtaken by them - the
    // word is 0 again and this rank waits for ever.  Seen with four ranks on one busy GPU (the transport created in the middle of
    // a calibration: all four stuck in the constructor's self-test), never with an idle one.
    e = hipDeviceSynchronize();
    if (e != hipSuccess) return (int)e;
    hipIpcMemHandle_t h;
    e = hipIpcGetMemHandle(&h, *ptr);
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }
    std::memcpy(handle64, &h, 64);
    return PESR_OK;
}

PESR_API int pesr_peer_free(void* ptr) { return (int)hipFree(ptr); }

// Abandon an exchange that does not complete: every wait of THIS rank's streams is a ">= epoch" on a word of its own flag block, so
// filling the block with 0xffffffff (on a stream of its own: the stuck ones cannot be used) lets all of them run out.  The copies
// then move whatever the peers' buffers hold - the caller discards the result and the transport.
PESR_API int pesr_peer_release(void* my_flags, size_t bytes) {
    if (!my_flags || !bytes) return PESR_EINVAL;
    hipStream_t s;
    hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }
    e = hipMemsetAsync(my_flags, 0xff, bytes, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipStreamDestroy(s);
    if (e != hipSuccess) (void)hipGetLastError();
    return (int)e;
}

// IPC handle of the ALLOCATION that contains ptr (the caching allocator hands out pieces of its blocks) and ptr's offset in it.
PESR_API int pesr_peer_export(const void* ptr, unsigned char* handle64, size_t* offset, size_t* alloc_bytes) {
    if (!ptr || !handle64 || !offset) return PESR_EINVAL;
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    hipError_t e = hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)ptr);
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }
    hipIpcMemHandle_t h;
    e = hipIpcGetMemHandle(&h, (void*)base);
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }
    std::memcpy(handle64, &h, 64);
    *offset = (size_t)((const char*)ptr - (const char*)base);
    if (alloc_bytes) *alloc_bytes = size;
    return PESR_OK;
}

PESR_API int pesr_peer_open(const unsigned char* handle64, void** base) {
    if (!handle64 || !base) return PESR_EINVAL;
    hipIpcMemHandle_t h;
    std::memcpy(&h, handle64, 64);
    const hipError_t e = hipIpcOpenMemHandle(base, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) (void)hipGetLastError();
    return (int)e;
}

PESR_API int pesr_peer_close(void* base) { return (int)hipIpcCloseMemHandle(base); }

PESR_API int pesr_peer_ctx_create(int world, void** ctx) {
    if (!ctx || world < 1 || world > PEER_MAX) return PESR_EINVAL;
    PeerCtx* c = new PeerCtx();
    c->n = world - 1;
    hipError_t e = hipEventCreateWithFlags(&c->fork, hipEventDisableTiming);
    // HIP events release at DEVICE scope by default; the stores in front of a flag that another GPU's copy engine waits for must be
    // visible at SYSTEM scope (the default was relied on and had only ever run with both ranks behind one L2)
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->release, hipEventDisableTiming | hipEventReleaseToSystem);
    for (int i = 0; i < c->n && e == hipSuccess; ++i) {
        e = hipStreamCreateWithFlags(&c->side[i], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->join[i], hipEventDisableTiming);
    }
    if (e != hipSuccess) { (void)hipGetLastError(); delete c; return (int)e; }
    *ctx = c;
    return PESR_OK;
}

PESR_API int pesr_peer_ctx_destroy(void* ctx) {
    PeerCtx* c = (PeerCtx*)ctx;
    if (!c) return PESR_OK;
    (void)hipEventDestroy(c->fork);
    (void)hipEventDestroy(c->release);
    for (int i = 0; i < c->n; ++i) { (void)hipStreamDestroy(c->side[i]); (void)hipEventDestroy(c->join[i]); }
    delete c;
    return PESR_OK;
}

PESR_API int pesr_peer_allreduce(const void* args, void* stream_) {
    const PeerArgs& a = *(const PeerArgs*)args;
    hipStream_t stream = (hipStream_t)stream_;
    if (a.world < 1 || a.world > PEER_MAX || a.rank < 0 || a.rank >= a.world || (a.numel & 3) || !a.mine || !a.my_flags) return PESR_EINVAL;
    if (a.world == 1) return PESR_OK;
    const size_t slice = ((a.numel + a.world - 1) / a.world + 3) & ~(size_t)3;          // floats per rank, a multiple of 4
    auto lo = [&](int r) { const size_t v = (size_t)r * slice; return v < a.numel ? v : a.numel; };
    auto len = [&](int r) { return lo(r + 1) - lo(r); };
    enum { READY = 0, REDUCED = 1, DONE = 2 };
    hipError_t e = hipSuccess;
#define PEER_CK(X) do { e = (X); if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; } } while (0)
    auto tell = [&](int kind) -> hipError_t {
        for (int d = 1; d < a.world; ++d) {
            const int p = (a.rank + d) % a.world;
            const hipError_t r = hipStreamWriteValue32(stream, a.peer_flags[p] + kind * PEER_MAX + a.rank, a.epoch, 0);
            if (r != hipSuccess) return r;
        }
        return hipSuccess;
    };
    auto await = [&](int kind, int p) { return hipStreamWaitValue32(stream, a.my_flags + kind * PEER_MAX + p, a.epoch, hipStreamWaitValueGte, 0xffffffffu); };
    // A phase's waits and copies run on one side stream per peer - all xGMI links at once -, forked from and joined back into
    // `stream` by events (a wait for one slow peer does not hold up the copies from the others).
    PeerCtx* const c = (PeerCtx*)a.ctx;
    if (!c || c->n != a.world - 1) return PESR_EINVAL;
    auto phase = [&](int kind, bool gather) -> hipError_t {
        hipError_t r = hipEventRecord(c->fork, stream);
        for (int d = 1; d < a.world && r == hipSuccess; ++d) {
            const int p = (a.rank + d) % a.world;
            hipStream_t s = c->side[d - 1];
            r = hipStreamWaitEvent(s, c->fork, 0);
            if (r == hipSuccess) r = hipStreamWaitValue32(s, a.my_flags + kind * PEER_MAX + p, a.epoch, hipStreamWaitValueGte, 0xffffffffu);
            if (r == hipSuccess) {
                if (!gather) {       // my slice of peer p's buffer -> scratch
                    if (len(a.rank)) r = hipMemcpyAsync(a.scratch + (size_t)(p - (p > a.rank ? 1 : 0)) * slice, a.peer[p] + lo(a.rank),
                                                        len(a.rank) * sizeof(float), hipMemcpyDefault, s);
                } else if (len(p)) { // peer p's reduced slice -> my buffer
                    r = hipMemcpyAsync(a.mine + lo(p), a.peer[p] + lo(p), len(p) * sizeof(float), hipMemcpyDefault, s);
                }
            }
            if (r == hipSuccess) r = hipEventRecord(c->join[d - 1], s);
            if (r == hipSuccess) r = hipStreamWaitEvent(stream, c->join[d - 1], 0);
        }
        return r;
    };
    // 1. reduce-scatter: my slice of every peer's buffer -> scratch, then the one kernel
    //    (`stream` has waited for the compute stream's event: the system-scope release event makes the backward kernels' stores
    //    to the flat gradient buffer visible to the peers' copy engines before READY is)
    PEER_CK(hipEventRecord(c->release, stream));
    PEER_CK(tell(READY));
    PEER_CK(phase(READY, false));
    const size_t my_n = len(a.rank);
    if (my_n) {
        const size_t n4 = my_n >> 2;
        const unsigned grid = (unsigned)((n4 + 255) / 256 < 1024 ? (n4 + 255) / 256 : 1024);
        hipLaunchKernelGGL(peer_reduce_kernel, dim3(grid), dim3(256), 0, stream, a.mine + lo(a.rank), a.scratch, a.world, a.rank, n4, slice >> 2);
        const int rc = pesr_launch_status();
        if (rc) return rc;
    }
    // 2. all-gather: every peer's reduced slice -> my buffer (a peer has read MY copy of its slice before it says REDUCED).
    //    (system-scope release of the reduce kernel's stores in front of the flag the peers' copies wait for)
    PEER_CK(hipEventRecord(c->release, stream));
    PEER_CK(tell(REDUCED));
    PEER_CK(phase(REDUCED, true));
    // 3. nobody reads my buffer any more once every peer says DONE: only then may the stream's next work write it
    PEER_CK(tell(DONE));
    for (int d = 1; d < a.world; ++d) PEER_CK(await(DONE, (a.rank + d) % a.world));
#undef PEER_CK
    return PESR_OK;
}

// Which engine moves a peer copy?  (round 6)  Times hipMemcpyAsync(dst <- src, bytes) on `copy_stream`
// three ways, HIP events on that stream: alone; started right after a kernel that occupies every wave slot of this GPU for hog_us
// microseconds was launched on a second stream; and the hog kernel itself.  A copy executed by a blit KERNEL cannot start before the hog ends
// (out_ms[1] ~ hog + alone); one executed by a COPY ENGINE (SDMA) is not held up (out_ms[1] ~ out_ms[0]).
// out_ms[3] = {copy alone, copy under the hog, hog}.  Synchronises; a measurement helper, not part of the step.
PESR_API int pesr_peer_copy_probe(const void* src, void* dst, size_t bytes, int hog_us, float* out_ms) {
    if (!src || !dst || !bytes || !out_ms || hog_us < 100 || hog_us > 50000) return PESR_EINVAL;
    hipStream_t sc = nullptr, sh = nullptr;
    hipEvent_t e[6] = {};
    hipError_t r = hipStreamCreateWithFlags(&sc, hipStreamNonBlocking);
    if (r == hipSuccess) r = hipStreamCreateWithFlags(&sh, hipStreamNonBlocking);
    for (int i = 0; i < 6 && r == hipSuccess; ++i) r = hipEventCreate(&e[i]);
    int dev = 0, cus = 256;
    if (r == hipSuccess) r = hipGetDevice(&dev);
    if (r == hipSuccess) r = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (r == hipSuccess) r = hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, sc);           // warm-up (mappings, first-use setup)
    if (r == hipSuccess) r = hipStreamSynchronize(sc);
    if (r == hipSuccess) r = hipEventRecord(e[0], sc);
    if (r == hipSuccess) r = hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, sc);
    if (r == hipSuccess) r = hipEventRecord(e[1], sc);
    if (r == hipSuccess) r = hipStreamSynchronize(sc);
    if (r == hipSuccess) {
        r = hipEventRecord(e[4], sh);
        hipLaunchKernelGGL(peer_hog_kernel, dim3(2 * cus), dim3(1024), 0, sh, (long long)hog_us * 100);      // 2 x 16 waves per CU = every slot
        if (r == hipSuccess) r = hipGetLastError();
        if (r == hipSuccess) r = hipEventRecord(e[5], sh);
        if (r == hipSuccess) r = hipEventRecord(e[2], sc);
        if (r == hipSuccess) r = hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, sc);
        if (r == hipSuccess) r = hipEventRecord(e[3], sc);
    }
    if (r == hipSuccess) r = hipStreamSynchronize(sc);
    if (r == hipSuccess) r = hipStreamSynchronize(sh);
    for (int k = 0; k < 3 && r == hipSuccess; ++k) r = hipEventElapsedTime(&out_ms[k], e[2 * k], e[2 * k + 1]);
    for (int i = 0; i < 6; ++i) if (e[i]) (void)hipEventDestroy(e[i]);
    if (sc) (void)hipStreamDestroy(sc);
    if (sh) (void)hipStreamDestroy(sh);
    if (r != hipSuccess) (void)hipGetLastError();
    return (int)r;
}
