// kurbm_peer.hip -- kurbm_peer_*: the data-parallel exchange of the CD update WITHOUT a collective library -- a two-shot
// all-reduce through peer pointers (hipIpc), its second shot fused into the launch that applies the update.
//
// Reference: none -- ku/ebm has no multi-device path.  The exchange follows from rbm.py:125-134 being SUMS over
// the batch: any exact sum over the ranks, added in a fixed order, is legal.  RCCL (kurbm_comm.hip) stays the
// default; this is the plan B for BASELINE.json config 3, where a 3.2 MB ncclAllReduce costs more than the ~35 us the 6x scaling
// target leaves: xGMI is point to point, so with every rank reading 1/N of every peer's buffer all 7 links
// of a GPU carry 0.4 MB at the same time (~2.6 us of wire) where one ring moves 2 x 7/8 x 3.2 MB over one link (~37 us).
//
// Every rank owns ONE exchange buffer, allocated here, exported as a hipIpc handle and mapped by every other rank:
//     [ flags: ready | summed | counter ]  [ delta: the rank's own packed sums ]  [ sum: the rank's band of the total ]
// One exchange, epoch e (a counter that only grows; all ranks call in lock step):
//   shot 1 (k_peer_sum_band, after the launch that wrote `delta`): publish ready = e; wait for every rank's ready >= e; for the
//          rank's own band of the packed buffer -- rows [r band, (r + 1) band) of dW -- and, on EVERY rank, the bias tail: add
//          the N deltas IN RANK ORDER (identical bits everywhere), store to the own `sum`; the last workgroup publishes summed = e.
//   shot 2 (k_reduce_apply_split with a PeerSrc: kurbm_bf16.hip): a tile of W waits for the summed flag of the band its rows
//          belong to and reads them straight from that rank's `sum` -- the all-gather IS the apply's read; W += lr * sum, both
//          mirrors rewritten, as in the single-GPU reduce launch.  (kurbm_peer_allreduce_sum_f32: a plain gather instead.)
// Flags are system-scope atomics on fine-grained memory; every wait is bounded (KURBM_PEER_TIMEOUT_MS, default 10 s) and a
// timeout sets bit 1 of the context's status word (kurbm_ctx_status) and skips the work instead of hanging the GPU.
// Reuse is safe without double buffers: a rank rewrites `delta` only after its shot 2 has seen EVERY band's summed flag (all
// peers are done reading deltas), and `sum` only after every peer's ready flag of the NEXT epoch (all peers are done applying).
//
// Testable on ONE GPU: two processes on device 0 exchange handles and run the same protocol (tests/test_peer_exchange.py) --
// RCCL refuses two ranks on one device, plain IPC does not.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "../../include/kurbm.h"
#include "kurbm_kernels.h"
#include "kurbm_comm.h"
#include "kurbm_device.h"

using kurbm::fail_msg;

struct kurbm_peer {
    int device = 0, nranks = 1, rank = 0;
    int n_vis = 0, n_hid = 0;
    size_t n = 0;                     // floats per region: n_vis * n_hid + n_hid + n_vis, rounded up to 4
    size_t bytes = 0;
    char* base = nullptr;             // this rank's buffer
    char* peer[kurbm::PEER_MAX] = {}; // every rank's buffer as mapped here (peer[rank] == base)
    bool opened[kurbm::PEER_MAX] = {};
    int connected = 0;
    unsigned epoch = 0;
    unsigned long long timeout_ticks = 1000000000ull;   // of the 100 MHz constant clock (s_memrealtime)
    int fine_grained = 0;
};

namespace {

constexpr size_t FLAGS_BYTES = 512;
constexpr int OFF_READY = 0, OFF_SUMMED = 64, OFF_COUNTER = 128;

#define HIP_TRY(expr)                                                                                 \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) return fail_msg(KURBM_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

inline size_t region_bytes(size_t n) { return (n * sizeof(float) + 255) / 256 * 256; }
inline float* delta_of(char* b) { return reinterpret_cast<float*>(b + FLAGS_BYTES); }
inline float* sum_of(char* b, size_t n) { return reinterpret_cast<float*>(b + FLAGS_BYTES + region_bytes(n)); }
inline unsigned* flag_of(char* b, int off) { return reinterpret_cast<unsigned*>(b + off); }

struct PeerBandArgs {
    const float* delta[kurbm::PEER_MAX];
    const unsigned* ready[kurbm::PEER_MAX];
    float* sum;
    unsigned* my_ready;
    unsigned* my_summed;
    unsigned* counter;
    unsigned* status;
    unsigned long long timeout_ticks;
    unsigned epoch;
    int nranks;
    long long lo4, hi4;            // this rank's band of the packed buffer, in float4 units
    long long tail_lo4, tail_hi4;  // the range EVERY rank sums (the bias tail; empty for the plain all-reduce)
};

using kurbm::wait_flag_ge;

__global__ __launch_bounds__(256) void k_peer_sum_band(PeerBandArgs a) {
    __shared__ int ok;
    const int t = threadIdx.x;
    if (t == 0) ok = 1;
    if (blockIdx.x == 0 && t == 0)   // everything this stream wrote to `delta` is complete (kernel boundary) and, with this release, visible
        __hip_atomic_store(a.my_ready, a.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    __syncthreads();
    if (t < a.nranks && !wait_flag_ge(a.ready[t], a.epoch, a.timeout_ticks)) ok = 0;
    __syncthreads();
    if (!ok) {   // a peer never arrived: report, leave `sum` and the summed flag alone (the peers' own waits then end the same way)
        if (t == 0) __hip_atomic_fetch_or(a.status, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const long long nb = a.hi4 - a.lo4, nt = a.tail_hi4 - a.tail_lo4;
    for (long long i = (long long)blockIdx.x * 256 + t; i < nb + nt; i += (long long)gridDim.x * 256) {
        const long long q = i < nb ? a.lo4 + i : a.tail_lo4 + (i - nb);
        f32x4 s = *reinterpret_cast<const f32x4*>(a.delta[0] + 4 * q);
        for (int r = 1; r < a.nranks; ++r) s += *reinterpret_cast<const f32x4*>(a.delta[r] + 4 * q);   // rank order: the same bits on every rank
        *reinterpret_cast<f32x4*>(a.sum + 4 * q) = s;
    }
    __threadfence_system();
    __syncthreads();
    if (t == 0) {
        const unsigned done = __hip_atomic_fetch_add(a.counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (done == gridDim.x - 1) {   // the last workgroup: every band element has been stored (and fenced) -- publish
            __hip_atomic_store(a.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.my_summed, a.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

struct PeerGatherArgs {
    const float* sum[kurbm::PEER_MAX];
    const unsigned* summed[kurbm::PEER_MAX];
    float* out;
    unsigned* status;
    unsigned long long timeout_ticks;
    unsigned epoch;
    int nranks;
    long long band4, n4;   // float4s per band, in all
};

// shot 2 of the plain all-reduce: out[band q] = rank q's sum[band q]; one workgroup per 1024 floats of a band
__global__ __launch_bounds__(256) void k_peer_gather(PeerGatherArgs a) {
    __shared__ int ok;
    const int t = threadIdx.x;
    const long long i0 = (long long)blockIdx.x * 256;
    if (i0 >= a.n4) return;
    const int q = (int)(i0 / a.band4);          // (band4 is a multiple of 256: a workgroup lies in one band)
    if (t == 0) ok = wait_flag_ge(a.summed[q], a.epoch, a.timeout_ticks) ? 1 : 0;
    __syncthreads();
    if (!ok) {
        if (t == 0) __hip_atomic_fetch_or(a.status, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const long long i = i0 + t;
    if (i < a.n4) *reinterpret_cast<f32x4*>(a.out + 4 * i) = *reinterpret_cast<const f32x4*>(a.sum[q] + 4 * i);
}

int check_peer(const kurbm_peer* x) {
    if (!x) return fail_msg(KURBM_ERR_ARG, "peer exchange is null");
    if (!x->connected) return fail_msg(KURBM_ERR_ARG, "peer exchange is not connected (kurbm_peer_connect)");
    return KURBM_OK;
}

// shot 1 over the packed buffer: bands [r band4, (r + 1) band4) of the first `w4` float4s, every rank also [tail_lo4, n4)
int launch_sum_band(kurbm_peer* x, unsigned* status, long long band4, long long w4, long long tail_lo4, long long n4, hipStream_t st) {
    PeerBandArgs a;
    memset(&a, 0, sizeof a);
    for (int r = 0; r < x->nranks; ++r) {
        a.delta[r] = delta_of(x->peer[r]);
        a.ready[r] = flag_of(x->peer[r], OFF_READY);
    }
    a.sum = sum_of(x->base, x->n);
    a.my_ready = flag_of(x->base, OFF_READY);
    a.my_summed = flag_of(x->base, OFF_SUMMED);
    a.counter = flag_of(x->base, OFF_COUNTER);
    a.status = status;
    a.timeout_ticks = x->timeout_ticks;
    a.epoch = x->epoch;
    a.nranks = x->nranks;
    a.lo4 = (long long)x->rank * band4 < w4 ? (long long)x->rank * band4 : w4;
    a.hi4 = a.lo4 + band4 < w4 ? a.lo4 + band4 : w4;
    a.tail_lo4 = tail_lo4; a.tail_hi4 = n4;
    const long long work = (a.hi4 - a.lo4) + (n4 - tail_lo4);
    long long nblk = (work + 1023) / 1024;      // four float4 per thread
    if (nblk < 1) nblk = 1;
    if (nblk > 128) nblk = 128;                 // a small grid: it must not keep a peer PROCESS on the same GPU from running (tests)
    hipLaunchKernelGGL(k_peer_sum_band, dim3((unsigned)nblk), dim3(256), 0, st, a);
    HIP_TRY(hipGetLastError());
    return KURBM_OK;
}

}  // namespace

namespace kurbm {

// (kurbm_api.hip: kurbm_cd_step_x3_peer) the exchange of the packed sums that the step has just written to peer_delta(x):
// shot 1 here, and what shot 2 -- the reduce / apply launch -- needs to read the bands
float* peer_delta(kurbm_peer* x) { return delta_of(x->base); }
int peer_geometry_ok(const kurbm_peer* x, int device, int n_vis, int n_hid) {
    if (int e = check_peer(x)) return e;
    if (x->device != device) return fail_msg(KURBM_ERR_ARG, "peer exchange is on device %d, context on %d", x->device, device);
    if (x->n_vis != n_vis || x->n_hid != n_hid) return fail_msg(KURBM_ERR_ARG, "peer exchange was created for %d x %d", x->n_vis, x->n_hid);
    if (n_hid & 3) return fail_msg(KURBM_ERR_UNSUPPORTED, "the peer exchange needs n_hid %% 4 == 0 (16-byte rows of the packed dW)");
    return KURBM_OK;
}
int peer_band_rows(const kurbm_peer* x) { return ((x->n_vis + x->nranks - 1) / x->nranks + 31) / 32 * 32; }
int peer_exchange_shot1(kurbm_peer* x, unsigned* status, hipStream_t st, PeerSrc* src) {
    ++x->epoch;
    const long long row4 = x->n_hid / 4;
    const long long w4 = (long long)x->n_vis * row4, n4 = (long long)(x->n / 4);
    const int band_rows = peer_band_rows(x);
    if (int e = launch_sum_band(x, status, (long long)band_rows * row4, w4, w4, n4, st)) return e;
    memset(src, 0, sizeof *src);
    for (int r = 0; r < x->nranks; ++r) {
        src->sum[r] = sum_of(x->peer[r], x->n);
        src->summed[r] = flag_of(x->peer[r], OFF_SUMMED);
    }
    src->own_sum = sum_of(x->base, x->n);
    src->status = status;
    src->timeout_ticks = x->timeout_ticks;
    src->epoch = x->epoch;
    src->band_rows = band_rows;
    src->nranks = x->nranks;
    return KURBM_OK;
}

}  // namespace kurbm

extern "C" {

size_t kurbm_peer_handle_bytes(void) { return sizeof(hipIpcMemHandle_t); }

int kurbm_peer_create(int device, int nranks, int rank, int n_vis, int n_hid, kurbm_peer** out) {
    if (!out) return fail_msg(KURBM_ERR_ARG, "out is null");
    *out = nullptr;
    if (nranks < 1 || nranks > kurbm::PEER_MAX || rank < 0 || rank >= nranks) return fail_msg(KURBM_ERR_ARG, "rank %d of %d (at most %d ranks)", rank, nranks, kurbm::PEER_MAX);
    if (n_vis <= 0 || n_hid <= 0) return fail_msg(KURBM_ERR_ARG, "n_vis / n_hid must be positive");
    int prev = 0;
    HIP_TRY(hipGetDevice(&prev));
    HIP_TRY(hipSetDevice(device));
    kurbm_peer* x = new kurbm_peer;
    x->device = device; x->nranks = nranks; x->rank = rank; x->n_vis = n_vis; x->n_hid = n_hid;
    x->n = ((size_t)n_vis * n_hid + n_hid + n_vis + 3) / 4 * 4;
    x->bytes = FLAGS_BYTES + 2 * region_bytes(x->n);
    if (const char* ms = getenv("KURBM_PEER_TIMEOUT_MS")) { if (atoll(ms) > 0) x->timeout_ticks = (unsigned long long)atoll(ms) * 100000ull; }
    // fine-grained device memory: system-scope atomics and fences on it are what another device observes over xGMI.
    // (KURBM_PEER_COARSE=1: plain hipMalloc -- enough for processes that share ONE device, where the L2 is common.)
    hipError_t e = hipErrorUnknown;
    const char* coarse = getenv("KURBM_PEER_COARSE");
    if (!(coarse && atoi(coarse) > 0)) {
        e = hipExtMallocWithFlags(reinterpret_cast<void**>(&x->base), x->bytes, hipDeviceMallocFinegrained);
        if (e == hipSuccess) x->fine_grained = 1;
        else (void)hipGetLastError();
    }
    if (e != hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&x->base), x->bytes);
    if (e == hipSuccess) e = hipMemset(x->base, 0, x->bytes);
    (void)hipSetDevice(prev);
    if (e != hipSuccess) {
        if (x->base) (void)hipFree(x->base);
        delete x;
        return fail_msg(KURBM_ERR_HIP, "exchange buffer (%zu bytes): %s", x->bytes, hipGetErrorString(e));
    }
    x->peer[rank] = x->base;
    if (nranks == 1) x->connected = 1;
    *out = x;
    return KURBM_OK;
}

int kurbm_peer_handle(kurbm_peer* x, void* handle, size_t handle_bytes) {
    if (!x || !handle) return fail_msg(KURBM_ERR_ARG, "null argument");
    if (handle_bytes < sizeof(hipIpcMemHandle_t)) return fail_msg(KURBM_ERR_ARG, "handle buffer must hold %zu bytes", sizeof(hipIpcMemHandle_t));
    hipIpcMemHandle_t h;
    int prev = 0;
    HIP_TRY(hipGetDevice(&prev));
    HIP_TRY(hipSetDevice(x->device));
    const hipError_t e = hipIpcGetMemHandle(&h, x->base);
    (void)hipSetDevice(prev);
    if (e != hipSuccess) return fail_msg(KURBM_ERR_HIP, "hipIpcGetMemHandle: %s", hipGetErrorString(e));
    memset(handle, 0, handle_bytes);
    memcpy(handle, &h, sizeof h);
    return KURBM_OK;
}

int kurbm_peer_connect(kurbm_peer* x, const void* handles, size_t bytes) {
    if (!x || !handles) return fail_msg(KURBM_ERR_ARG, "null argument");
    if (bytes < (size_t)x->nranks * sizeof(hipIpcMemHandle_t)) return fail_msg(KURBM_ERR_ARG, "handles: %d x %zu bytes expected", x->nranks, sizeof(hipIpcMemHandle_t));
    int prev = 0;
    HIP_TRY(hipGetDevice(&prev));
    HIP_TRY(hipSetDevice(x->device));
    int rc = KURBM_OK;
    for (int r = 0; r < x->nranks && rc == KURBM_OK; ++r) {
        if (r == x->rank || x->opened[r]) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, static_cast<const char*>(handles) + (size_t)r * sizeof h, sizeof h);
        void* p = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) rc = fail_msg(KURBM_ERR_HIP, "hipIpcOpenMemHandle (rank %d's buffer): %s", r, hipGetErrorString(e));
        else { x->peer[r] = static_cast<char*>(p); x->opened[r] = true; }
    }
    (void)hipSetDevice(prev);
    if (rc == KURBM_OK) x->connected = 1;
    return rc;
}

int kurbm_peer_ranks(const kurbm_peer* x) { return x ? x->nranks : fail_msg(KURBM_ERR_ARG, "peer exchange is null"); }

int kurbm_peer_allreduce_sum_f32(kurbm_ctx* ctx, kurbm_peer* x, float* buf, size_t n, kurbm_stream_t stream) {
    if (!ctx || (!buf && n)) return fail_msg(KURBM_ERR_ARG, "null argument");
    if (int e = check_peer(x)) return e;
    if (n == 0) return KURBM_OK;
    if (n > x->n) return fail_msg(KURBM_ERR_ARG, "the exchange buffer holds %zu floats, %zu asked for", x->n, n);
    if ((reinterpret_cast<uintptr_t>(buf) & 15u)) return fail_msg(KURBM_ERR_ARG, "buf must be 16-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    unsigned* status = kurbm::ctx_status_word(ctx);
    const size_t n4 = (n + 3) / 4;
    // the tail of a ragged last float4 beyond n: zeros from every rank
    if (n & 3) HIP_TRY(hipMemsetAsync(delta_of(x->base) + (n / 4) * 4, 0, 4 * sizeof(float), st));
    HIP_TRY(hipMemcpyAsync(delta_of(x->base), buf, n * sizeof(float), hipMemcpyDeviceToDevice, st));
    ++x->epoch;
    long long band4 = ((long long)n4 + x->nranks - 1) / x->nranks;
    band4 = (band4 + 255) / 256 * 256;
    if (int e = launch_sum_band(x, status, band4, (long long)n4, (long long)n4, (long long)n4, st)) return e;
    PeerGatherArgs g;
    memset(&g, 0, sizeof g);
    for (int r = 0; r < x->nranks; ++r) { g.sum[r] = sum_of(x->peer[r], x->n); g.summed[r] = flag_of(x->peer[r], OFF_SUMMED); }
    g.status = status; g.timeout_ticks = x->timeout_ticks; g.epoch = x->epoch; g.nranks = x->nranks;
    g.band4 = band4; g.n4 = (long long)(n / 4);
    // whole float4s go straight to buf; a ragged last one through the own delta region (free again: every peer is past shot 1
    // of this epoch once its summed flag is up -- but only the gather knows; so the ragged tail is copied from the owner's sum
    // by a second, tiny gather into a scratch float4 of the own flags block)
    g.out = buf;
    if (g.n4 > 0) {
        hipLaunchKernelGGL(k_peer_gather, dim3((unsigned)((g.n4 + 255) / 256)), dim3(256), 0, st, g);
        HIP_TRY(hipGetLastError());
    }
    if (n & 3) {
        // the last (partial) float4: gathered whole into scratch, then its valid floats copied out
        float* scratch = reinterpret_cast<float*>(x->base + 256);
        PeerGatherArgs t = g;
        const long long last = (long long)(n / 4);
        const int q = (int)(last / band4);
        t.sum[0] = g.sum[q] + 4 * last; t.summed[0] = g.summed[q];
        t.out = scratch; t.band4 = 256; t.n4 = 1;
        hipLaunchKernelGGL(k_peer_gather, dim3(1), dim3(256), 0, st, t);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(buf + 4 * last, scratch, (n & 3) * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    return KURBM_OK;
}

void kurbm_peer_destroy(kurbm_peer* x) {
    if (!x) return;
    int prev = 0;
    const bool have_prev = hipGetDevice(&prev) == hipSuccess;
    (void)hipSetDevice(x->device);
    (void)hipDeviceSynchronize();
    for (int r = 0; r < x->nranks; ++r)
        if (x->opened[r] && x->peer[r]) (void)hipIpcCloseMemHandle(x->peer[r]);
    if (x->base) (void)hipFree(x->base);
    if (have_prev) (void)hipSetDevice(prev);
    delete x;
}

}  // extern "C"
