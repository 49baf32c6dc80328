// kurbm_comm.hip -- kurbm_comm_*: the sum all-reduce of the packed [dW | db_h | db_v] statistics over the GPUs of a
// node, on RCCL (xGMI).  Reference: none -- ku/ebm has no multi-device path; the exchange follows from
// rbm.py:125-134 being SUMS over the batch.
//
// RCCL is bound at run time (dlopen of the soname librccl.so.1, so a host process that already carries an RCCL --
// a PyTorch process does -- shares that one instance; KURBM_RCCL_LIB names another file), never at link time:
// the single-GPU library has no RCCL dependency and loads on a machine without it.
//
// The handful of RCCL types and constants this file needs are declared here, as rccl.h (2.x) declares them, so that the
// library BUILDS without the RCCL headers too; the functions themselves come from dlsym.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;          // NCCL_UNIQUE_ID_BYTES
typedef enum { ncclSuccess = 0 } ncclResult_t;                // (any other value: passed to ncclGetErrorString)
typedef enum { ncclSum = 0 } ncclRedOp_t;
typedef enum { ncclFloat32 = 7 } ncclDataType_t;
}

#include <cstdlib>
#include <cstring>
#include <mutex>

#include "../../include/kurbm.h"
#include "kurbm_comm.h"

using kurbm::fail_msg;

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    const char* error = nullptr;
};

Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {getenv("KURBM_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            if (!n || !*n) continue;
            r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
        }
        if (!r.handle) { r.error = "librccl.so.1 not found (set KURBM_RCCL_LIB)"; return; }
#define KURBM_SYM(field, name)                                                  \
        r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.handle, name));  \
        if (!r.field) { r.error = "RCCL library lacks " name; return; }
        KURBM_SYM(GetUniqueId, "ncclGetUniqueId")
        KURBM_SYM(CommInitRank, "ncclCommInitRank")
        KURBM_SYM(CommInitAll, "ncclCommInitAll")
        KURBM_SYM(CommDestroy, "ncclCommDestroy")
        KURBM_SYM(CommCount, "ncclCommCount")
        KURBM_SYM(CommUserRank, "ncclCommUserRank")
        KURBM_SYM(AllReduce, "ncclAllReduce")
        KURBM_SYM(GetErrorString, "ncclGetErrorString")
#undef KURBM_SYM
    });
    return &r;
}

#define RCCL_TRY(R, expr)                                                                                     \
    do {                                                                                                      \
        ncclResult_t e_ = (expr);                                                                             \
        if (e_ != ncclSuccess) return fail_msg(KURBM_ERR_COMM, "%s: %s", #expr, (R)->GetErrorString(e_));     \
    } while (0)
#define HIP_TRY(expr)                                                                                 \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) return fail_msg(KURBM_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// the comm stream and the events of a communicator whose ncclComm_t exists; the device is current
int finish_comm(kurbm_comm* c) {
    // KURBM_COMM_PRIORITY=1: a high-priority comm stream (an all-reduce that is ready goes ahead of queued GEMM workgroups).
    // Default: normal priority -- measured on MI355X / ROCm 7 (tools/dp_times.py): an event hand-off main -> side -> main
    // takes 32 us with a normal-priority side stream and 132 us with a high-priority one, and the two-range data-parallel
    // step 188 us against 745 us.
    const char* pr = getenv("KURBM_COMM_PRIORITY");
    if (pr && atoi(pr) > 0) {
        int least = 0, greatest = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIP_TRY(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, greatest));
    } else {
        HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    }
    for (int i = 0; i < kurbm_comm::MAX_CHUNKS; ++i) HIP_TRY(hipEventCreateWithFlags(&c->ev_ready[i], hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming));
    return KURBM_OK;
}

}  // namespace

namespace kurbm {
int comm_allreduce_sum(kurbm_comm* c, float* buf, size_t n, hipStream_t st) {
    Rccl* R = rccl();
    if (R->error) return fail_msg(KURBM_ERR_COMM, "%s", R->error);
    if (n == 0) return KURBM_OK;
    RCCL_TRY(R, R->AllReduce(buf, buf, n, ncclFloat32, ncclSum, static_cast<ncclComm_t>(c->nccl), st));
    return KURBM_OK;
}
}  // namespace kurbm

extern "C" {

int kurbm_comm_unique_id(void* id, size_t id_bytes) {
    Rccl* R = rccl();
    if (R->error) return fail_msg(KURBM_ERR_COMM, "%s", R->error);
    if (!id || id_bytes < sizeof(ncclUniqueId)) return fail_msg(KURBM_ERR_ARG, "id buffer must hold %zu bytes", sizeof(ncclUniqueId));
    ncclUniqueId u;
    RCCL_TRY(R, R->GetUniqueId(&u));
    memset(id, 0, id_bytes);
    memcpy(id, &u, sizeof u);
    return KURBM_OK;
}

int kurbm_comm_init_rank(int device, int nranks, int rank, const void* id, size_t id_bytes, kurbm_comm** out) {
    if (!out) return fail_msg(KURBM_ERR_ARG, "out is null");
    *out = nullptr;
    Rccl* R = rccl();
    if (R->error) return fail_msg(KURBM_ERR_COMM, "%s", R->error);
    if (!id || id_bytes < sizeof(ncclUniqueId)) return fail_msg(KURBM_ERR_ARG, "id must hold %zu bytes", sizeof(ncclUniqueId));
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail_msg(KURBM_ERR_ARG, "rank %d of %d", rank, nranks);
    // (the caller's current device is restored on every path, as kurbm_comm_init_all and kurbm_comm_destroy do)
    int prev = 0;
    HIP_TRY(hipGetDevice(&prev));
    HIP_TRY(hipSetDevice(device));
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    ncclComm_t nc = nullptr;
    const ncclResult_t r = R->CommInitRank(&nc, nranks, u, rank);
    if (r != ncclSuccess) {
        (void)hipSetDevice(prev);
        return fail_msg(KURBM_ERR_COMM, "ncclCommInitRank: %s", R->GetErrorString(r));
    }
    kurbm_comm* c = new kurbm_comm;
    c->nccl = nc; c->device = device; c->nranks = nranks; c->rank = rank;
    const int e = finish_comm(c);
    (void)hipSetDevice(prev);
    if (e) { kurbm_comm_destroy(c); return e; }
    *out = c;
    return KURBM_OK;
}

int kurbm_comm_init_all(int ndev, const int* devs, kurbm_comm** out) {
    if (!out || ndev < 1 || ndev > 64) return fail_msg(KURBM_ERR_ARG, "bad arguments");
    for (int i = 0; i < ndev; ++i) out[i] = nullptr;
    Rccl* R = rccl();
    if (R->error) return fail_msg(KURBM_ERR_COMM, "%s", R->error);
    ncclComm_t nc[64];
    RCCL_TRY(R, R->CommInitAll(nc, ndev, devs));
    int prev = 0;
    HIP_TRY(hipGetDevice(&prev));
    int rc = KURBM_OK;
    for (int i = 0; i < ndev; ++i) {
        kurbm_comm* c = new kurbm_comm;
        c->nccl = nc[i]; c->device = devs ? devs[i] : i; c->nranks = ndev; c->rank = i;
        out[i] = c;
        if (rc == KURBM_OK) {
            if (hipSetDevice(c->device) != hipSuccess) rc = fail_msg(KURBM_ERR_HIP, "hipSetDevice(%d) failed", c->device);
            else rc = finish_comm(c);
        }
    }
    (void)hipSetDevice(prev);
    if (rc != KURBM_OK)
        for (int i = 0; i < ndev; ++i) { kurbm_comm_destroy(out[i]); out[i] = nullptr; }
    return rc;
}

int kurbm_comm_count(const kurbm_comm* c) {
    if (!c) return fail_msg(KURBM_ERR_ARG, "comm is null");
    Rccl* R = rccl();
    int n = 0;
    RCCL_TRY(R, R->CommCount(static_cast<ncclComm_t>(c->nccl), &n));
    return n;
}

int kurbm_comm_rank(const kurbm_comm* c) {
    if (!c) return fail_msg(KURBM_ERR_ARG, "comm is null");
    Rccl* R = rccl();
    int r = 0;
    RCCL_TRY(R, R->CommUserRank(static_cast<ncclComm_t>(c->nccl), &r));
    return r;
}

int kurbm_allreduce_sum_f32(kurbm_comm* c, float* buf, size_t n, kurbm_stream_t stream) {
    if (!c || (!buf && n)) return fail_msg(KURBM_ERR_ARG, "null argument");
    return kurbm::comm_allreduce_sum(c, buf, n, static_cast<hipStream_t>(stream));
}

void kurbm_comm_destroy(kurbm_comm* c) {
    if (!c) return;
    Rccl* R = rccl();
    int prev = 0;
    const bool have_prev = hipGetDevice(&prev) == hipSuccess;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->nccl && !R->error) (void)R->CommDestroy(static_cast<ncclComm_t>(c->nccl));
    for (int i = 0; i < kurbm_comm::MAX_CHUNKS; ++i)
        if (c->ev_ready[i]) (void)hipEventDestroy(c->ev_ready[i]);
    if (c->ev_done) (void)hipEventDestroy(c->ev_done);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (have_prev) (void)hipSetDevice(prev);
    delete c;
}

}  // extern "C"
