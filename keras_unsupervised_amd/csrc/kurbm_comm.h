// kurbm_comm.h -- the data-parallel exchange of the CD update: one RCCL communicator per GPU, a comm stream and
// the events that order it against the compute stream, all owned by the library (include/kurbm.h: kurbm_comm_*).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

struct kurbm_comm {
    void* nccl = nullptr;            // ncclComm_t
    int device = 0, nranks = 1, rank = 0;
    hipStream_t stream = nullptr;    // all-reduces of the chunked step run here, beside the compute stream
    static constexpr int MAX_CHUNKS = 8;
    hipEvent_t ev_ready[MAX_CHUNKS] = {};   // compute stream: chunk c of the packed sums is complete
    hipEvent_t ev_done = nullptr;           // comm stream: the last all-reduce has finished
};

namespace kurbm {
// In-place sum all-reduce of n floats on stream `st` (any stream of the communicator's device).  0 or KURBM_ERR_*.
int comm_allreduce_sum(kurbm_comm* c, float* buf, size_t n, hipStream_t st);
// records the thread-local message kurbm_last_error() returns and passes `code` through (kurbm_api.hip)
int fail_msg(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
}  // namespace kurbm
