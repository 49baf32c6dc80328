// stream_handoff_probe.hip -- what a dependency between two HIP streams costs on this GPU, three ways:
//   events        hipEventRecord + hipStreamWaitEvent
//   wait-value    hipStreamWriteValue32 + hipStreamWaitValue32 on signal memory
//   one stream    the same three kernels on one stream (the floor)
// Each iteration: kernel A on main, hand-off, kernel B on side, hand-off, kernel C on main.
//   hipcc -O2 --offload-arch=gfx950 tools/probes/stream_handoff_probe.hip -o /tmp/handoff && /tmp/handoff
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void touch(float* p, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] += 1.0f;
}

int main() {
    const int n = 1 << 20, iters = 300;
    float* x;
    CK(hipMalloc(&x, n * 4));
    CK(hipMemset(x, 0, n * 4));
    hipStream_t m, s;
    CK(hipStreamCreateWithFlags(&m, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e1, e2, t0, t1;
    CK(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
    CK(hipEventCreate(&t0));
    CK(hipEventCreate(&t1));
    uint32_t *f1, *f2;   // (signal memory comes in 8-byte allocations)
    CK(hipExtMallocWithFlags((void**)&f1, 8, hipMallocSignalMemory));
    CK(hipExtMallocWithFlags((void**)&f2, 8, hipMallocSignalMemory));
    CK(hipMemset(f1, 0, 8));
    CK(hipMemset(f2, 0, 8));
    auto run = [&](int mode, const char* name) -> int {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(t0, m));
            auto h0 = std::chrono::steady_clock::now();
            for (int i = 1; i <= iters; ++i) {
                const uint32_t v = rep * iters + i;
                touch<<<n / 256, 256, 0, m>>>(x, n);
                if (mode == 0) {
                    CK(hipEventRecord(e1, m));
                    CK(hipStreamWaitEvent(s, e1, 0));
                    touch<<<n / 256, 256, 0, s>>>(x, n);
                    CK(hipEventRecord(e2, s));
                    CK(hipStreamWaitEvent(m, e2, 0));
                } else if (mode == 1) {
                    CK(hipStreamWriteValue32(m, f1, v + 1000000u * mode, 0));
                    CK(hipStreamWaitValue32(s, f1, v + 1000000u * mode, hipStreamWaitValueGte, 0xFFFFFFFFu));
                    touch<<<n / 256, 256, 0, s>>>(x, n);
                    CK(hipStreamWriteValue32(s, f2, v + 1000000u * mode, 0));
                    CK(hipStreamWaitValue32(m, f2, v + 1000000u * mode, hipStreamWaitValueGte, 0xFFFFFFFFu));
                } else {
                    touch<<<n / 256, 256, 0, m>>>(x, n);
                }
                touch<<<n / 256, 256, 0, m>>>(x, n);
            }
            auto h1 = std::chrono::steady_clock::now();
            CK(hipEventRecord(t1, m));
            CK(hipDeviceSynchronize());
            float ms = 0.f;
            CK(hipEventElapsedTime(&ms, t0, t1));
            if (rep == 1)
                printf("%-12s %7.2f us per iteration on the GPU, %7.2f us of host enqueue\n", name, ms * 1e3f / iters,
                       std::chrono::duration<double, std::micro>(h1 - h0).count() / iters);
        }
        return 0;
    };
    if (run(2, "one stream")) return 1;
    if (run(0, "events")) return 1;
    if (run(1, "wait-value")) return 1;
    return 0;
}
