// any_order_probe.hip -- can a DEPENDENT kernel's workgroups start under the tail of its predecessor on the same stream?
//
// hipExtLaunchKernel(..., flags = hipExtAnyOrderLaunch) clears the AQL barrier bit of a dispatch: the command processor may
// launch it before the packet in front has COMPLETED.  Packets of one queue are still consumed in order, so every workgroup
// of kernel A has been placed before the first workgroup of kernel B is -- a B workgroup that spins on a flag of A can
// therefore never starve the A workgroup it waits for (one workgroup per CU: both kernels take 144 KB of LDS).
//
// Measured here, with s_memrealtime (100 MHz, one clock for the whole device):
//   A: 256 workgroups x 768 threads, spin ~20 us, publish a flag per workgroup (sc1 store of a tagged word)
//   B: the same grid; records its start, then waits (bounded) for the flag of A workgroup (b + 37) % 256
//   gap     = min(B start) - max(A end)        negative: B workgroups were resident before A had finished
//   B done  = max(B end) - max(A end)
// three ways: plain launches (barrier bit set), any-order launch of B, and any-order with the spin dependency.
//   hipcc -O2 --offload-arch=gfx950 tools/probes/any_order_probe.hip -o /tmp/anyorder && /tmp/anyorder
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) unsigned gu32;

__device__ __forceinline__ u64 now() {
    u64 t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

// rec[b] = {start, flag seen, end, timed out}
__global__ __launch_bounds__(768) void k_phase(unsigned* flags_out, const unsigned* flags_in, unsigned epoch, u64* rec, int spin_ticks,
                                                int dep_shift, int vary) {
    __shared__ unsigned char big[144 * 1024];   // one workgroup per CU
    const int b = blockIdx.x;
    u64 t0 = 0, t1 = 0, tmo = 0;
    if (threadIdx.x == 0) {
        t0 = now();
        if (flags_in) {
            const unsigned* f = flags_in + ((b + dep_shift) % gridDim.x) * 32;   // (a 128-byte line per flag)
            u64 lim = t0 + 200000;   // 2 ms
            while (__hip_atomic_load((gu32*)f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
                __builtin_amdgcn_s_sleep(2);
                if (now() > lim) { tmo = 1; break; }
            }
        }
        t1 = now();
        while (now() < t1 + spin_ticks + vary * ((b * 7) % 16)) __builtin_amdgcn_s_sleep(1);   // (vary: A's workgroups end up to 15 x vary ticks apart)
        big[b & 1023] = (unsigned char)b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (flags_out) __hip_atomic_store((gu32*)(flags_out + b * 32), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        u64* r = rec + (size_t)b * 4;
        r[0] = t0; r[1] = t1; r[2] = now(); r[3] = tmo + big[(b + 1) & 1023] * 0;
    }
}

int main() {
    const int G = 256;
    unsigned *fa, *fb;
    u64 *ra, *rb;
    CK(hipMalloc(&fa, G * 128));
    CK(hipMalloc(&fb, G * 128));
    CK(hipMalloc(&ra, G * 32));
    CK(hipMalloc(&rb, G * 32));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    std::vector<u64> ha(G * 4), hb(G * 4);
    const char* names[3] = {"plain", "any-order", "any-order + flag wait"};
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 4; ++rep) {
            const unsigned epoch = 1 + mode * 16 + rep;
            CK(hipMemsetAsync(fa, 0, G * 128, st));
            CK(hipMemsetAsync(ra, 0, G * 32, st));
            CK(hipMemsetAsync(rb, 0, G * 32, st));
            const unsigned* none = nullptr;
            unsigned* noneo = nullptr;
            hipLaunchKernelGGL(k_phase, dim3(G), dim3(768), 0, st, fa, none, epoch, ra, 2000, 0, 50);
            if (mode == 0)
                hipLaunchKernelGGL(k_phase, dim3(G), dim3(768), 0, st, noneo, none, epoch, rb, 2000, 0, 0);
            else
                hipExtLaunchKernelGGL(k_phase, dim3(G), dim3(768), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, noneo,
                                      mode == 2 ? (const unsigned*)fa : none, epoch, rb, 2000, 37, 0);
            CK(hipGetLastError());
            CK(hipStreamSynchronize(st));
            CK(hipMemcpy(ha.data(), ra, G * 32, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hb.data(), rb, G * 32, hipMemcpyDeviceToHost));
            u64 a_end = 0, a_start = ~0ull, b_start = ~0ull, b_start_max = 0, b_end = 0, tmo = 0, early = 0;
            for (int i = 0; i < G; ++i) {
                a_start = std::min(a_start, ha[i * 4]);
                a_end = std::max(a_end, ha[i * 4 + 2]);
                b_start = std::min(b_start, hb[i * 4]);
                b_start_max = std::max(b_start_max, hb[i * 4]);
                b_end = std::max(b_end, hb[i * 4 + 2]);
                tmo += hb[i * 4 + 3];
            }
            for (int i = 0; i < G; ++i) early += hb[i * 4] < a_end;
            if (rep)
                printf("%-22s A %.2f us; B first start %+.2f us / last start %+.2f us after A's last end; B done %+.2f us; "
                       "%llu B workgroups started before A ended; timeouts %llu\n",
                       names[mode], (a_end - a_start) * 0.01, ((double)b_start - (double)a_end) * 0.01,
                       ((double)b_start_max - (double)a_end) * 0.01, ((double)b_end - (double)a_end) * 0.01, early, tmo);
        }
    }
    return 0;
}
