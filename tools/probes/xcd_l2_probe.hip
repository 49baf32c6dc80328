// xcd_l2_probe.hip -- can two workgroups of ONE XCD hand a word to each other through that XCD's L2, past their L1s, without
// paying the memory-side round trip of an agent-scope (sc1) access?
//
// Workgroup 0 (XCD 0: workgroup i of a grid runs on XCD i % 8; the probe prints XCC_ID to confirm) waits ~30 us, then stores an
// epoch number into a flag word; workgroups 8, 16, ..., 248 (the other 31 CUs of XCD 0) have been polling that word since the
// launch began -- so a stale copy sits in their L1 -- and record when they first see the new value (s_memrealtime, 100 MHz, one
// clock for the device).  Methods of the poll:
//   0  plain load                       (expected: never sees it -- the L1 line stays valid)
//   1  load with sc0                    (workgroup scope: what the compiler emits in threadgroup-split mode)
//   2  buffer_inv sc0 + plain load
//   3  load with sc1                    (agent scope: past the L2 -- the reference, what kurbm_small.hip's grid barrier pays)
//   4  s_dcache_inv + scalar load       (the scalar cache misses into the L2, not through the vector L1)
//   5  atomic or 0 (returning)
//   6  load with nt
// The store: plain (methods 0-2, 4-6: it is written through the L1 into the L2) or sc1 (method 3).
// Output per method: how many of the 31 pollers saw the word, the median and the maximum delay behind the store.
//   hipcc -O2 --offload-arch=gfx950 tools/probes/xcd_l2_probe.hip -o /tmp/xcdl2 && /tmp/xcdl2
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef unsigned long long u64;

__device__ __forceinline__ u64 now() {
    u64 t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

template <int METHOD>
__device__ __forceinline__ unsigned poll(unsigned* p) {
    const __amdgpu_buffer_rsrc_t d = __builtin_amdgcn_make_buffer_rsrc(p, 0, 0xFFFFFFFF, 0x00020000);
    if constexpr (METHOD == 0) return (unsigned)__builtin_amdgcn_raw_buffer_load_b32(d, 0, 0, 0);
    if constexpr (METHOD == 1) return (unsigned)__builtin_amdgcn_raw_buffer_load_b32(d, 0, 0, 1);
    if constexpr (METHOD == 2) {
        asm volatile("buffer_inv sc0" ::: "memory");
        return (unsigned)__builtin_amdgcn_raw_buffer_load_b32(d, 0, 0, 0);
    }
    if constexpr (METHOD == 3) return (unsigned)__builtin_amdgcn_raw_buffer_load_b32(d, 0, 0, 16);
    if constexpr (METHOD == 4) {
        unsigned v;
        asm volatile("s_dcache_inv\n\ts_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
        return v;
    }
    if constexpr (METHOD == 5) return __hip_atomic_fetch_or(p, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if constexpr (METHOD == 6) return (unsigned)__builtin_amdgcn_raw_buffer_load_b32(d, 0, 0, 2);
    return 0;
}

template <int METHOD>
__global__ void k_probe(unsigned* flag, u64* out, unsigned* xcc_out, unsigned epoch) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) xcc_out[blockIdx.x] = xcc & 15u;
    if ((blockIdx.x & 7) != 0 || threadIdx.x != 0) return;
    const u64 t0 = now();
    if (blockIdx.x == 0) {
        while (now() - t0 < 3000) __builtin_amdgcn_s_sleep(8);     // 30 us: the pollers are in their loops, the word is in their L1s
        const __amdgpu_buffer_rsrc_t d = __builtin_amdgcn_make_buffer_rsrc(flag, 0, 0xFFFFFFFF, 0x00020000);
        const u64 t = now();
        __builtin_amdgcn_raw_buffer_store_b32(epoch, d, 0, 0, METHOD == 3 ? 16 : 0);
        out[0] = t;
    } else {
        u64 seen = 0;
        while (now() - t0 < 20000) {                               // 200 us at most
            if (poll<METHOD>(flag) == epoch) { seen = now(); break; }
            __builtin_amdgcn_s_sleep(1);
        }
        out[blockIdx.x >> 3] = seen;
    }
}

template <int METHOD>
static int run(const char* name, unsigned* flag, u64* out, unsigned* xcc, unsigned epoch) {
    CK(hipMemset(out, 0, 32 * sizeof(u64)));
    hipLaunchKernelGGL(k_probe<METHOD>, dim3(256), dim3(64), 0, nullptr, flag, out, xcc, epoch);
    CK(hipDeviceSynchronize());
    u64 h[32];
    CK(hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost));
    std::vector<double> d;
    for (int i = 1; i < 32; ++i)
        if (h[i]) d.push_back((double)((long long)(h[i] - h[0])) * 0.01);
    std::sort(d.begin(), d.end());
    if (d.empty()) printf("%-34s seen by  0 / 31\n", name);
    else printf("%-34s seen by %2zu / 31   delay behind the store: min %6.2f  median %6.2f  max %6.2f us\n", name, d.size(), d.front(),
                d[d.size() / 2], d.back());
    return 0;
}

int main() {
    unsigned *flag, *xcc;
    u64* out;
    CK(hipMalloc(reinterpret_cast<void**>(&flag), 4096));
    CK(hipMalloc(reinterpret_cast<void**>(&xcc), 256 * sizeof(unsigned)));
    CK(hipMalloc(reinterpret_cast<void**>(&out), 32 * sizeof(u64)));
    CK(hipMemset(flag, 0, 4096));
    unsigned epoch = 1;
    for (int rep = 0; rep < 2; ++rep) {
        if (run<0>("0 plain load", flag, out, xcc, epoch++)) return 1;
        if (run<1>("1 sc0 load", flag, out, xcc, epoch++)) return 1;
        if (run<2>("2 buffer_inv sc0 + plain load", flag, out, xcc, epoch++)) return 1;
        if (run<3>("3 sc1 load (sc1 store)", flag, out, xcc, epoch++)) return 1;
        if (run<4>("4 s_dcache_inv + scalar load", flag, out, xcc, epoch++)) return 1;
        if (run<5>("5 atomic or 0", flag, out, xcc, epoch++)) return 1;
        if (run<6>("6 nt load", flag, out, xcc, epoch++)) return 1;
    }
    unsigned hx[256];
    CK(hipMemcpy(hx, xcc, sizeof hx, hipMemcpyDeviceToHost));
    bool rr = true;
    for (int b = 8; b < 256; ++b) rr = rr && hx[b] == hx[b & 7];
    printf("XCC_ID of workgroups 0..7: %u %u %u %u %u %u %u %u; workgroup i on the XCD of workgroup i %% 8: %s\n", hx[0], hx[1], hx[2], hx[3],
           hx[4], hx[5], hx[6], hx[7], rr ? "yes" : "NO");
    return 0;
}
