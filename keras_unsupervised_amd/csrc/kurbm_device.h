// kurbm_device.h -- device helpers shared by the fp32 and bf16 kernel files.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kurbm {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// The kernel-argument segment is cold in the scalar cache at every launch, and hipcc fetches the arguments of a large struct as
// they are needed: k_gemm_pb's entry waited for SEVEN scalar loads one after the other, ~7 000 cycles (3.3 us) between a wave's
// first instruction and its first request for data (s_memtime stamps).  Touch every 64-byte line
// of the segment at once, wait once: the argument loads behind this hit in the scalar cache.
// (Inline asm: as plain loads hipcc folds them into its own argument loads and waits three times.  ONE asm statement holds
//  every load AND the wait: the destination registers are early-clobber outputs that nothing reads, so the compiler can neither
//  copy nor spill one while a load is still in flight to it, and its own lgkmcnt bookkeeping never sees these loads pending --
//  the statement ends with the counter at zero (scalar loads return out of order, so the compiler's own waits for whatever it
//  had in flight are lgkmcnt(0) too and stay correct).  Up to 12 lines = 768 bytes; a shorter segment
//  repeats its last line, so no load reaches past the segment.)
typedef const __attribute__((address_space(4))) void* kernarg_ptr_t;
template <int NBYTES>
__device__ __forceinline__ void kernarg_warm(kernarg_ptr_t ka) {
    constexpr int L = (NBYTES + 63) / 64;
    static_assert(L >= 1 && L <= 12, "argument segment: at most 12 lines of 64 bytes");
#define KURBM_KA_OFF(i) "i"(64 * ((i) < L ? (i) : L - 1))
    uint32_t d0, d1, d2, d3, d4, d5, d6, d7, d8, d9, d10, d11;
    asm volatile(
        "s_load_dword %0, %12, %13\n\ts_load_dword %1, %12, %14\n\ts_load_dword %2, %12, %15\n\ts_load_dword %3, %12, %16\n\t"
        "s_load_dword %4, %12, %17\n\ts_load_dword %5, %12, %18\n\ts_load_dword %6, %12, %19\n\ts_load_dword %7, %12, %20\n\t"
        "s_load_dword %8, %12, %21\n\ts_load_dword %9, %12, %22\n\ts_load_dword %10, %12, %23\n\ts_load_dword %11, %12, %24\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&s"(d0), "=&s"(d1), "=&s"(d2), "=&s"(d3), "=&s"(d4), "=&s"(d5), "=&s"(d6), "=&s"(d7), "=&s"(d8), "=&s"(d9), "=&s"(d10),
          "=&s"(d11)
        : "s"(ka), KURBM_KA_OFF(0), KURBM_KA_OFF(1), KURBM_KA_OFF(2), KURBM_KA_OFF(3), KURBM_KA_OFF(4), KURBM_KA_OFF(5), KURBM_KA_OFF(6),
          KURBM_KA_OFF(7), KURBM_KA_OFF(8), KURBM_KA_OFF(9), KURBM_KA_OFF(10), KURBM_KA_OFF(11)
        : "memory");
#undef KURBM_KA_OFF
}
#ifndef KURBM_WARM_ARGS
#define KURBM_WARM_ARGS 1   // (0: A/B builds)
#endif
template <int NBYTES>
__device__ __forceinline__ void warm_kernel_arguments() {
    if (KURBM_WARM_ARGS) kernarg_warm<NBYTES>((kernarg_ptr_t)__builtin_amdgcn_kernarg_segment_ptr());
}

// ------------------------------------------------------------------------------------
// Philox4x32-10 (Random123 constants) -- same contract as oracle/philox.py
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32x32->64 multiply (v_mad_u64_u32) per product instead of a mul_hi / mul_lo pair
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ float u32_to_unit(uint32_t x) {
    return __uint_as_float((x & 0x7FFFFFu) | 0x3F800000u) - 1.0f;
}

__device__ __forceinline__ float sigmoidf_fast(float x) {
    // v_exp_f32 + v_rcp_f32 (1 ulp each): __fdividef expands to the full IEEE division sequence on
    // gfx950 (div_scale / rcp / 4 fma / div_fmas / div_fixup), ten instructions per element
    return __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}

// z = sqrt(-2 ln(1 - ua)) cos(2 pi ub): one standard normal from two uniforms of [0, 1) (oracle/philox.py: normal()).
// On the hardware transcendentals -- v_log_f32 (log2), v_sqrt_f32, v_cos_f32 (its argument is in REVOLUTIONS: cos(2 pi x)) --
// at 8 cycles each instead of three libm calls of ~20-40 instructions: 32 draws per lane and launch in the Gaussian h->v
// epilogue.  ln(1 - ua) for small ua by its series (1 - ua sits next to 1, where log2's absolute error is all there is).
// KURBM_PRECISE_NORMAL: the libm form.
__device__ __forceinline__ float box_muller(float ua, float ub) {
#ifdef KURBM_PRECISE_NORMAL
    return sqrtf(-2.0f * logf(1.0f - ua)) * cospif(2.0f * ub);
#else
    const float ln_series = -ua * (1.0f + ua * (0.5f + ua * (0.33333334f + 0.25f * ua)));     // |rel. error| < ua^4 / 5
    const float ln_hw = 0.69314718f * __builtin_amdgcn_logf(1.0f - ua);
    const float ln1m = ua < 0.0078125f ? ln_series : ln_hw;
    return __builtin_amdgcn_sqrtf(-2.0f * ln1m) * __builtin_amdgcn_cosf(ub);
#endif
}

// log(1 + e^x), overflow-free: max(x, 0) + log(1 + t), t = e^-|x| in (0, 1].  On the hardware transcendentals (v_exp_f32,
// v_log_f32: 8 cycles each): libm's log1pf is ~40 instructions, and a free-energy epilogue evaluates 32 of these per lane -- the
// softplus GEMM took 35 us against 25 for the same GEMM without it.  log(1 + t) for small t by its series (1 + t sits next to 1,
// where log2's absolute error, ~1e-7, would be the whole result); elsewhere that absolute error is < 2e-7 of a term >= 0.0078.
__device__ __forceinline__ float softplusf(float x) {
    const float t = __expf(-fabsf(x));
    const float series = t * (1.0f - t * (0.5f - 0.33333334f * t));                 // |error| < t^4 / 4 < 1e-9
    const float hw = 0.69314718f * __builtin_amdgcn_logf(1.0f + t);
    return fmaxf(x, 0.0f) + (t < 0.0078125f ? series : hw);
}


// float -> bf16 bits, round to nearest even (inputs are finite on this path)
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float x) {
    const uint32_t u = __float_as_uint(x);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float bf16_bits_to_f32(uint32_t h) { return __uint_as_float(h << 16); }
// two floats -> packed bf16 pair, round to nearest even: ONE gfx950 instruction (no builtin for it in ROCm 7.2)
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
// neighbouring lane of the pair (lane ^ 1) by DPP quad_perm [1,0,3,2]: no LDS crossbar round trip
__device__ __forceinline__ float pair_swap(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));
}


// Sum over the 16 lanes of a DPP row (the lanes that share lane >> 4), the total in every lane: four DPP adds on the vector ALU.
// (As __shfl_xor chains -- ds_bpermute through the LDS crossbar, four dependent ones per sum, sixteen sums per lane -- the row sums of
//  a softplus epilogue took ~8 000 cycles of a half step: stamps of the score's GEMMs, round 4.)
__device__ __forceinline__ float row16_sum(float t) {
    t += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(t), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    t += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(t), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    t += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(t), 0x124, 0xF, 0xF, true));   // row_ror:4
    t += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(t), 0x128, 0xF, 0xF, true));   // row_ror:8
    return t;
}

// Exact three-way split of an fp32 value into bf16 pieces, x == hi + mid + lo (round to nearest even at
// every stage: the two residuals are exact in fp32 and the last one has at most 8 significant bits).
// piece j of x, j = 0, 1, 2:
__device__ __forceinline__ uint32_t bf16_piece_bits(float x, int j) {
    uint32_t h = f32_to_bf16_bits(x);
    if (j == 0) return h;
    x -= bf16_bits_to_f32(h);
    h = f32_to_bf16_bits(x);
    if (j == 1) return h;
    x -= bf16_bits_to_f32(h);
    return f32_to_bf16_bits(x);
}


// Bounded wait on a flag another agent writes (the peer exchange, kurbm_peer.hip): until *flag >= epoch, by system-scope acquire
// loads -- the writer is another device, or another process on this one -- with the constant 100 MHz clock as the bound.  False on
// a timeout: the caller reports through the context's status word and skips its work; nothing spins for ever.
__device__ __forceinline__ unsigned long long realtime_ticks() {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
__device__ inline bool wait_flag_ge(const unsigned* flag, unsigned epoch, unsigned long long timeout_ticks) {
    const unsigned long long t0 = realtime_ticks();
    for (;;) {
        if (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) >= epoch) return true;   // (epochs only grow: 2^32 steps)
        __builtin_amdgcn_s_sleep(8);
        if (realtime_ticks() - t0 > timeout_ticks) return false;
    }
}

// Byte planes of 0/1 values (A operand of k_gemm_pb<..., AB>) are K-PERMUTED inside every group of 64 elements: element
// 32 ks + 8 s + j (j < 8) of a group sits at byte 16 s + 8 ks + j, so that the bytes of both k-steps which lane group s of the
// MFMA operand needs are 16 consecutive bytes.  c -> the position of column c in its row.
__host__ __device__ inline int kperm64(int c) { return (c & ~0x38) | ((c & 0x18) << 1) | ((c & 0x20) >> 2); }

}  // namespace kurbm
