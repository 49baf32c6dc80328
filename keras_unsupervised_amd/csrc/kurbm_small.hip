// kurbm_small.hip -- one CD-1 update of a SMALL RBM in ONE launch (the reference example's own size: 784 -> 128, batch 128;
// BASELINE.json configs[0]: 784 x 256, batch 64).
//
// At these sizes a step is ~0.1-0.3 GFLOP: the five launches of kurbm_cd_step cost five launch latencies (~75 us per step,
// tools/bench_example_config.py) for ~3 us of arithmetic.  Here the whole step -- rbm.py:120-134 in the fused form: h_pos sample,
// v_neg sample, h_neg probabilities, dW / db_h / db_v applied -- is one grid-resident kernel of four phases.  Two schedules:
// `local` (default; the whole device is launched): a 16-row band of the batch stays on ONE XCD through phases 1-3 -- its planes
// and its barrier words go from CU to CU through that XCD's L2 (plain stores, non-temporal loads) -- and only phase 4 waits for
// the whole grid; else every phase spans the grid with a device-scope barrier behind it (counters and a generation word in the
// context's status block).  Every wait is bounded and reports through kurbm_ctx_status instead of hanging.  The score of
// fit(verbose = 1) is a second kernel on the same schedules (k_score_small, below).
//   1  h_pos = (u < act(v . W + b_h))            one 16 x 16 output tile per workgroup pass, its eight waves split k
//   2  v_neg = (u < sigmoid(h_pos . W^T + b_v))  or  loc + N(0, 1)   (Gaussian visibles: Box-Muller of two Philox planes)
//   3  h_neg = sigmoid(v_neg . W + b_h)
//   4  W += lr (v_pos^T h_pos - v_neg^T h_neg): one 16 x 16 tile of W per WAVE, k = the batch, applied from the accumulators (no
//      slabs); the bias column sums by waves of their own, fixed order (bit-reproducible).  Phases 1-3 leave their planes
//      TRANSPOSED as well ([unit][batch row]: the four rows a lane holds are one 16-byte store), so that k = the batch is
//      contiguous here: 16-byte loads instead of four strided dwords per operand and chunk
//      (`local` schedule: the tasks are dealt from the last workgroup down, and a workgroup with no tile in phase 2 or 3 computes the
//      positive half of its tile there, as soon as phase 1 is done on every XCD)
// Products on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 fma chains, like kurbm_kernels.hip), operands straight
// from L2 -- the whole problem is a few MB -- with the same k-slot permutation trick: a lane's four consecutive k feed four
// successive MFMAs.  Same Philox counters as every other path (include/kurbm.h), so the draws are the oracle's; the sums are
// added in another order than kurbm_cd_step's, so the two paths agree to fp32 rounding, not bit for bit.
// Scope: CD-1 from the data, no persistent chain, update applied in place (`which` honoured); anything else takes kurbm_cd_step.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "../../include/kurbm.h"
#include "kurbm_kernels.h"
#include "kurbm_comm.h"
#include "kurbm_device.h"

namespace kurbm {


typedef float f32x4 __attribute__((ext_vector_type(4)));

// All workgroups of the grid have arrived; false after a timeout.
// * No RELEASE fence: an agent-scope release writes back a whole L2 (the first version of this kernel took 180 us per step for its
//   fences).  The planes that cross a barrier -- h_pos, v_neg, h_neg -- are written by agent-scope (write-through) stores and a wave
//   drains its stores before it arrives; they are read by agent-scope loads (past this CU's L1 and this XCD's L2), so no acquire
//   either (one invalidate behind the barrier and plain loads instead: 73 against 53 us per step -- every phase loses its cached W).
// * Two levels: a workgroup arrives on one of eight counters (blockIdx.x % 8: one XCD's share of the grid, or a mix of two -- only the
//   count matters here), the last one of a counter on the grid's
//   counter, the last of those bumps the generation word everybody polls -- 256 arrivals on ONE address are 256 serialised atomics
//   at the memory side (~4 us); 32 on each of eight addresses in parallel, then 8, are not.
__device__ __forceinline__ bool grid_barrier(const SmallArgs& a, unsigned& gen) {
    __shared__ int ok;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        ok = 1;
        const int xcd = blockIdx.x & 7, nx = ((int)gridDim.x - xcd + 7) >> 3, ngroups = gridDim.x < 8 ? (int)gridDim.x : 8;
        unsigned* cx = a.bar + 16 * xcd;          // one 64-byte line per counter
        unsigned* cg = a.bar + 16 * 8;
        unsigned* gw = a.bar + 16 * 9;
        bool bump = false;
        if (__hip_atomic_fetch_add(cx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)nx - 1u) {
            __hip_atomic_store(cx, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__hip_atomic_fetch_add(cg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)ngroups - 1u) {
                __hip_atomic_store(cg, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bump = true;
            }
        }
        if (bump) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the counters are back at zero before anybody can arrive again)
            __hip_atomic_store(gw, gen + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            const unsigned long long t0 = realtime_ticks();
            while (__hip_atomic_load(gw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) {
                __builtin_amdgcn_s_sleep(1);
                if (realtime_ticks() - t0 > a.timeout_ticks) { ok = 0; break; }
            }
        }
    }
    __syncthreads();
    ++gen;
    return ok != 0;
}

// The XCD this workgroup runs on (XCC_ID, wave-uniform).  The `local` schedule groups workgroups by it -- NOT by blockIdx.x % 8: the
// dispatcher deals workgroups to the XCDs in turn, but a grid starts where the one before it stopped (found when the tests ran
// in another order: status bit 3 of the first version).  Within a group a workgroup's rank is blockIdx.x / 8: consecutive
// workgroups go to consecutive XCDs, so the eight of one octet land on eight different XCDs whatever the starting point, and each
// XCD gets every rank once; if a launch ever broke that, a rank would be missing in some group and its barrier would time out
// (status bit 2) -- never a silent wrong answer.
__device__ __forceinline__ int xcc_of_workgroup() {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    return (int)(xcc & 7u);
}

// The workgroups of ONE XCD (group g = its XCC_ID, up to 32 of them) have arrived: the barrier between the phases of the
// `local` schedule.  No atomics (they execute at the memory side): every workgroup owns ONE word of its group's 128-byte line,
// stores the barrier's number there (plain: through the L1 into the XCD's L2) and wave 0 polls the whole line with non-temporal
// loads, a lane per workgroup, until every word has reached that number -- one L2 round trip to arrive, one per poll (0.24 us
// each: xcd_l2_probe) where the grid's barrier pays three memory-side ones.  A wave has drained its plane stores before it
// arrives.  The numbers only grow (signed differences); a timeout reports through the status word like the grid barrier's.
__device__ __forceinline__ bool group_barrier(const SmallArgs& a, int g, unsigned target) {
    __shared__ int gok;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x < 64) {
        const int rank = blockIdx.x >> 3, n = (int)gridDim.x >> 3;
        const __amdgpu_buffer_rsrc_t d = __builtin_amdgcn_make_buffer_rsrc(a.bar + 160 + 32 * g, 0, 0xFFFFFFFF, 0x00020000);
        if (threadIdx.x == 0) __builtin_amdgcn_raw_buffer_store_b32(target, d, 4 * rank, 0, 0);
        const int lane = threadIdx.x;
        const unsigned long long t0 = realtime_ticks();
        bool ok = true;
        for (;;) {
            const unsigned f = lane < n ? (unsigned)__builtin_amdgcn_raw_buffer_load_b32(d, 4 * lane, 0, 2) : target;
            if (__all((int)(f - target) >= 0)) break;
            __builtin_amdgcn_s_sleep(1);
            if (realtime_ticks() - t0 > a.timeout_ticks) { ok = false; break; }
        }
        if (threadIdx.x == 0) gok = ok ? 1 : 0;
    }
    __syncthreads();
    return gok != 0;
}

// plane accessors.  SC = 1: agent scope (see grid_barrier) -- past this CU's L1 and this XCD's L2.  SC = 2: the scope of ONE XCD:
// a plain store (the L1 is write-through: the data is in the XCD's L2 once the store has drained) and a NON-TEMPORAL load, which
// is served by that L2 past the L1 -- for planes whose writer and readers all run on one XCD (the `local` schedule below) and
// which are never read any other way, so that no L1 ever holds a line of them.  tools/probes/xcd_l2_probe.hip measured the ways a
// word gets from one CU of an XCD to another: nt load 0.24 us behind the store, scalar load after s_dcache_inv 0.24, sc1 store +
// sc1 load 0.64; a plain load, an sc0 load (workgroup scope hits the L1 unless the kernel runs in threadgroup-split mode) and a
// load behind `buffer_inv sc0` never see it, and neither does an atomic (it executes at the memory side, behind the dirty L2
// line).  SC = 0: plain.
typedef uint32_t u32x4p __attribute__((ext_vector_type(4)));
template <int SC> struct ScopeAux { static constexpr int aux = SC == 1 ? 16 : (SC == 2 ? 2 : 0); };   // sc1 / nt
__device__ __forceinline__ __amdgpu_buffer_rsrc_t plane_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0xFFFFFFFF, 0x00020000);
}
// (`base` wave-uniform, the element index per lane: a buffer access takes its descriptor from scalar registers)
template <int SC> __device__ __forceinline__ float ld_plane(const float* base, unsigned idx) {
    if constexpr (SC == 0) return base[idx];
    else if constexpr (SC == 1) return __hip_atomic_load(base + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(plane_rsrc(base), (int)(4u * idx), 0, ScopeAux<2>::aux));
}
template <int SC> __device__ __forceinline__ void st_plane(float* base, unsigned idx, float v) {
    if constexpr (SC == 1) __hip_atomic_store(base + idx, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), plane_rsrc(base), (int)(4u * idx), 0, 0);
}
// ... 16 bytes at once: buffer accesses, `base` wave-uniform
template <int SC> __device__ __forceinline__ f32x4 ld_plane4(const float* base, unsigned byte_off) {
    if constexpr (SC == 0) return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + byte_off);
    else return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(plane_rsrc(base), (int)byte_off, 0, ScopeAux<SC>::aux));
}
__device__ __forceinline__ void st_plane4(float* base, unsigned byte_off, f32x4 v) {   // (agent scope)
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4p, v), plane_rsrc(base), (int)byte_off, 0, ScopeAux<1>::aux);
}

// One 16 x 16 tile of C = A . B over the k chunks [c0, c1) of 16 (stride cs): the lane (x = lane & 15, slot = lane >> 4) feeds
// k = 16 c + 4 slot + e to MFMA e of a chunk -- A and B agree on that map, so any operand layout works:
//   KC (k contiguous in memory):  one 16-byte load per chunk      X[x][k]  = base[x * ld + k]
//   KS (k strided):               four 4-byte loads per chunk     X[k][x]  = base[k * ld + x]
// x_ok / k < K guard ragged shapes (zeros contribute nothing).  NEG: the A values enter negated.
// A and B point at the TILE (KC: its first row; KS: its first column), so x = lane & 15 indexes both.
// A_PL / B_PL: the operand is a plane another workgroup wrote in an earlier phase -- loads of scope 1 (agent) or 2 (its XCD).
template <bool A_KC, bool B_KC, bool NEG, int A_PL = 0, int B_PL = 0, int UN = 8>
__device__ __forceinline__ void tile_mma(f32x4& acc, const float* __restrict__ A, int lda, bool a_ok, const float* __restrict__ B, int ldb, bool b_ok,
                                         int K, int c0, int c1, int cs, int x, int slot) {
    // UN chunks at a time: ALL their loads are issued before the first MFMA -- the operands come from L2 with a few waves per CU, so
    // a chunk-by-chunk loop would pay one L2 round trip per chunk.  (Built and measured TWICE, round 4, and NOT kept: every load
    // unconditional on clamped indices with a select behind it -- 222 branches instead of 641, but hipcc then waits for the loads one
    // by one (200 s_waitcnt vmcnt instead of 46): phase 1 took 20.5 instead of 5.6 us.  The guarded form keeps its loads in flight.)
    // (UN = 2 or 4 where a wave has no more chunks than that -- h -> v of a narrow hidden layer: the dead chunks of an unrolled eight
    //  would still issue their MFMAs, on zeros)
    for (int cb = c0; cb < c1; cb += UN * cs) {
        float av[UN][4], bv[UN][4];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int c = cb + u * cs;
            const int k = 16 * c + 4 * slot;
            const bool live = c < c1;
            if (A_KC) {
                if (live && a_ok && k + 3 < K) {
                    const f32x4 t = ld_plane4<A_PL>(A, 4u * (unsigned)(x * lda + k));
                    av[u][0] = t.x; av[u][1] = t.y; av[u][2] = t.z; av[u][3] = t.w;
                }
                else
#pragma unroll
                    for (int e = 0; e < 4; ++e) av[u][e] = (live && a_ok && k + e < K) ? ld_plane<A_PL>(A, (unsigned)(x * lda + k + e)) : 0.f;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) av[u][e] = (live && a_ok && k + e < K) ? ld_plane<A_PL>(A, (unsigned)((k + e) * lda + x)) : 0.f;
            }
            if (B_KC) {
                if (live && b_ok && k + 3 < K) {
                    const f32x4 t = ld_plane4<B_PL>(B, 4u * (unsigned)(x * ldb + k));
                    bv[u][0] = t.x; bv[u][1] = t.y; bv[u][2] = t.z; bv[u][3] = t.w;
                }
                else
#pragma unroll
                    for (int e = 0; e < 4; ++e) bv[u][e] = (live && b_ok && k + e < K) ? ld_plane<B_PL>(B, (unsigned)(x * ldb + k + e)) : 0.f;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[u][e] = (live && b_ok && k + e < K) ? ld_plane<B_PL>(B, (unsigned)((k + e) * ldb + x)) : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(NEG ? -av[u][e] : av[u][e], bv[u][e], acc, 0, 0, 0);
    }
}

// A half step over all 16 x 16 tiles of out [rows][N].  A workgroup takes one tile per pass and its eight waves split the k
// chunks -- or TWO neighbouring tiles per pass, four waves each, where that saves a pass (h -> v: 49 column tiles of a short k) --
// and meet in LDS (added in wave order); the first wave of a tile finishes it: bias, activation, draw, store -- row-major into
// `out` (nullable) and transposed into outT [N][ldt] (rows of the tile past the batch as zeros: phase 4 reads whole chunks).
// HV = false: out = f(in . W + b_h) (k = visible units, W read as [k][n]); HV = true: out = f(in . W^T + b_v) (W read as [n][k]).
constexpr int SMALL_WAVES = 8;
// IN_PL / OUT_SC: the scope of the input plane's loads and of the row-major output's stores (0 plain, 1 agent, 2 one XCD).
// LOCAL: row tile tm belongs to the workgroups of XCD tm % 8 (`grp` = the XCD this workgroup runs on: xcc_of_workgroup), which then
// exchange the row-major planes of phases 1-3 through their own L2 and meet at a barrier of their own.
// RP (the score's phases): the finishing wave also leaves a partial row sum of its tile -- over the tile's 16 columns, by DPP adds --
// in rowpart[tn][row] at agent scope: RP = 1 softplus of the pre-activations (the hidden term of a free energy, rbm.py:73-75),
// RP = 2 the outputs times this site's bias (v' . b_v, its visible term).
template <bool HV, int IN_PL, int OUT_SC, bool LOCAL, bool TWO, int RP = 0>
__device__ __forceinline__ void half_step_tiles(const SmallArgs& a, const float* __restrict__ in, int ld_in, float* __restrict__ out, int ldo,
                                float* __restrict__ outT, int act, int noise, const RngArgs& rng, float* red, float* __restrict__ rowpart = nullptr, int grp = 0) {
    const int K = HV ? a.n_hid : a.n_vis, N = HV ? a.n_vis : a.n_hid;
    const float* bias = HV ? a.b_v : a.b_h;
    const int tiles_m = (a.rows + 15) / 16, tiles_n = (N + 15) / 16, nch = (K + 15) / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, x = lane & 15, slot = lane >> 4;
    const int rank = blockIdx.x >> 3, ngrp = (int)gridDim.x >> 3;   // (LOCAL: the grid is a whole number of octets)
    // slots: the tiles (pairs of tiles) this workgroup's passes cover -- LOCAL: slot q = rank, rank + ngrp, ... of every row tile
    // tm = grp, grp + 8, ...; else q = blockIdx.x, + gridDim.x, ... of the whole list
    const int units = LOCAL ? tiles_n : tiles_m * tiles_n, cap = LOCAL ? ngrp : (int)gridDim.x;
    constexpr int per = TWO ? 2 : 1, kw = TWO ? 4 : SMALL_WAVES;
    const int sub = TWO ? (wave >> 2) : 0, wsub = TWO ? (wave & 3) : wave;
    const int nslots = (units + per - 1) / per;
    // LOCAL: the slots of ALL row tiles of this XCD (grp, grp + 8, ...) form one list dealt over its ranks -- at batch 256 the 16
    // tiles of a v -> h phase take one pass on 16 workgroups, not two passes on eight
    const int n_rt = LOCAL ? (tiles_m - grp + 7) / 8 : 1;
    for (int uq = LOCAL ? rank : (int)blockIdx.x; uq < n_rt * nslots; uq += cap) {
        const int rt = LOCAL ? uq / nslots : 0, q = uq - rt * nslots;
        const int u = q * per + sub;                       // this wave's tile among the units
        const bool live = !TWO || u < units;
        const int tm = LOCAL ? grp + 8 * rt : u / tiles_n, tn = LOCAL ? u : u - tm * tiles_n;
        const int m = tm * 16 + x, n = tn * 16 + x;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#ifdef KURBM_SMALL_STAMPS
#define KURBM_FST(i) do { if (HV && blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<unsigned long long*>(a.status + 768)[i] = realtime_ticks(); } while (0)
#else
#define KURBM_FST(i) do { } while (0)
#endif
        KURBM_FST(0);
        // (requested here: in the epilogue its latency -- an L2 round trip -- would be exposed, 0.5 us of every phase)
        const float bias_col = (wsub == 0 && live && n < N) ? bias[n] : 0.f;
        // ... and the finishing wave's Philox blocks do not depend on the product: drawn now, under the operand loads' latency
        uint32_t w1[4] = {0u, 0u, 0u, 0u}, w2[4] = {0u, 0u, 0u, 0u};
        if (wsub == 0 && live && n < N && noise != NOISE_NONE) {
            const uint64_t grow = rng.row0 + (uint64_t)(tm * 16 + 4 * slot);
            philox4x32_10((uint32_t)n, (uint32_t)(grow >> 2), rng.stream_id, rng.step, rng.seed_lo, rng.seed_hi, w1);
            if (noise == NOISE_GAUSSIAN)
                philox4x32_10((uint32_t)n, (uint32_t)(grow >> 2), rng.stream_id | 0x80000000u, rng.step, rng.seed_lo, rng.seed_hi, w2);
        }
        if (live) {
            const float* At = in + (size_t)tm * 16 * ld_in;                                      // rows of the batch, k contiguous
            if (HV) {                                                                            // W^T: rows = visible units
                const float* Bt = a.W + (size_t)tn * 16 * a.ldw;
                if (nch <= 2 * kw) tile_mma<true, true, false, IN_PL, 0, 2>(acc, At, ld_in, m < a.rows, Bt, a.ldw, n < N, K, wsub, nch, kw, x, slot);
                else if (nch <= 4 * kw) tile_mma<true, true, false, IN_PL, 0, 4>(acc, At, ld_in, m < a.rows, Bt, a.ldw, n < N, K, wsub, nch, kw, x, slot);
                else tile_mma<true, true, false, IN_PL>(acc, At, ld_in, m < a.rows, Bt, a.ldw, n < N, K, wsub, nch, kw, x, slot);
            }
            else    tile_mma<true, false, false, IN_PL>(acc, At, ld_in, m < a.rows, a.W + tn * 16, a.ldw, n < N, K, wsub, nch, kw, x, slot);                 // W as [k][n]
        }
        *reinterpret_cast<f32x4*>(red + (wave * 64 + lane) * 4) = acc;
        KURBM_FST(1);
        __syncthreads();
        KURBM_FST(2);
        if (wsub == 0 && live) {
            f32x4 s = *reinterpret_cast<const f32x4*>(red + (wave * 64 + lane) * 4);
            for (int w = 1; w < kw; ++w) s += *reinterpret_cast<const f32x4*>(red + ((wave + w) * 64 + lane) * 4);
            const int col = tn * 16 + x, row0 = tm * 16 + 4 * slot;     // C layout: lane holds rows row0 .. row0 + 3 of column col
            f32x4 rp = {0.f, 0.f, 0.f, 0.f};
            if (col < N) {
                const float b = bias_col;                                   // (col == n, row0 == tm * 16 + 4 * slot: the draws above)
                f32x4 yt = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float xv = s[r] + b;
                    const float p = act == ACT_SIGMOID ? sigmoidf_fast(xv) : (act == ACT_RELU ? fmaxf(xv, 0.f) : xv);
                    float y = p;
                    if (noise == NOISE_BERNOULLI) y = (u32_to_unit(w1[r]) < p) ? 1.f : 0.f;
                    else if (noise == NOISE_GAUSSIAN) y = p + box_muller(u32_to_unit(w1[r]), u32_to_unit(w2[r]));
                    if (row0 + r < a.rows) {
                        if (out) st_plane<OUT_SC>(out, (unsigned)((row0 + r) * ldo + col), y);
                        yt[r] = y;
                        if (RP == 1) rp[r] = softplusf(xv);
                        if (RP == 2) rp[r] = y * b;
                    }
                }
                if (outT) st_plane4(outT, 4u * (unsigned)(col * a.ldt + row0), yt);
            }
            if constexpr (RP != 0) {   // (every lane of the wave: the sums run over the 16 lanes of a row group)
#pragma unroll
                for (int r = 0; r < 4; ++r) rp[r] = row16_sum(rp[r]);
                if (x == 0) st_plane4(rowpart, 4u * (unsigned)(tn * a.ldt + row0), rp);
            }
        }
        KURBM_FST(3);
        __syncthreads();
        KURBM_FST(4);
    }
}

// h -> v: two tiles per pass (four waves each) where that saves a pass over the list of a dealer's slots -- n_rt row tiles of `units`
// column tiles over `cap` workgroups -- and k is short (nch chunks of 16)
__device__ __forceinline__ bool two_tiles_per_pass(int n_rt, int units, int cap, int nch) {
    const int one = n_rt * units, two = n_rt * ((units + 1) / 2);
    return nch <= 32 && (two + cap - 1) / cap < (one + cap - 1) / cap;
}

// (TWO tiles per pass where that saves a pass and k is short -- decided the same way by every workgroup of the grid)
template <bool HV, int IN_PL, int OUT_SC, bool LOCAL, int RP = 0>
__device__ __forceinline__ void half_step_small(const SmallArgs& a, const float* __restrict__ in, int ld_in, float* __restrict__ out, int ldo,
                                                float* __restrict__ outT, int act, int noise, const RngArgs& rng, float* red,
                                                float* __restrict__ rowpart = nullptr, int grp = 0) {
    if constexpr (HV) {
        const int tiles_m = (a.rows + 15) / 16, tiles_n = (a.n_vis + 15) / 16, nch = (a.n_hid + 15) / 16;
        const int n_rt = LOCAL ? (tiles_m - grp + 7) / 8 : 1;
        if (two_tiles_per_pass(n_rt, LOCAL ? tiles_n : tiles_m * tiles_n, LOCAL ? (int)gridDim.x >> 3 : (int)gridDim.x, nch)) {
            half_step_tiles<HV, IN_PL, OUT_SC, LOCAL, true, RP>(a, in, ld_in, out, ldo, outT, act, noise, rng, red, rowpart, grp);
            return;
        }
    }
    half_step_tiles<HV, IN_PL, OUT_SC, LOCAL, false, RP>(a, in, ld_in, out, ldo, outT, act, noise, rng, red, rowpart, grp);
}

// Phase 1 is done on every XCD (eight agent-scope words, each set by the first workgroup of a group behind the group's first
// barrier): the wait of a workgroup that takes the positive half of its phase-4 tile early.  Bounded; workgroup-uniform result.
__device__ __forceinline__ bool early_wait(const SmallArgs& a, unsigned* done1, unsigned target) {
    __shared__ int eok;
    if (threadIdx.x < 64) {
        const unsigned long long t0 = realtime_ticks();
        bool seen = false;
        for (;;) {
            const unsigned f = threadIdx.x < 8 ? __hip_atomic_load(done1 + 16 * threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : target;
            if (__all((int)(f - target) >= 0)) { seen = true; break; }
            __builtin_amdgcn_s_sleep(1);
            if (realtime_ticks() - t0 > a.timeout_ticks) break;
        }
        if (threadIdx.x == 0) eok = seen ? 1 : 0;
    }
    __syncthreads();
    const bool r = eok != 0;
    __syncthreads();                                   // (eok may be written again by a second wait)
    return r;
}

// v_pos^T . h_pos of the 16 x 16 tile of W that is phase 4's task `task` (rbm.py:125: the positive half of dW), k = the batch rows.
// NOT inlined (three call sites, two unroll variants each: inlined, the kernel ran out of registers), and handed plain values, not
// the argument struct (a struct passed by reference to a real call is kept in scratch).
__device__ __attribute__((noinline)) f32x4 positive_product_fn(const float* v, int ldv, int n_vis, const float* h_posT, int ldt, int n_hid,
                                                                int rows, int task, int tiles_h, int nch, int x, int slot) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int ti = task / tiles_h, tj = task - ti * tiles_h;
    const float* A = v + ti * 16;
    const float* B = h_posT + (size_t)tj * 16 * ldt;
    const bool a_ok = ti * 16 + x < n_vis, b_ok = tj * 16 + x < n_hid;
    if (nch <= 4) tile_mma<false, true, false, 0, 1, 4>(acc, A, ldv, a_ok, B, ldt, b_ok, rows, 0, nch, 1, x, slot);   // (batches up to 64 rows)
    else tile_mma<false, true, false, 0, 1>(acc, A, ldv, a_ok, B, ldt, b_ok, rows, 0, nch, 1, x, slot);
    return acc;
}
__device__ __forceinline__ void positive_product(f32x4& acc, const SmallArgs& a, int task, int tiles_h, int nch, int x, int slot) {
    acc = positive_product_fn(a.v, a.ldv, a.n_vis, a.h_posT, a.ldt, a.n_hid, a.rows, task, tiles_h, nch, x, slot);
}

#ifdef KURBM_SMALL_STAMPS   // diagnostic build: phase boundaries of workgroup 0 (100 MHz ticks) into words 40.. of the status block
#define KURBM_SST(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<unsigned long long*>(a.status + 40)[i] = realtime_ticks(); } while (0)
#else
#define KURBM_SST(i) do { } while (0)
#endif
template <bool LOCAL>
__global__ __launch_bounds__(64 * SMALL_WAVES) void k_cd1_small(SmallArgs a) {
    __shared__ __attribute__((aligned(16))) float red[SMALL_WAVES * 64 * 4];
    unsigned gen = 0;
    if (threadIdx.x == 0) gen = __hip_atomic_load(a.bar + 16 * 9, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    gen = __builtin_amdgcn_readfirstlane(gen);   // (only thread 0 uses it; uniform for tidiness)
    const int act_h = a.gauss ? ACT_RELU : ACT_SIGMOID, act_v = a.gauss ? ACT_LINEAR : ACT_SIGMOID;
    bool ok = true;
    // phase 4's tasks, one per wave: the 16 x 16 tiles of W, then 16-column groups of db_h and db_v.  LOCAL: dealt from the LAST
    // workgroup down -- the high ranks of every XCD's group have no tile in phases 1-3, so a wave there can take the positive half
    // of its tile's statistics (v_pos^T . h_pos: ready once phase 1 is done everywhere) while the others run phases 2 and 3
    const int lane = threadIdx.x & 63, x = lane & 15, slot = lane >> 4;
    const int tiles_v = (a.n_vis + 15) / 16, tiles_h = (a.n_hid + 15) / 16, nch = (a.rows + 15) / 16;
    const int n_w = (a.which & 1) ? tiles_v * tiles_h : 0, n_bh = (a.which & 2) ? tiles_h : 0, n_bv = (a.which & 4) ? tiles_v : 0;
    const int nwaves = gridDim.x * SMALL_WAVES;
    const int task0 = (LOCAL ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x) * SMALL_WAVES + (threadIdx.x >> 6);
    f32x4 acc_early = {0.f, 0.f, 0.f, 0.f};
    bool have_early = false;
    KURBM_SST(0);
    if constexpr (LOCAL) {
        // The `local` schedule: row tile tm of phases 1-3 belongs to the workgroups of XCD tm % 8, which hand h_pos and v_neg to
        // each other through their L2 and meet at barriers of their own; only the transposed planes -- phase 4's operands, read by
        // everybody -- leave at agent scope, and only phase 4 waits for the whole grid.
        const int grp = xcc_of_workgroup();
        unsigned ggen = 0;
        if (threadIdx.x == 0) {
            // this workgroup's own word: the number of the last barrier its group completed (every word of a group holds the same)
            const __amdgpu_buffer_rsrc_t d = __builtin_amdgcn_make_buffer_rsrc(a.bar + 160 + 32 * grp, 0, 0xFFFFFFFF, 0x00020000);
            ggen = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(d, 4 * (int)(blockIdx.x >> 3), 0, 2);
        }
        ggen = __shfl(ggen, 0);            // (wave 0 polls with it; the other waves do not use it)
        half_step_small<false, 0, 2, true>(a, a.v, a.ldv, a.h_pos, a.ldh, a.h_posT, act_h, NOISE_BERNOULLI, a.rng_h, red, nullptr, grp);
        KURBM_SST(1);
        ok = group_barrier(a, grp, ggen + 1u) && ok;
        KURBM_SST(2);
        unsigned* done1 = a.bar + 464;                     // eight words, 64 bytes apart: "phase 1 is done on XCD g"
        bool idle2 = false, idle3 = false;
        const bool wg_has_w_task = (int)(task0 - (threadIdx.x >> 6)) < n_w;     // (workgroup-uniform: its first wave's task is a tile of W)
        {
            // "phase 1 is done on XCD grp" for everybody (its transposed plane was drained at agent scope before the barrier)
            const int rank = blockIdx.x >> 3, ngrp = (int)gridDim.x >> 3;
            if (ok && rank == 0 && threadIdx.x == 0) __hip_atomic_store(done1 + 16 * grp, ggen + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // does this workgroup have a tile in phase 2?  (half_step_small's rule: two tiles per pass where that saves a pass)
            const int n_rt = ((a.rows + 15) / 16 - grp + 7) / 8;      // row tiles of this XCD: its phase-2 / phase-3 slots are dealt over the ranks
            const int two = two_tiles_per_pass(n_rt, tiles_v, ngrp, tiles_h) ? 2 : 1;
            idle2 = rank >= n_rt * ((tiles_v + two - 1) / two);
            idle3 = rank >= n_rt * tiles_h;
            if (ok && idle2 && wg_has_w_task) {
                ok = early_wait(a, done1, ggen + 1u);
                if (ok && task0 < n_w) { positive_product(acc_early, a, task0, tiles_h, nch, x, slot); have_early = true; }
            }
        }
        if (ok) half_step_small<true, 2, 2, true>(a, a.h_pos, a.ldh, a.v_neg, a.ldn, a.v_negT, act_v, a.gauss ? NOISE_GAUSSIAN : NOISE_BERNOULLI, a.rng_v, red, nullptr, grp);
        KURBM_SST(3);
        ok = group_barrier(a, grp, ggen + 2u) && ok;
        KURBM_SST(4);
        if (ok && !idle2 && idle3 && wg_has_w_task) {      // (busy in phase 2, idle in phase 3: the positive half now)
            ok = early_wait(a, done1, ggen + 1u);
            if (ok && task0 < n_w) { positive_product(acc_early, a, task0, tiles_h, nch, x, slot); have_early = true; }
        }
        if (ok) half_step_small<false, 2, 2, true>(a, a.v_neg, a.ldn, nullptr, 0, a.h_negT, ACT_SIGMOID, NOISE_NONE, a.rng_h, red, nullptr, grp);
        KURBM_SST(5);
        // In front of phase 4, not the grid's counter barrier (four memory-side trips in a row) but: the band's own barrier, a
        // "phase 3 done on XCD g" word from the group's first workgroup, and the workgroups that HAVE a task wait for the eight
        // words (one trip to publish, one per poll); the others are done.  Nobody writes W before every XCD is through phase 3.
        ok = group_barrier(a, grp, ggen + 3u) && ok;
        {
            unsigned* done3 = a.bar + 472;                 // (the other half of done1's 64-byte lines)
            if (ok && (blockIdx.x >> 3) == 0 && threadIdx.x == 0) __hip_atomic_store(done3 + 16 * grp, ggen + 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ok && (int)(task0 - (threadIdx.x >> 6)) >= n_w + n_bh + n_bv) return;     // (workgroup-uniform: no task for its first wave, none for the others)
            if (ok) ok = early_wait(a, done3, ggen + 3u);
        }
        (void)gen;
        KURBM_SST(6);
    } else {
        // 1: h_pos ~ p(h | v_pos)                                                   rbm.py:120
        half_step_small<false, 0, 1, false>(a, a.v, a.ldv, a.h_pos, a.ldh, a.h_posT, act_h, NOISE_BERNOULLI, a.rng_h, red);
        KURBM_SST(1);
        ok = grid_barrier(a, gen) && ok;
        KURBM_SST(2);
        // 2: v_neg ~ p(v | h_pos)                                                   rbm.py:121-123 / :143-144
        if (ok) half_step_small<true, 1, 1, false>(a, a.h_pos, a.ldh, a.v_neg, a.ldn, a.v_negT, act_v, a.gauss ? NOISE_GAUSSIAN : NOISE_BERNOULLI, a.rng_v, red);
        KURBM_SST(3);
        ok = grid_barrier(a, gen) && ok;
        KURBM_SST(4);
        // 3: h_neg = sigmoid(v_neg . W + b_h): probabilities in both modes          rbm.py:124 / :145
        if (ok) half_step_small<false, 1, 1, false>(a, a.v_neg, a.ldn, nullptr, 0, a.h_negT, ACT_SIGMOID, NOISE_NONE, a.rng_h, red);
        KURBM_SST(5);
        ok = grid_barrier(a, gen) && ok;
        KURBM_SST(6);
    }
    if (!ok) {
        if (threadIdx.x == 0) __hip_atomic_fetch_or(a.status, 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    // 4: the statistics, applied
    for (int task = task0; task < n_w + n_bh + n_bv; task += nwaves) {
        if (task < n_w) {
            const int ti = task / tiles_h, tj = task - ti * tiles_h;
            const int i = ti * 16 + x, j = tj * 16 + x;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const int col = tj * 16 + x, row0 = ti * 16 + 4 * slot;
            float wv[4];                                   // the tile of W itself: requested first, needed last
#pragma unroll
            for (int r = 0; r < 4; ++r) wv[r] = (col < a.n_hid && row0 + r < a.n_vis) ? a.W[(size_t)(row0 + r) * a.ldw + col] : 0.f;
            // dW = v_pos^T . h_pos - v_neg^T . h_neg   (rbm.py:125-126), k = the batch rows: contiguous in the transposed planes, strided
            // in the data.  (One batch of loads per product.  Both products' loads in ONE batch: 15.0 against 8.3 us for this phase.)
            if (have_early && task == task0) acc = acc_early;       // (the positive half was done beside phases 2 and 3: same MFMAs, same order)
            else positive_product(acc, a, task, tiles_h, nch, x, slot);
            {
                const float* An = a.v_negT + (size_t)ti * 16 * a.ldt;
                const float* Bn = a.h_negT + (size_t)tj * 16 * a.ldt;
                if (nch <= 4) tile_mma<true, true, true, 1, 1, 4>(acc, An, a.ldt, i < a.n_vis, Bn, a.ldt, j < a.n_hid, a.rows, 0, nch, 1, x, slot);
                else tile_mma<true, true, true, 1, 1>(acc, An, a.ldt, i < a.n_vis, Bn, a.ldt, j < a.n_hid, a.rows, 0, nch, 1, x, slot);
            }
            if (col < a.n_hid)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (row0 + r < a.n_vis) a.W[(size_t)(row0 + r) * a.ldw + col] = wv[r] + a.lr * acc[r];      // rbm.py:127-128
        } else {
            // a bias group: lane (column x, row group slot) adds rows 16 e + 4 slot .. + 3 of its column for e = 0, 1, ... (16-byte loads
            // from the transposed planes, whose rows past the batch are zeros; the data itself is row-major: dwords), then the four row
            // groups meet by shuffles, ((g0 + g1) + (g2 + g3)): a fixed order
            const bool hid = task < n_w + n_bh;
            const int g = hid ? task - n_w : task - n_w - n_bh;
            const int col = g * 16 + x, N = hid ? a.n_hid : a.n_vis;
            const float* posT = hid ? a.h_posT : nullptr;
            const float* negT = hid ? a.h_negT : a.v_negT;
            float s = 0.f;
            if (col < N)
                for (int r0 = 4 * slot; r0 < a.rows; r0 += 64) {
                    f32x4 pq[4], nq[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = r0 + 16 * e;
                        const bool live = r < a.rows;
                        nq[e] = live ? ld_plane4<1>(negT, 4u * (unsigned)(col * a.ldt + r)) : f32x4{0.f, 0.f, 0.f, 0.f};
                        if (hid) pq[e] = live ? ld_plane4<1>(posT, 4u * (unsigned)(col * a.ldt + r)) : f32x4{0.f, 0.f, 0.f, 0.f};
                        else
#pragma unroll
                            for (int q = 0; q < 4; ++q) pq[e][q] = (r + q < a.rows) ? a.v[(size_t)(r + q) * a.ldv + col] : 0.f;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int q = 0; q < 4; ++q) s += pq[e][q] - nq[e][q];
                }
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            if (slot == 0 && col < N) {
                float* b = hid ? a.b_h : a.b_v;
                b[col] += a.lr * s;                                                                    // rbm.py:130-134
            }
        }
    }
    KURBM_SST(7);
}

// ------------------------------------------------------------------------------------
// The per-step score of fit(verbose = 1), rbm.py:225-233, of a small RBM in ONE launch: mean_r |F(v_r) - F(v'_r)| with v' a fresh
// one-step reconstruction (sites rng_h, rng_v of the score's chain).  Three phases on the schedules of k_cd1_small --
//   1  h = sample of the hidden site from v          + the softplus row partials of v   (F's hidden term)
//   2  v' = sample of the visible site from h        + the row partials of v' . b_v
//   3  (v' . W + b_h: nothing stored)                + the softplus row partials of v'
// -- and no grid-wide barrier: a workgroup that is done counts itself in, and the LAST one adds the partials up (fixed order),
// takes the mean and leaves.  v . b_v of the data is computed beside phase 1 by a workgroup with nothing else to do there.
// Row partials: rp1 [tiles_h][ldt], rpv [tiles_v][ldt], rp2 [tiles_h][ldt], vb [ldt] in the workspace (agent scope).
template <bool LOCAL>
__global__ __launch_bounds__(64 * SMALL_WAVES) void k_score_small(SmallArgs a) {
    __shared__ __attribute__((aligned(16))) float red[SMALL_WAVES * 64 * 4];
    __shared__ int last;
    const int act_h = a.gauss ? ACT_RELU : ACT_SIGMOID, act_v = a.gauss ? ACT_LINEAR : ACT_SIGMOID;
    const int tiles_h = (a.n_hid + 15) / 16, tiles_v = (a.n_vis + 15) / 16, tiles_m = (a.rows + 15) / 16;
    float* rp1 = a.h_posT;                              // (the transposed planes' space: nothing of a step is alive here)
    float* rpv = rp1 + (size_t)tiles_h * a.ldt;
    float* rp2 = rpv + (size_t)tiles_v * a.ldt;
    float* vb = rp2 + (size_t)tiles_h * a.ldt;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // v . b_v of the data, a wave per row: LOCAL -- the last workgroup of every XCD's group, for the rows of that group's row tiles
    // (it has no tile in phase 1: at most 16 column tiles there); else the grid's last workgroup, every row
    auto data_dot = [&](int r) {
        float t = 0.f;
        for (int c = lane; c < a.n_vis; c += 64) t += a.v[(size_t)r * a.ldv + c] * a.b_v[c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
        if (lane == 0) st_plane<1>(vb, (unsigned)r, t);
    };
    bool ok = true;
    if constexpr (LOCAL) {
        unsigned ggen = 0;
        const int grp = xcc_of_workgroup(), rank = blockIdx.x >> 3, ngrp = (int)gridDim.x >> 3;
        if (threadIdx.x == 0) {
            const __amdgpu_buffer_rsrc_t d = __builtin_amdgcn_make_buffer_rsrc(a.bar + 160 + 32 * grp, 0, 0xFFFFFFFF, 0x00020000);
            ggen = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(d, 4 * rank, 0, 2);
        }
        ggen = __shfl(ggen, 0);
        KURBM_SST(0);
        half_step_small<false, 0, 2, true, 1>(a, a.v, a.ldv, a.h_pos, a.ldh, nullptr, act_h, NOISE_BERNOULLI, a.rng_h, red, rp1, grp);
        if (rank == ngrp - 1)
            for (int tm = grp; tm < tiles_m; tm += 8)
                for (int r = tm * 16 + wave; r < tm * 16 + 16 && r < a.rows; r += SMALL_WAVES) data_dot(r);
        KURBM_SST(1);
        ok = group_barrier(a, grp, ggen + 1u) && ok;
        KURBM_SST(2);
        if (ok) half_step_small<true, 2, 2, true, 2>(a, a.h_pos, a.ldh, a.v_neg, a.ldn, nullptr, act_v, a.gauss ? NOISE_GAUSSIAN : NOISE_BERNOULLI, a.rng_v, red, rpv, grp);
        KURBM_SST(3);
        ok = group_barrier(a, grp, ggen + 2u) && ok;
        KURBM_SST(4);
        if (ok) half_step_small<false, 2, 2, true, 1>(a, a.v_neg, a.ldn, nullptr, 0, nullptr, ACT_LINEAR, NOISE_NONE, a.rng_h, red, rp2, grp);
        KURBM_SST(5);
    } else {
        unsigned gen = 0;
        if (threadIdx.x == 0) gen = __hip_atomic_load(a.bar + 16 * 9, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        gen = __builtin_amdgcn_readfirstlane(gen);
        half_step_small<false, 0, 1, false, 1>(a, a.v, a.ldv, a.h_pos, a.ldh, nullptr, act_h, NOISE_BERNOULLI, a.rng_h, red, rp1);
        if (blockIdx.x == gridDim.x - 1)
            for (int r = wave; r < a.rows; r += SMALL_WAVES) data_dot(r);
        ok = grid_barrier(a, gen) && ok;
        if (ok) half_step_small<true, 1, 1, false, 2>(a, a.h_pos, a.ldh, a.v_neg, a.ldn, nullptr, act_v, a.gauss ? NOISE_GAUSSIAN : NOISE_BERNOULLI, a.rng_v, red, rpv);
        ok = grid_barrier(a, gen) && ok;
        if (ok) half_step_small<false, 1, 1, false, 1>(a, a.v_neg, a.ldn, nullptr, 0, nullptr, ACT_LINEAR, NOISE_NONE, a.rng_h, red, rp2);
    }
    if (!ok && threadIdx.x == 0) __hip_atomic_fetch_or(a.status, 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // count this workgroup in (its partials have drained); the last one finishes
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned* cnt = a.bar + 448;
        const unsigned prev = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = prev == gridDim.x - 1u;
        if (last) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
#ifdef KURBM_SMALL_STAMPS   // (the last workgroup's own stamps: when it found itself last, when the score was stored)
    if (last && threadIdx.x == 0) reinterpret_cast<unsigned long long*>(a.status + 40)[6] = realtime_ticks();
#endif
    if (!last) return;
    // |F(v) - F(v')| of every row, four lanes per row (lane q of a quad takes the partials of column tiles q, q + 4, ...: sixteen
    // requested at a time -- one at a time, one thread per row, this was 65 memory-side round trips in a row, ~40 us), the quad's
    // four sums added as (s0 + s1) + (s2 + s3); then a fixed-order tree over the rows
    float* sums = red;                                  // 512 floats of the 2048
    float d = 0.f;
    const int part = threadIdx.x & 3;
    // (the first sixteen partials of all three sums -- all of them up to 1024 units a side -- and v . b_v are requested together: one
    //  memory-side round trip instead of four in a row)
    auto first16 = [&](float (&q)[16], const float* part_sums, int ntile, int r, bool live) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 16; ++i) q[i] = (live && part + 4 * i < ntile) ? ld_plane<1>(part_sums, (unsigned)((part + 4 * i) * a.ldt + r)) : 0.f;
    };
    auto finish = [&](const float (&q)[16], const float* part_sums, int ntile, int r, bool live) __attribute__((always_inline)) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc += q[i];
        for (int t0 = part + 64; t0 < ntile; t0 += 64) {      // (more than 64 column tiles: the rest, sixteen at a time)
            float q2[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) q2[i] = (live && t0 + 4 * i < ntile) ? ld_plane<1>(part_sums, (unsigned)((t0 + 4 * i) * a.ldt + r)) : 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc += q2[i];
        }
        acc += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(acc), 0xB1, 0xF, 0xF, true));   // lane ^ 1
        acc += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(acc), 0x4E, 0xF, 0xF, true));   // lane ^ 2
        return acc;
    };
    for (int r0 = 0; r0 < a.rows; r0 += 16 * SMALL_WAVES) {          // (every lane of a quad makes the same trips)
        const int r = r0 + ((int)threadIdx.x >> 2);
        const bool live = r < a.rows;
        float qa[16], qb[16], qc[16];
        first16(qa, rp1, tiles_h, r, live); first16(qb, rpv, tiles_v, r, live); first16(qc, rp2, tiles_h, r, live);
        const float dot = (live && part == 0) ? ld_plane<1>(vb, (unsigned)r) : 0.f;
        const float fh = finish(qa, rp1, tiles_h, r, live), fv1 = finish(qb, rpv, tiles_v, r, live), fh1 = finish(qc, rp2, tiles_h, r, live);
        if (live && part == 0) {
            const float f = dot + fh, f1 = fv1 + fh1;
            if (a.F) { a.F[r] = -f; a.F[a.rows + r] = -f1; }
            d += fabsf(f1 - f);                             // |(-f) - (-f1)|
        }
    }
    sums[threadIdx.x] = d;
    __syncthreads();
    for (int w = 32 * SMALL_WAVES; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sums[threadIdx.x] += sums[threadIdx.x + w];
        __syncthreads();
    }
    // the score and a 1.0f behind it as ONE 8-byte store at system scope: `score` may be pinned host memory, which the host
    // polls for that second word instead of synchronising with the device (kurbm_cd_epoch_small_scored)
    if (threadIdx.x == 0) {
        const float sc = ok ? sums[0] / (float)a.rows : __builtin_nanf("");
        const unsigned long long both = (unsigned long long)__float_as_uint(sc) | ((unsigned long long)__float_as_uint(1.0f) << 32);
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(a.score), both, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#ifdef KURBM_SMALL_STAMPS
        reinterpret_cast<unsigned long long*>(a.status + 40)[7] = realtime_ticks();
#endif
    }
}

hipError_t launch_score_small(const SmallArgs& a, int nblk, hipStream_t st) {
    if (a.local) hipLaunchKernelGGL(k_score_small<true>, dim3(nblk), dim3(64 * SMALL_WAVES), 0, st, a);
    else hipLaunchKernelGGL(k_score_small<false>, dim3(nblk), dim3(64 * SMALL_WAVES), 0, st, a);
    return hipGetLastError();
}

// which XCD does workgroup i of a grid run on?  (kurbm_ctx_create: the `local` schedule needs i -> i % 8, up to a renaming of XCDs)
__global__ void k_xcc_probe(unsigned* out) {
    unsigned xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) out[blockIdx.x] = xcc & 15u;
}
hipError_t launch_xcc_probe(unsigned* out, int nblk, hipStream_t st) {
    hipLaunchKernelGGL(k_xcc_probe, dim3(nblk), dim3(64 * SMALL_WAVES), 0, st, out);
    return hipGetLastError();
}

hipError_t launch_cd1_small(const SmallArgs& a, int nblk, hipStream_t st) {
    if (a.local) hipLaunchKernelGGL(k_cd1_small<true>, dim3(nblk), dim3(64 * SMALL_WAVES), 0, st, a);
    else hipLaunchKernelGGL(k_cd1_small<false>, dim3(nblk), dim3(64 * SMALL_WAVES), 0, st, a);
    return hipGetLastError();
}

}  // namespace kurbm
