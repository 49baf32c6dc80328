// kurbm_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the RBM contrastive-divergence path.
//
// One templated fp32 MFMA GEMM core (v_mfma_f32_16x16x4_f32, exact f32 fma chains) with
// three operand-layout variants and fused epilogues:
//
//   half step v->h   x = v.W + b_h      A = v  [row][k] (k contiguous)   B = W [k][n]
//   half step h->v   x = h.W^T + b_v    A = h  [row][k]                  B = W [n][k]  (W read
//                                                                        as stored, never transposed)
//   statistics       dW = v+^T.h+ - v-^T.h-   A = v [k][m], B = h [k][n], K = batch, two signed
//                                       segments, split over the batch into fp32 slabs
//
// Reference op sequences replaced (reference ku/ebm/rbm.py): :46-47 / :82-83 (v->h),
// :52-53 / :121-123 (h->v), :124 (h_neg), :125-134 (updates), :73-75 (free energy).
//
// Design notes (DESIGN.md has the long form):
//   * 256 threads = 4 waves per workgroup; every wave owns a WM x WN output tile as TM x TN
//     accumulators of 16x16 (4 VGPRs each).  The host side sizes tiles and split-K so that two
//     workgroups share a CU whenever the problem allows it.
//   * K is consumed in tiles of 32.  Operands whose k index is contiguous in memory are
//     staged "x-major" ([x][36]); a lane fetches 4 consecutive k with one ds_read_b128 and
//     feeds them to 4 successive MFMAs.  The MFMA sums over k, so any pairing of k between
//     the four k-slots of a 16x16x4 is valid as long as A and B agree: both use
//     k(r, slot, t) = 16 r + 4 slot + t.  Operands whose k index is the slow one are staged
//     "k-major" ([k][BX+4]) and read with ds_read_b32 at that same k.
//   * Global -> register -> LDS staging, two LDS buffers, ONE barrier per k-tile.  A tile is 8 MFMA
//     groups; tile t+1 is fetched in groups 0-3, parked in groups 4-6, and the next tile's first
//     fragments are read in group 7, all pinned between the MFMAs with sched_barrier(0).
//   * The epilogue runs in the accumulator layout, then every output plane is transposed through a
//     per-wave LDS patch so that global memory sees whole rows (16 B per lane).
//   * The Philox block of an output element is keyed by (col, row >> 2): its four words are
//     the four rows a lane holds in one 16x16 accumulator, so one Philox call serves one
//     accumulator and no lane computes a block it does not use.
//   * Column sums needed by the bias updates are produced deterministically as per-row-tile
//     partials by the epilogue (no atomics anywhere: results are bit-reproducible).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "kurbm_kernels.h"
#include "kurbm_device.h"

namespace kurbm {

typedef uint32_t u32x4s __attribute__((ext_vector_type(4)));

// Diagnostic build only (-DKURBM_STAMPS, libkurbm_stamps.so): s_memtime brackets around the
// prologue / k loop / epilogue and the 8 MFMA groups of a tile, written to GemmArgs::stamps (16 x u64 per wave).  The shipped library has no stamps.
#ifdef KURBM_STAMPS
#define KURBM_STAMP(var)                                                                        \
    do {                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                      \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");             \
        __builtin_amdgcn_sched_barrier(0);                                                      \
    } while (0)
#define KURBM_ON(bit) (!(g.dbg_off & (bit)))   /* ablation switches of the diagnostic build */
#else
#define KURBM_STAMP(var) do { } while (0)
#define KURBM_ON(bit) true
#endif

constexpr int BK = 32;          // k-tile
constexpr int NTHREADS = 256;   // 4 waves
constexpr int LDX = BK + 4;     // x-major LDS row (floats): 144 B, 16-B aligned, conflict-free b128

__global__ void k_philox_uniform(float* __restrict__ out, int rows, int cols, int ld, RngArgs rng) {
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    const int rg = blockIdx.y;  // group of 4 rows
    if (col >= cols) return;
    const uint64_t grow = rng.row0 + (uint64_t)rg * 4;
    uint32_t w[4];
    philox4x32_10((uint32_t)col, (uint32_t)(grow >> 2), rng.stream_id, rng.step, rng.seed_lo, rng.seed_hi, w);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = rg * 4 + r;
        if (row < rows) out[(size_t)row * ld + col] = u32_to_unit(w[r]);
    }
}

// ------------------------------------------------------------------------------------
// operand staging: global -> registers -> LDS
// ------------------------------------------------------------------------------------
template <int BX, bool KMAJOR>
struct Stage {
    static constexpr int LDK = BX + 4;  // k-major LDS row (floats); (BX+4) % 8 == 4 for BX in {64,112,128}
    static constexpr int CHUNKS = BX * BK / 4;   // 16-B chunks per tile
    static constexpr int ITERS = (CHUNKS + NTHREADS - 1) / NTHREADS;
    static constexpr int TILE_FLOATS = KMAJOR ? BK * LDK : BX * LDX;
    static constexpr bool RAGGED = (CHUNKS % NTHREADS) != 0;   // last chunk only for some lanes

    struct Regs { f32x4 r[ITERS]; };   // one tile in flight, 16 B per chunk

    unsigned goff[ITERS];    // per-lane element offset of each chunk from (src + tile origin)
    int soff[ITERS];         // per-lane float offset of each chunk in the LDS tile
    int tid;

    // chunk q of the tile -> (k, x) of its first element; consecutive lanes walk the
    // contiguous memory direction, so a wave reads whole 128-B (x-major) / 512-B (k-major) runs
    static __device__ __forceinline__ void coords(int q, int& kk, int& xx) {
        if (KMAJOR) { kk = q / (BX / 4); xx = 4 * (q % (BX / 4)); }
        else        { xx = q / (BK / 4); kk = 4 * (q % (BK / 4)); }
    }

    // Per-lane offsets for the full-tile fast path.  Chunks whose x lies outside [0, X) are
    // pointed at x = 0: they only ever feed output rows/columns the epilogue never stores.
    __device__ __forceinline__ void init(int ld, int x0, int X, int tid_) {
        tid = tid_;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int q = it * NTHREADS + tid;
            int kk, xx;
            coords(q, kk, xx);
            const int x = (x0 + xx < X) ? x0 + xx : 0;
            goff[it] = KMAJOR ? (unsigned)(kk * ld + x) : (unsigned)(x * ld + kk);
            soff[it] = KMAJOR ? kk * LDK + xx : xx * LDX + kk;
        }
    }

    __device__ __forceinline__ bool lane_has(int it) const { return !RAGGED || it * NTHREADS + tid < CHUNKS; }

    // Fast path (tiles whose 32 k are all valid): one chunk, no masks, no address arithmetic
    // beyond origin + offset.  `origin` = src + k0 (x-major) or src + k0 * ld (k-major), uniform.
    __device__ __forceinline__ void load_chunk(Regs& R, const float* __restrict__ origin, int it) const {
        if (lane_has(it)) R.r[it] = *reinterpret_cast<const f32x4*>(origin + goff[it]);
    }
    template <bool SIGNED>
    __device__ __forceinline__ void park_chunk(const Regs& R, float* __restrict__ s, float sgn, int it) const {
        if (lane_has(it)) *reinterpret_cast<f32x4*>(s + soff[it]) = SIGNED ? R.r[it] * sgn : R.r[it];
    }

    // Masked path for the k-tail tile.  Branch-free loads: out-of-range chunks read element 0
    // of the matrix (always mapped) and are zeroed in park_masked(); a conditional load makes
    // hipcc wait vmcnt(0) right behind every load, which serialises the tile's fetches.
    __device__ __forceinline__ void load_masked(Regs& R, const float* __restrict__ src, int ld, int x0, int X,
                                                int k0, int kend) const {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int q = it * NTHREADS + tid;
            int kk, xx;
            coords(q, kk, xx);
            const int k = k0 + kk, x = x0 + xx;
            const bool ok = (k < kend) & (x < X) & (!RAGGED | (q < CHUNKS));
            const size_t off = KMAJOR ? ((size_t)k * ld + x) : ((size_t)x * ld + k);
            R.r[it] = *reinterpret_cast<const f32x4*>(src + (ok ? off : (size_t)0));
        }
    }

    // Zero what lies outside [0,X) x [0,kend), apply the segment sign, write the LDS tile.
    __device__ __forceinline__ void park_masked(const Regs& R, float* __restrict__ s, int x0, int X, int k0,
                                                int kend, float sgn) const {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int q = it * NTHREADS + tid;
            int kk, xx;
            coords(q, kk, xx);
            const int k = k0 + kk, x = x0 + xx;
            const bool ok = (k < kend) & (x < X);
            const int lim = KMAJOR ? (X - x) : (kend - k);  // valid elements along the contiguous direction
            f32x4 v = R.r[it];
            v.x = ok ? v.x * sgn : 0.f;
            v.y = (ok & (lim > 1)) ? v.y * sgn : 0.f;
            v.z = (ok & (lim > 2)) ? v.z * sgn : 0.f;
            v.w = (ok & (lim > 3)) ? v.w * sgn : 0.f;
            if (!RAGGED || q < CHUNKS)
                *reinterpret_cast<f32x4*>(s + (KMAJOR ? kk * LDK + xx : xx * LDX + kk)) = v;
        }
    }
};

// Fragment fetch for MFMA 16x16x4: element e (0..3) of frag[i] is the operand of the MFMA
// at k-step (r, e): k = 16 r + 4 (lane >> 4) + e, row/col = x_base + 16 i + (lane & 15).
template <int BX, bool KMAJOR, int T>
__device__ __forceinline__ void fetch_frags(const float* __restrict__ s, int xw, int r, int lane,
                                            f32x4 (&frag)[T]) {
    const int l15 = lane & 15, slot = lane >> 4;
    if (KMAJOR) {
        constexpr int LDK = BX + 4;
#pragma unroll
        for (int i = 0; i < T; ++i) {
            const float* p = s + (16 * r + 4 * slot) * LDK + xw + 16 * i + l15;
            frag[i].x = p[0];
            frag[i].y = p[LDK];
            frag[i].z = p[2 * LDK];
            frag[i].w = p[3 * LDK];
        }
    } else {
#pragma unroll
        for (int i = 0; i < T; ++i)
            frag[i] = *reinterpret_cast<const f32x4*>(s + (xw + 16 * i + l15) * LDX + 16 * r + 4 * slot);
    }
}

// ------------------------------------------------------------------------------------
// the GEMM kernel
// ------------------------------------------------------------------------------------
template <int BM, int BN, int WAVES_M, int WAVES_N, bool A_KM, bool B_KM, int EPI, int NOISE>
__global__ __launch_bounds__(NTHREADS) void k_gemm(GemmArgs g) {
    warm_kernel_arguments<sizeof(GemmArgs)>();   // (kurbm_device.h: one wait for the argument segment, not one per use)
    static_assert(WAVES_M * WAVES_N == 4, "4 waves");
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int TM = WM / 16, TN = WN / 16;
    static_assert(WM % 16 == 0 && WN % 16 == 0, "wave tile");
    using StA = Stage<BM, A_KM>;
    using StB = Stage<BN, B_KM>;
    constexpr int A_FL = StA::TILE_FLOATS, B_FL = StB::TILE_FLOATS;
    // epilogue scratch: per-wave transpose patch / column-sum staging (half step), row-sum staging (softplus)
    constexpr int EPI_FL = (EPI == EPI_HALFSTEP)
                               ? ((4 * WM * (WN + 4) > WAVES_M * (64 / (WN / 4)) * BN) ? 4 * WM * (WN + 4) : WAVES_M * (64 / (WN / 4)) * BN)
                               : ((EPI == EPI_SOFTPLUS) ? WAVES_N * BM : 0);
    constexpr int SMEM_FL = (2 * (A_FL + B_FL) > EPI_FL) ? 2 * (A_FL + B_FL) : EPI_FL;
    __shared__ __attribute__((aligned(16))) float smem[SMEM_FL];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    // ---- block -> (tile, k-slice): XCD-aware bijective remap (blocks b and b+8 share an XCD,
    // so give each XCD a contiguous run of tiles: neighbours share the A row panel in its L2)
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, q = nwg >> 3, rr = nwg & 7;
        bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
    }
    const int tiles_mn = g.grid_m * g.grid_n;
    // tile_major (statistics GEMM): all k-slices of an output tile are neighbours in the remapped
    // order, i.e. run on ONE XCD, and the reduce kernel visits tiles in the same order -- the slabs of
    // a tile are then written and read back through the same L2 instead of crossing the fabric
    const int z = g.tile_major ? bid % g.nsplit : bid / tiles_mn;
    const int tmn = g.tile_major ? bid / g.nsplit : bid - z * tiles_mn;
    // walk the SHORTER grid dimension fastest: an XCD's contiguous run of tiles then spans the whole
    // short dimension and a slice of the long one, which minimises the operand bytes its L2 must hold
    const int bm = g.m_fastest ? tmn % g.grid_m : tmn / g.grid_n;
    const int bn = g.m_fastest ? tmn / g.grid_m : tmn - bm * g.grid_n;
    const int m0 = bm * BM, n0 = bn * BN;

    // k space: `nseg` segments of extent K; each has nfull = K / 32 full tiles (+ one partial
    // "tail" tile when K % 32 != 0).  The full tiles of all segments form one list that is split
    // over the z slices; the last slice also takes the tail tiles.
    const int nfull = g.nkt;
    const int t_begin = z * g.kt_per_split;
    int t_end = t_begin + g.kt_per_split;
    if (t_end > g.kt_total) t_end = g.kt_total;
    const int nt = t_end > t_begin ? t_end - t_begin : 0;
    const bool do_tail = (g.K & (BK - 1)) != 0 && z == g.nsplit - 1;

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    StA sa;
    StB sb;
    sa.init(g.lda, m0, g.M, tid);
    sb.init(g.ldb, n0, g.N, tid);
    typename StA::Regs ra;   // the tile in flight between global memory and LDS
    typename StB::Regs rb;
    float* sA0 = smem;
    float* sB0 = smem + 2 * A_FL;
    constexpr bool SIGNED = (EPI == EPI_SLAB);   // only the statistics GEMM has a negative segment
    constexpr int NA = StA::ITERS, NB = StB::ITERS, NCH = NA + NB;
    static_assert(NCH <= 8, "at most two fetches per MFMA group");

    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0;
    (void)ts0; (void)ts1; (void)ts2; (void)ts3;
#ifdef KURBM_STAMPS
    unsigned long long seg_grp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    KURBM_STAMP(ts0);

    // full tile t (clamped to the slice) -> operand origins and sign
    auto tile_of = [&](int t, const float*& oa, const float*& ob, float& sgn) {
        t = t < t_end ? t : t_end - 1;
        const int seg = (t >= nfull) ? 1 : 0;
        const size_t k0 = (size_t)(t - seg * nfull) * BK;
        oa = (seg ? g.A1 : g.A0) + (A_KM ? k0 * g.lda : k0);
        ob = (seg ? g.B1 : g.B0) + (B_KM ? k0 * g.ldb : k0);
        sgn = seg ? -1.0f : 1.0f;
    };
    auto mfma_group = [&](const f32x4 (&fa)[TM], const f32x4 (&fb)[TN], int e) {
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[mi][e], fb[ni][e], acc[mi][ni], 0, 0, 0);
    };

    f32x4 fa0[TM], fb0[TN], fa1[TM], fb1[TN];   // fragments of the current tile: k 0..15 and k 16..31

    if (nt > 0) {
        // fill: tile 0 -> LDS buffer 0, then its first fragments
        const float *oa, *ob;
        float sgn;
        tile_of(t_begin, oa, ob, sgn);
#pragma unroll
        for (int c = 0; c < NA; ++c) sa.load_chunk(ra, oa, c);
#pragma unroll
        for (int c = 0; c < NB; ++c) sb.load_chunk(rb, ob, c);
#pragma unroll
        for (int c = 0; c < NA; ++c) sa.template park_chunk<SIGNED>(ra, sA0, sgn, c);
#pragma unroll
        for (int c = 0; c < NB; ++c) sb.template park_chunk<false>(rb, sB0, 1.0f, c);
        __syncthreads();
        fetch_frags<BM, A_KM, TM>(sA0, wm * WM, 0, lane, fa0);
        fetch_frags<BN, B_KM, TN>(sB0, wn * WN, 0, lane, fb0);
        KURBM_STAMP(ts1);

        // One k-tile = 8 MFMA groups (one per k-step of 4), software pipelined:
        //   groups 0-3  fetch tile i+1 from global memory into registers (two 16-B chunks per group)
        //               and, in group 1, read the k 16..31 fragments of the current LDS buffer
        //   groups 4-6  park tile i+1 in the other LDS buffer (its loads are >= 3 groups old)
        //   end of 6    the tile's ONLY barrier: every wave has read all its fragments of the current
        //               buffer and written its share of the next one
        //   group 7     read the next tile's k 0..15 fragments (fa0/fb0 are dead since group 3)
        // sched_barrier(0) pins this order; inside a group hipcc schedules freely.  Past the end of
        // the slice the loop re-fetches / re-parks the last tile (in bounds, never read), so the
        // body is branch-free.
        int t_fetch = t_begin + 1;            // the tile fetched (and parked) during the current one
        tile_of(t_fetch, oa, ob, sgn);
        auto one_tile = [&](const int cur) {  // cur is a literal at both call sites: LDS offsets fold
            const float* cA = sA0 + cur * A_FL;
            const float* cB = sB0 + cur * B_FL;
            float* nA = sA0 + (cur ^ 1) * A_FL;
            float* nB = sB0 + (cur ^ 1) * B_FL;
#ifdef KURBM_STAMPS
            unsigned long long tg[9];
#endif
#pragma unroll
            for (int grp = 0; grp < 8; ++grp) {
                __builtin_amdgcn_sched_barrier(0);
#ifdef KURBM_STAMPS
                KURBM_STAMP(tg[grp]);
#endif
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    if (c / 2 == grp && KURBM_ON(1)) {           // fetch
                        if (c < NA) sa.load_chunk(ra, oa, c);
                        else sb.load_chunk(rb, ob, c - NA);
                    }
                    if (4 + (c * 3) / NCH == grp && KURBM_ON(2)) {   // park, 3+ groups after the fetch
                        if (c < NA) sa.template park_chunk<SIGNED>(ra, nA, sgn, c);
                        else sb.template park_chunk<false>(rb, nB, 1.0f, c - NA);
                    }
                }
                if (grp == 1 && KURBM_ON(4)) {
                    fetch_frags<BM, A_KM, TM>(cA, wm * WM, 1, lane, fa1);
                    fetch_frags<BN, B_KM, TN>(cB, wn * WN, 1, lane, fb1);
                }
                if (grp == 7) {
                    if (KURBM_ON(4)) {
                        fetch_frags<BM, A_KM, TM>(nA, wm * WM, 0, lane, fa0);
                        fetch_frags<BN, B_KM, TN>(nB, wn * WN, 0, lane, fb0);
                    }
                    // scalar bookkeeping for the NEXT tile's fetches rides under this group's MFMAs
                    ++t_fetch;
                    tile_of(t_fetch, oa, ob, sgn);
                }
                if (grp < 4) mfma_group(fa0, fb0, grp);
                else mfma_group(fa1, fb1, grp - 4);
                if (grp == 6 && KURBM_ON(8)) __syncthreads();
            }
            __builtin_amdgcn_sched_barrier(0);
#ifdef KURBM_STAMPS
            KURBM_STAMP(tg[8]);
#pragma unroll
            for (int q = 0; q < 8; ++q) seg_grp[q] += tg[q + 1] - tg[q];
#endif
        };
        int i = 0;
        for (; i + 1 < nt; i += 2) {
            one_tile(0);
            one_tile(1);
        }
        if (i < nt) one_tile(0);
    }
    if (do_tail) {
        // the partial k-tile of each segment: masked, unpipelined (at most two tiles per launch)
        for (int seg = 0; seg < g.nseg; ++seg) {
            const int k0 = nfull * BK;
            const float sgn = seg ? -1.0f : 1.0f;
            __syncthreads();
            sa.load_masked(ra, seg ? g.A1 : g.A0, g.lda, m0, g.M, k0, g.K);
            sb.load_masked(rb, seg ? g.B1 : g.B0, g.ldb, n0, g.N, k0, g.K);
            sa.park_masked(ra, sA0, m0, g.M, k0, g.K, sgn);
            sb.park_masked(rb, sB0, n0, g.N, k0, g.K, 1.0f);
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                if (r == 1 && g.K - k0 <= 16) break;   // k 16..31 of the tile are all padding
                fetch_frags<BM, A_KM, TM>(sA0, wm * WM, r, lane, fa1);
                fetch_frags<BN, B_KM, TN>(sB0, wn * WN, r, lane, fb1);
#pragma unroll
                for (int e = 0; e < 4; ++e) mfma_group(fa1, fb1, e);
            }
        }
    }
    __syncthreads();   // LDS is reused by the epilogues
    KURBM_STAMP(ts2);
#ifdef KURBM_STAMPS
#define KURBM_STAMP_OUT()                                                                        \
    do {                                                                                         \
        KURBM_STAMP(ts3);                                                                        \
        if (g.stamps && lane == 0) {                                                             \
            unsigned long long* o = g.stamps + ((size_t)blockIdx.x * 4 + wave) * 16;             \
            o[0] = ts1 - ts0; o[1] = ts2 - ts1; o[2] = ts3 - ts2; o[3] = ts0;                    \
            for (int q = 0; q < 8; ++q) o[4 + q] = seg_grp[q];                                   \
        }                                                                                        \
    } while (0)
#else
#define KURBM_STAMP_OUT() do { } while (0)
#endif

    const int l15 = lane & 15, slot = lane >> 4;

    // ---------------- epilogue: raw partial sums to a slab (statistics GEMM) ------------
    if (EPI == EPI_SLAB) {
        float* slab = g.slab + (size_t)z * g.slab_stride;
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                const int col = n0 + wn * WN + ni * 16 + l15;
                const int rowb = m0 + wm * WM + mi * 16 + slot * 4;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (rowb + r < g.M && col < g.N) slab[(size_t)(rowb + r) * g.ld_slab + col] = acc[mi][ni][r];
            }
        KURBM_STAMP_OUT();
        return;
    }

    // ---------------- epilogue: bias + activation + draw (+ column-difference partials) -
    // The draw happens in the accumulator layout (one Philox block = the 4 rows a lane owns in one
    // 16x16 accumulator).  Every output plane is then transposed through the wave's private LDS
    // patch so global memory sees whole rows: 16 B per lane, 4*WN contiguous bytes per row, instead
    // of 64-B fragments -- and the reference plane of the column-difference sums is read the same way.
    if (EPI == EPI_HALFSTEP) {
        constexpr int LDE = WN + 4;                 // patch row (floats): +4 keeps b32 writes conflict-free
        constexpr int LPR = WN / 4;                 // lanes per output row (16 B each)
        constexpr int RPI = 64 / LPR;               // rows per wave-instruction
        constexpr int NPASS = (WM + RPI - 1) / RPI;
        float* patch = smem + wave * (WM * LDE);
        const bool want_diff = (g.ref != nullptr);

        float pv[TM][TN][4], sv[TM][TN][4], uv[TM][TN][4];
        // the activation is a launch-wide constant: branch on it ONCE, around the whole element nest
        auto elementwise = [&](auto act_tag) {
            constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                const int col = n0 + wn * WN + ni * 16 + l15;
                const float bias = (col < g.N) ? g.bias[col] : 0.f;
#pragma unroll
                for (int mi = 0; mi < TM; ++mi) {
                    const int rowb = m0 + wm * WM + mi * 16 + slot * 4;
                    uint32_t w[4] = {0u, 0u, 0u, 0u}, w2[4] = {0u, 0u, 0u, 0u};
                    if (NOISE != NOISE_NONE) {
                        const uint64_t grow = g.rng.row0 + (uint64_t)rowb;
                        philox4x32_10((uint32_t)col, (uint32_t)(grow >> 2), g.rng.stream_id, g.rng.step,
                                      g.rng.seed_lo, g.rng.seed_hi, w);
                        if (NOISE == NOISE_GAUSSIAN)
                            philox4x32_10((uint32_t)col, (uint32_t)(grow >> 2), g.rng.stream_id | 0x80000000u,
                                          g.rng.step, g.rng.seed_lo, g.rng.seed_hi, w2);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float x = acc[mi][ni][r] + bias;
                        float p;
                        if (ACT == ACT_SIGMOID) p = sigmoidf_fast(x);
                        else if (ACT == ACT_RELU) p = fmaxf(x, 0.f);
                        else p = x;
                        float sm = p;
                        const float ua = u32_to_unit(w[r]);
                        if (NOISE == NOISE_BERNOULLI) {
                            sm = (ua < p) ? 1.0f : 0.0f;
                        } else if (NOISE == NOISE_GAUSSIAN) {
                            const float ub = u32_to_unit(w2[r]);
                            sm = p + box_muller(ua, ub);
                        }
                        pv[mi][ni][r] = p; sv[mi][ni][r] = sm; uv[mi][ni][r] = ua;
                    }
                }
            }
        };
        if (g.act == ACT_SIGMOID) elementwise(std::integral_constant<int, ACT_SIGMOID>{});
        else if (g.act == ACT_RELU) elementwise(std::integral_constant<int, ACT_RELU>{});
        else elementwise(std::integral_constant<int, ACT_LINEAR>{});

        float csum[4] = {0.f, 0.f, 0.f, 0.f};
        const int prow = lane / LPR, pc4 = lane - prow * LPR;   // this lane's place in a pass
        const bool lane_on = lane < RPI * LPR;
        const int gcol = n0 + wn * WN + 4 * pc4;

        // one output plane: registers -> patch -> whole-row stores (+ ref - value column sums)
        auto flush = [&](const float (&val)[TM][TN][4], float* __restrict__ out, bool diff) {
            __syncthreads();   // patch free: k loop / previous plane finished
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        patch[(mi * 16 + slot * 4 + r) * LDE + ni * 16 + l15] = val[mi][ni][r];
            __syncthreads();
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                const int lrow = ps * RPI + prow;
                const int grow = m0 + wm * WM + lrow;
                if (lane_on && lrow < WM && grow < g.M && gcol < g.N) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(patch + lrow * LDE + 4 * pc4);
                    const int nval = g.N - gcol;   // >= 1
                    if (out) {
                        float* o = out + (size_t)grow * g.ldo + gcol;
                        if (nval >= 4) *reinterpret_cast<f32x4*>(o) = v;
                        else { o[0] = v.x; if (nval > 1) o[1] = v.y; if (nval > 2) o[2] = v.z; }
                    }
                    if (diff) {
                        // the ref row is padded to ld % 4 == 0, so the 16-B read stays inside it
                        const f32x4 rf = *reinterpret_cast<const f32x4*>(g.ref + (size_t)grow * g.ldref + gcol);
                        csum[0] += rf.x - v.x;
                        if (nval > 1) csum[1] += rf.y - v.y;
                        if (nval > 2) csum[2] += rf.z - v.z;
                        if (nval > 3) csum[3] += rf.w - v.w;
                    }
                }
            }
        };
        const bool diff_on_sample = (NOISE != NOISE_NONE);
        if (g.out_prob || (want_diff && !diff_on_sample)) flush(pv, g.out_prob, want_diff && !diff_on_sample);
        if (NOISE != NOISE_NONE && (g.out_sample || want_diff)) flush(sv, g.out_sample, want_diff);
        if (NOISE != NOISE_NONE && g.out_u) flush(uv, g.out_u, false);

        if (want_diff) {
            // column partials of this workgroup: sum over the RPI row groups of each wave and the
            // WAVES_M waves that share a column, in a fixed order (bit-reproducible)
            __syncthreads();
            float* red = smem;   // [WAVES_M * RPI][BN]
            if (lane_on) {
                float* dst = red + (wm * RPI + prow) * BN + wn * WN + 4 * pc4;
                dst[0] = csum[0]; dst[1] = csum[1]; dst[2] = csum[2]; dst[3] = csum[3];
            }
            __syncthreads();
            if (tid < BN) {
                float t = 0.f;
#pragma unroll
                for (int i = 0; i < WAVES_M * RPI; ++i) t += red[i * BN + tid];
                if (n0 + tid < g.N) g.colpart[(size_t)bm * g.ld_colpart + n0 + tid] = t;
            }
        }
        KURBM_STAMP_OUT();
        return;
    }

    // ---------------- epilogue: softplus row sums (free energy) ------------------------
    if (EPI == EPI_SOFTPLUS) {
        float rsum[TM][4];
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) rsum[mi][r] = 0.f;
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            const int col = n0 + wn * WN + ni * 16 + l15;
            const bool cok = col < g.N;
            const float bias = cok ? g.bias[col] : 0.f;
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (cok) rsum[mi][r] += softplusf(acc[mi][ni][r] + bias);
        }
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float t = row16_sum(rsum[mi][r]);       // (kurbm_device.h: DPP adds, not ds_bpermute chains)
                if (l15 == 0) smem[wn * BM + wm * WM + mi * 16 + slot * 4 + r] = t;
            }
        __syncthreads();
        if (tid < BM) {
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < WAVES_N; ++i) t += smem[i * BM + tid];
            if (m0 + tid < g.M) g.rowpart[(size_t)bn * g.ld_rowpart + m0 + tid] = t;
        }
        return;
    }
}

// ------------------------------------------------------------------------------------
// slab reduction (+ parameter update)
// ------------------------------------------------------------------------------------
// The first nbias8 blocks (nbias rounded up to 8, so that the blocks behind keep their XCD) reduce the bias partials
// (kurbm_kernels.h: bias_colsum_wave); the nblk_w blocks after them sum the split-K slabs of dW (4 columns per thread) and
// either add lr * dW into W or store dW densely.
__global__ __launch_bounds__(256) void k_reduce_apply(ReduceArgs a, int nbias, int nbias8) {
    warm_kernel_arguments<sizeof(ReduceArgs) + 8>();   // (kurbm_device.h: one wait for the argument segment, not one per use)
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < nbias8) {
        if ((int)blockIdx.x < nbias) bias_colsum_wave(a, blockIdx.x * 4 + (tid >> 6), tid & 63);
        return;
    }
    const int blk = (int)blockIdx.x - nbias8;
    if (blk < a.nblk_w) {
        if (a.tile_bm > 0) {
            // tile order: the same XCD remap as the GEMM, then (tile, part of the tile)
            const int nwg = a.nblk_w;
            int bid = blk;
            const int xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
            bid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
            const int tile = bid / a.parts, part = bid - tile * a.parts;
            const int bm = a.m_fastest ? tile % a.grid_m : tile / a.grid_n;
            const int bn = a.m_fastest ? tile / a.grid_m : tile - bm * a.grid_n;
            const int g4 = a.tile_bn / 4;
            const int rows_pp = (a.tile_bm + a.parts - 1) / a.parts;
            // one block = rows_pp x tile_bn floats; lanes walk the float4 groups of a row
            for (int e = tid; e < rows_pp * g4; e += 256) {
                const int r = part * rows_pp + e / g4;
                const int ii = bm * a.tile_bm + r, jj = bn * a.tile_bn + 4 * (e % g4);
                if (r >= a.tile_bm || ii >= a.n_vis || jj >= a.n_hid) continue;
                const float* sp = a.slab + (size_t)ii * a.ld_slab + jj;
                f32x4 s = {0.f, 0.f, 0.f, 0.f};
                int zz = 0;
                for (; zz + 4 <= a.nslab; zz += 4) {
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(sp + (size_t)(zz + 0) * a.slab_stride);
                    const f32x4 v1 = *reinterpret_cast<const f32x4*>(sp + (size_t)(zz + 1) * a.slab_stride);
                    const f32x4 v2 = *reinterpret_cast<const f32x4*>(sp + (size_t)(zz + 2) * a.slab_stride);
                    const f32x4 v3 = *reinterpret_cast<const f32x4*>(sp + (size_t)(zz + 3) * a.slab_stride);
                    s += v0; s += v1; s += v2; s += v3;
                }
                for (; zz < a.nslab; ++zz) s += *reinterpret_cast<const f32x4*>(sp + (size_t)zz * a.slab_stride);
                if (jj + 3 < a.n_hid) {
                    if (a.W) {
                        f32x4* w = reinterpret_cast<f32x4*>(a.W + (size_t)ii * a.ldw + jj);
                        *w = *w + s * a.lr;
                    }
                    if (a.delta_w) {
                        float* d = a.delta_w + (size_t)ii * a.n_hid + jj;
                        if ((a.n_hid & 3) == 0) *reinterpret_cast<f32x4*>(d) = s;
                        else { d[0] = s.x; d[1] = s.y; d[2] = s.z; d[3] = s.w; }
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (jj + u < a.n_hid) {
                            if (a.delta_w) a.delta_w[(size_t)ii * a.n_hid + jj + u] = s[u];
                            if (a.W) a.W[(size_t)ii * a.ldw + jj + u] += a.lr * s[u];
                        }
                }
            }
            return;
        }
    }
    if (blk < a.nblk_w) {   // row order (no tile geometry given)
        const int groups = a.ld_slab / 4;  // float4 groups per row
        const long long q = (long long)blk * 256 + tid;
        if (q >= (long long)a.n_vis * groups) return;
        const int i = (int)(q / groups), j0 = (int)(q - (long long)i * groups) * 4;
        // slabs are summed in index order (bit-reproducible); four loads in flight per lane
        const float* sp = a.slab + (size_t)i * a.ld_slab + j0;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        int zz = 0;
        for (; zz + 4 <= a.nslab; zz += 4) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(sp + (size_t)(zz + 0) * a.slab_stride);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(sp + (size_t)(zz + 1) * a.slab_stride);
            const f32x4 v2 = *reinterpret_cast<const f32x4*>(sp + (size_t)(zz + 2) * a.slab_stride);
            const f32x4 v3 = *reinterpret_cast<const f32x4*>(sp + (size_t)(zz + 3) * a.slab_stride);
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; zz < a.nslab; ++zz) s += *reinterpret_cast<const f32x4*>(sp + (size_t)zz * a.slab_stride);
        if (j0 + 3 < a.n_hid) {   // whole 16-B group inside the matrix: vector read-modify-write
            if (a.W) {
                f32x4* w = reinterpret_cast<f32x4*>(a.W + (size_t)i * a.ldw + j0);
                *w = *w + s * a.lr;
            }
            if (a.delta_w) {
                float* d = a.delta_w + (size_t)i * a.n_hid + j0;
                if ((a.n_hid & 3) == 0) *reinterpret_cast<f32x4*>(d) = s;
                else { d[0] = s.x; d[1] = s.y; d[2] = s.z; d[3] = s.w; }
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = j0 + e;
                if (j < a.n_hid) {
                    if (a.delta_w) a.delta_w[(size_t)i * a.n_hid + j] = s[e];
                    if (a.W) a.W[(size_t)i * a.ldw + j] += a.lr * s[e];
                }
            }
        }
        return;
    }
}

// params += lr * delta for a packed [V*H | H | V] delta
__global__ __launch_bounds__(256) void k_apply_delta(ApplyArgs a) {
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long nw = (long long)a.n_vis * a.n_hid;
    if (q < nw) {
        if (a.W) {
            const int i = (int)(q / a.n_hid), j = (int)(q - (long long)i * a.n_hid);
            a.W[(size_t)i * a.ldw + j] += a.lr * a.delta[q];
        }
    } else if (q < nw + a.n_hid) {
        if (a.b_h) a.b_h[q - nw] += a.lr * a.delta[q];
    } else if (q < nw + a.n_hid + a.n_vis) {
        if (a.b_v) a.b_v[q - nw - a.n_hid] += a.lr * a.delta[q];
    }
}

// F[row] = -( v[row].b_v + sum over column tiles of the softplus row partials )
__global__ __launch_bounds__(256) void k_free_energy_finish(FinishArgs a) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + wave;
    if (row >= a.rows) return;
    float t = 0.f;
    for (int c = lane; c < a.n_vis; c += 64) t += a.v[(size_t)row * a.ldv + c] * a.b_v[c];
    for (int i = lane; i < a.ncol_tiles; i += 64) t += a.rowpart[(size_t)i * a.ld_rowpart + row];   // (one partial per lane)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
    if (lane == 0) a.F[row] = -t;
}

// The score of fit(verbose = 1) (rbm.py:225-233).  One wave per row: F(v) = -(v.b_v + its softplus partials), the same for
// the reconstruction v', |F(v) - F(v')| to absdiff[row]; then ONE block adds the rows up in a fixed order (double) and
// writes the mean: nothing returns to the host, the caller copies the float back when it wants to print it.
__global__ __launch_bounds__(256) void k_score_rows(ScoreArgs a) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + wave;
    if (row < a.rows) {
        // v.b_v and v'.b_v: 16-byte loads, all of a row's in flight at once (rows are 16-byte aligned, ld % 4 == 0)
        const int n4 = a.n_vis >> 2;
        const float* vr = a.v + (size_t)row * a.ldv;
        float t = 0.f, t1 = 0.f;
        if (a.v1b) {
            // v' as the BYTE plane the half step left for the next GEMM (0x40 = one, k-permuted in groups of 64: kurbm_device.h
            // kperm64): lane l takes bytes [16 l, 16 l + 16) of the row = columns g + 8 s + 0..7 and g + 32 + 8 s + 0..7
            // (g = 64 (l / 4), s = l % 4) -- no fp32 copy of v' is written or read for a 0/1 reconstruction
            const unsigned char* br = a.v1b + (size_t)row * a.ldv1b;
            for (int q0 = 16 * lane; q0 < a.ldv1b; q0 += 16 * 64) {
                const u32x4s w = *reinterpret_cast<const u32x4s*>(br + q0);
                const int g = q0 & ~63, s8 = ((q0 >> 4) & 3) * 8;
                const uint32_t wd[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int c = g + s8 + (j & 7) + ((j >> 3) << 5);
                    if (((wd[j >> 2] >> (8 * (j & 3))) & 0xFFu) && c < a.n_vis) t1 += a.b_v[c];
                }
            }
            for (int c0 = 0; c0 < n4; c0 += 64) {
                const int c = c0 + lane;
                if (c < n4) {
                    const f32x4 x = *reinterpret_cast<const f32x4*>(vr + 4 * c), b = *reinterpret_cast<const f32x4*>(a.b_v + 4 * c);
                    t += x.x * b.x + x.y * b.y + x.z * b.z + x.w * b.w;
                }
            }
            for (int c = 4 * n4 + lane; c < a.n_vis; c += 64) t += vr[c] * a.b_v[c];
        } else {
            const float* v1r = a.v1 + (size_t)row * a.ldv1;
            for (int c0 = 0; c0 < n4; c0 += 256) {
                f32x4 x[4], y[4], b[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = c0 + e * 64 + lane;
                    const bool ok = c < n4;
                    x[e] = ok ? *reinterpret_cast<const f32x4*>(vr + 4 * c) : f32x4{0.f, 0.f, 0.f, 0.f};
                    y[e] = ok ? *reinterpret_cast<const f32x4*>(v1r + 4 * c) : f32x4{0.f, 0.f, 0.f, 0.f};
                    b[e] = ok ? *reinterpret_cast<const f32x4*>(a.b_v + 4 * c) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    t += x[e].x * b[e].x + x[e].y * b[e].y + x[e].z * b[e].z + x[e].w * b[e].w;
                    t1 += y[e].x * b[e].x + y[e].y * b[e].y + y[e].z * b[e].z + y[e].w * b[e].w;
                }
            }
            for (int c = 4 * n4 + lane; c < a.n_vis; c += 64) {      // (n_vis % 4 columns)
                const float b = a.b_v[c];
                t += vr[c] * b;
                t1 += v1r[c] * b;
            }
        }
        // the softplus partials of the row's column tiles join the same reduction, one per lane (a single lane walking 2 x 16
        // partials in dependent loads was the longest chain of this launch)
        for (int i = lane; i < a.ncol_tiles; i += 64) t += a.rowpart[(size_t)i * a.ld_rowpart + row];
        for (int i = lane; i < a.ncol_tiles1; i += 64) t1 += a.rowpart1[(size_t)i * a.ld_rowpart + row];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { t += __shfl_xor(t, o); t1 += __shfl_xor(t1, o); }
        if (lane == 0) {
            const float F = -t, F1 = -t1;
            if (a.F) { a.F[row] = F; a.F[a.rows + row] = F1; }
            a.absdiff[row] = fabsf(F - F1);
        }
    }
}

// (one block: the rows' |F - F'| added in a fixed order, in double.  Folded into k_score_rows as "the last workgroup to finish" it
//  was SLOWER -- 1 024 workgroups draining write-through stores and one of them reading 4 096 values past the L2: round 4)
__global__ __launch_bounds__(1024) void k_score_mean(ScoreArgs a) {
    __shared__ double part[1024];
    double s = 0.;
    for (int r = threadIdx.x; r < a.rows; r += 1024) s += (double)a.absdiff[r];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) a.score[0] = (float)(part[0] / (double)a.rows);
}

hipError_t launch_score(const ScoreArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(k_score_rows, dim3((a.rows + 3) / 4), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_score_mean, dim3(1), dim3(1024), 0, st, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------
template <int BM, int BN, int WM_, int WN_, bool A_KM, bool B_KM, int EPI, int NOISE>
static hipError_t launch_cfg(const GemmArgs& g, hipStream_t st) {
    const int nblk = g.grid_m * g.grid_n * g.nsplit;
    hipLaunchKernelGGL((k_gemm<BM, BN, WM_, WN_, A_KM, B_KM, EPI, NOISE>), dim3(nblk), dim3(NTHREADS), 0, st, g);
    return hipGetLastError();
}

void tile_shape(int cfg, int* bm, int* bn) {
    static const int shapes[CFG_COUNT][2] = {{128, 128}, {128, 112}, {112, 128}, {128, 64}, {64, 128}, {64, 64}, {64, 112}, {112, 64}};
    *bm = shapes[cfg][0];
    *bn = shapes[cfg][1];
}

static unsigned long long* g_stamp_buffer = nullptr;
static int g_dbg_off = 0;
void set_stamp_buffer(unsigned long long* p) { g_stamp_buffer = p; }
unsigned long long* get_stamp_buffer() { return g_stamp_buffer; }
void set_debug_off(int mask) { g_dbg_off = mask; }

hipError_t launch_gemm(int layout, int cfg, int epi, const GemmArgs& g_in, hipStream_t st) {
    GemmArgs g = g_in;
    g.stamps = g_stamp_buffer;
    g.dbg_off = g_dbg_off;
#define KURBM_CASE(L, AK, BKM, E, NZ)                                                               \
    if (layout == L && epi == E && (E != EPI_HALFSTEP || g.noise == NZ)) {                          \
        switch (cfg) {                                                                              \
            case CFG_128x128: return launch_cfg<128, 128, 2, 2, AK, BKM, E, NZ>(g, st);             \
            case CFG_128x112: return launch_cfg<128, 112, 4, 1, AK, BKM, E, NZ>(g, st);             \
            case CFG_112x128: return launch_cfg<112, 128, 1, 4, AK, BKM, E, NZ>(g, st);             \
            case CFG_128x64: return launch_cfg<128, 64, 2, 2, AK, BKM, E, NZ>(g, st);               \
            case CFG_64x128: return launch_cfg<64, 128, 2, 2, AK, BKM, E, NZ>(g, st);               \
            case CFG_64x64: return launch_cfg<64, 64, 2, 2, AK, BKM, E, NZ>(g, st);                 \
            case CFG_64x112: return launch_cfg<64, 112, 4, 1, AK, BKM, E, NZ>(g, st);               \
            case CFG_112x64: return launch_cfg<112, 64, 1, 4, AK, BKM, E, NZ>(g, st);               \
            default: return hipErrorInvalidValue;                                                   \
        }                                                                                           \
    }
    KURBM_CASE(LAYOUT_VH, false, true, EPI_HALFSTEP, NOISE_NONE)
    KURBM_CASE(LAYOUT_VH, false, true, EPI_HALFSTEP, NOISE_BERNOULLI)
    KURBM_CASE(LAYOUT_HV, false, false, EPI_HALFSTEP, NOISE_NONE)
    KURBM_CASE(LAYOUT_HV, false, false, EPI_HALFSTEP, NOISE_BERNOULLI)
    KURBM_CASE(LAYOUT_HV, false, false, EPI_HALFSTEP, NOISE_GAUSSIAN)
    KURBM_CASE(LAYOUT_VH, false, true, EPI_SOFTPLUS, NOISE_NONE)
    KURBM_CASE(LAYOUT_OUTER, true, true, EPI_SLAB, NOISE_NONE)
#undef KURBM_CASE
    return hipErrorInvalidValue;
}

hipError_t launch_philox_uniform(float* out, int rows, int cols, int ld, const RngArgs& rng, hipStream_t st) {
    dim3 grid((cols + 255) / 256, (rows + 3) / 4);
    hipLaunchKernelGGL(k_philox_uniform, grid, dim3(256), 0, st, out, rows, cols, ld, rng);
    return hipGetLastError();
}

hipError_t launch_reduce_apply(const ReduceArgs& a, hipStream_t st) {
    const int nb = bias_blocks(a), nb8 = (nb + 7) / 8 * 8;
    hipLaunchKernelGGL(k_reduce_apply, dim3(a.nblk_w + nb8), dim3(256), 0, st, a, nb, nb8);
    return hipGetLastError();
}

hipError_t launch_apply_delta(const ApplyArgs& a, hipStream_t st) {
    const long long n = (long long)a.n_vis * a.n_hid + a.n_hid + a.n_vis;
    hipLaunchKernelGGL(k_apply_delta, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_free_energy_finish(const FinishArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(k_free_energy_finish, dim3((a.rows + 3) / 4), dim3(256), 0, st, a);
    return hipGetLastError();
}

}  // namespace kurbm
