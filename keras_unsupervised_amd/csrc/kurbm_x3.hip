// kurbm_x3.hip -- the NT bf16 GEMM of the x3 path: fp32 operands as exact bf16 triples (gfx950).
//
// v_mfma_f32_16x16x32_bf16 runs 16x faster than v_mfma_f32_16x16x4_f32, and an fp32 value is exactly
// hi + mid + lo with three bf16 pieces (kurbm_device.h: bf16_piece_bits).  In a CD step one operand of
// every product is a 0/1 sample (one exact piece) and the other is real-valued (W, or the h_neg
// probabilities): A . B = A . B_hi + A . B_mid + A . B_lo, three bf16 MFMAs into the SAME fp32
// accumulators.  So this kernel's k-tile holds ONE tile of A and up to PB = 3 tiles of B (the pieces of
// the same rows): per k-step a wave reads its A fragments once and multiplies them with the fragments
// of every piece.  Against running the three products as consecutive k ranges (kurbm_bf16.hip) that is
// a third fewer bytes from L2 and through LDS per MFMA -- the limit of a 128 x 128 bf16 tile on this chip.
//
// Segments (GemmArgsB::seg_codes): segment s multiplies piece `ia` of A (set 0 or 1) with pieces
// 0 .. npb-1 of B of the same set; set 1 enters negated (negative phase of the statistics).  A
// real-valued A (grey-level data) is handled as three segments (ia = 0, 1, 2 with npb = 3, 2, 1).
//
// Tile 128 x 128, k-tile 64, 8 waves (2 x 4, 64 x 32 outputs each), two per SIMD: while one waits for
// LDS or the barrier the other issues MFMAs.  Staging global -> registers -> LDS, double-buffered in LDS,
// fetched two tiles ahead, one barrier per tile (the scheme of kurbm_bf16.hip / kurbm_kernels.hip).
//
// Reference op sequences: ku/ebm/rbm.py:46-47 (v->h), :52-53 / :121-123 (h->v), :124 (h_neg), :125-134
// (statistics); the split is an implementation choice of this build.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "kurbm_kernels.h"
#include "kurbm_device.h"

namespace kurbm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// Diagnostic build only (-DKURBM_STAMPS): s_memtime at the phase boundaries of a workgroup (wave 0),
// written to GemmArgsB::stamps as 8 x u64: start, prologue done, k loop done, elementwise done, planes
// flushed, end.  The shipped library has no stamps.
#ifdef KURBM_STAMPS
#define KURBM_STAMP(var)                                                                        \
    do {                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                      \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");             \
        __builtin_amdgcn_sched_barrier(0);                                                      \
    } while (0)
#else
#define KURBM_STAMP(var) do { } while (0)
#endif

template <int BM, int BN, int WAVES_M, int WAVES_N, int BKB, int PB, int EPI, int NOISE>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void k_gemm_pb(GemmArgsB g) {
    constexpr int NT = 64 * WAVES_M * WAVES_N;
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int ROWB = 2 * BKB + 16;    // LDS row: k-tile + 16 B pad -> conflict-free ds_read_b128
    constexpr int CPR = BKB / 8;          // 16-B chunks per row
    constexpr int KS = BKB / 32;          // MFMA k-steps per tile
    static_assert(KS == 2, "fragment buffers alternate with the k-step");
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int TM = WM / 16, TN = WN / 16;
    constexpr int A_BYTES = BM * ROWB, B1_BYTES = BN * ROWB, B_BYTES = PB * B1_BYTES;
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int NA = BM * CPR / NT, NB1 = BN * CPR / NT;   // 16-B chunks per lane: A tile, ONE piece of B
    static_assert((BM * CPR) % NT == 0 && (BN * CPR) % NT == 0, "whole chunks per lane");
    constexpr int LDE = WN + 4;
    constexpr int EPI_BYTES = NW * WM * LDE * 4;
    constexpr int SMEM_BYTES = (2 * STAGE > EPI_BYTES) ? 2 * STAGE : EPI_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int l15 = lane & 15, slot = lane >> 4;
#ifdef KURBM_STAMPS
    unsigned long long ts[6] = {0, 0, 0, 0, 0, 0};
    KURBM_STAMP(ts[0]);
#define KURBM_STAMP_OUT()                                                                         \
    do {                                                                                          \
        KURBM_STAMP(ts[5]);                                                                       \
        if (g.stamps && tid == 0) {                                                               \
            unsigned long long* o = g.stamps + (size_t)blockIdx.x * 8;                            \
            for (int q = 0; q < 6; ++q) o[q] = ts[q];                                             \
        }                                                                                         \
    } while (0)
#else
#define KURBM_STAMP_OUT() do { } while (0)
#endif

    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {   // XCD-aware bijective remap (see kurbm_kernels.hip)
        const int xcd = bid & 7, q = nwg >> 3, rr = nwg & 7;
        bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
    }
    const int tiles_mn = g.grid_m * g.grid_n;
    const int z = bid / tiles_mn;
    const int tmn = bid - z * tiles_mn;
    const int bm = g.m_fastest ? tmn % g.grid_m : tmn / g.grid_n;
    const int bn = g.m_fastest ? tmn / g.grid_m : tmn - bm * g.grid_n;
    const int m0 = bm * BM, n0 = bn * BN;

    const int t_begin = z * g.kt_per_split;
    int t_end = t_begin + g.kt_per_split;
    if (t_end > g.kt_total) t_end = g.kt_total;
    const int nt = t_end > t_begin ? t_end - t_begin : 0;

    // per-lane staging map: chunk q -> (row q / CPR, 16-B chunk q % CPR); rows outside the matrix are
    // pointed at row 0 (they only feed outputs that are never stored)
    unsigned goffA[NA], goffB[NB1];
    int soffA[NA], soffB[NB1];
#pragma unroll
    for (int it = 0; it < NA; ++it) {
        const int q = it * NT + tid, row = q / CPR, ch = q % CPR;
        const int x = (m0 + row < g.M) ? m0 + row : 0;
        goffA[it] = (unsigned)(x * g.lda + 8 * ch);
        soffA[it] = row * ROWB + 16 * ch;
    }
#pragma unroll
    for (int it = 0; it < NB1; ++it) {
        const int q = it * NT + tid, row = q / CPR, ch = q % CPR;
        const int x = (n0 + row < g.N) ? n0 + row : 0;
        goffB[it] = (unsigned)(x * g.ldb + 8 * ch);
        soffB[it] = row * ROWB + 16 * ch;
    }
    struct Regs { u32x4 a[NA], b[PB][NB1]; };
    Regs r0, r1;   // two tiles in flight between global memory and LDS (fetched two tiles ahead)
    constexpr bool SIGNED = (EPI == EPI_SLAB);

    struct TileRef { const uint16_t* oa; const uint16_t* ob; size_t bplane; uint32_t flip; int npb; };
    auto tile_of = [&](int t) {
        TileRef r;
        t = t < t_end ? t : t_end - 1;
        int seg, kt;
        if (g.seg_fastest) {
            kt = g.inv_nseg ? (int)__umulhi((uint32_t)t, g.inv_nseg) : t;       // inv == 0: divisor 1
            seg = t - kt * g.nseg;
        } else {
            seg = g.inv_nkt ? (int)__umulhi((uint32_t)t, g.inv_nkt) : t;
            kt = t - seg * g.nkt;
        }
        const uint32_t code = (uint32_t)(g.seg_codes >> (5 * seg)) & 31u;
        const bool neg = (code & 16u) != 0u;
        const size_t k0 = (size_t)kt * BKB;
        r.oa = (neg ? g.A1 : g.A0) + (code & 3u) * (neg ? g.a_plane1 : g.a_plane0) + k0;
        r.ob = (neg ? g.B1 : g.B0) + k0;
        r.bplane = neg ? g.b_plane1 : g.b_plane0;
        r.flip = neg ? 0x80008000u : 0u;   // sign bits of a bf16 pair: set 1 enters negated
        r.npb = (int)((code >> 2) & 3u);
        return r;
    };
    // branch-free: a piece the segment does not use is fetched from the last one it does (same lines)
    auto fetch = [&](Regs& R, int t) {
        const TileRef r = tile_of(t);
#pragma unroll
        for (int it = 0; it < NA; ++it) R.a[it] = *reinterpret_cast<const u32x4*>(r.oa + goffA[it]);
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            const uint16_t* ob = r.ob + (size_t)(p < r.npb ? p : r.npb - 1) * r.bplane;
#pragma unroll
            for (int it = 0; it < NB1; ++it) R.b[p][it] = *reinterpret_cast<const u32x4*>(ob + goffB[it]);
        }
    };
    // park chunks [c0, c1) of the tile held in R: chunk order = A, piece 0 of B, piece 1, ...
    constexpr int NCH = NA + PB * NB1;
    auto park = [&](const Regs& R, int buf, const TileRef& r, int c0, int c1) {
        unsigned char* a = smem + buf * STAGE;
        unsigned char* b = a + A_BYTES;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (c < c0 || c >= c1) continue;
            if (c < NA) {
                u32x4 v = R.a[c];
                if (SIGNED) { v.x ^= r.flip; v.y ^= r.flip; v.z ^= r.flip; v.w ^= r.flip; }
                *reinterpret_cast<u32x4*>(a + soffA[c]) = v;
            } else {
                const int p = (c - NA) / NB1, it = (c - NA) % NB1;
                if (p < r.npb) *reinterpret_cast<u32x4*>(b + p * B1_BYTES + soffB[it]) = R.b[p][it];
            }
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    u32x4 fa[2][TM], fb[2][TN];   // fragment double buffers: A by k-step, B by micro-step
    auto frag_a = [&](int buf, int ks, u32x4 (&f)[TM]) {
        const unsigned char* c = smem + buf * STAGE + (wm * WM + l15) * ROWB + 16 * slot + 64 * ks;
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) f[mi] = *reinterpret_cast<const u32x4*>(c + mi * 16 * ROWB);
    };
    auto frag_b = [&](int buf, int ks, int p, u32x4 (&f)[TN]) {
        const unsigned char* c = smem + buf * STAGE + A_BYTES + p * B1_BYTES + (wn * WN + l15) * ROWB + 16 * slot + 64 * ks;
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) f[ni] = *reinterpret_cast<const u32x4*>(c + ni * 16 * ROWB);
    };
    auto mfmas = [&](const u32x4 (&a)[TM], const u32x4 (&b)[TN]) {
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8, a[mi]), __builtin_bit_cast(bf16x8, b[ni]), acc[mi][ni], 0, 0, 0);
    };

    // One tile with NPB pieces = KS * NPB micro-steps (k-step ks = u / NPB, piece p = u % NPB), pipelined:
    //   u = 0           request tile i+2 from global memory (registers L)
    //   middle steps    park tile i+1 (registers P, requested a whole tile ago) in the other LDS buffer
    //   last step       the tile's ONLY barrier, then read the NEXT tile's first fragments
    // Every micro-step's fragment reads are issued one step ahead of the MFMAs that use them.
    auto one_tile = [&](int i, const int cur, Regs& L, const Regs& P, auto npb_tag) {
        constexpr int NPB = decltype(npb_tag)::value;
        constexpr int NU = KS * NPB;
        const TileRef rp = tile_of(t_begin + i + 1);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            __builtin_amdgcn_sched_barrier(0);
            if (u == 0) fetch(L, t_begin + i + 2);
            if constexpr (NU >= 4) {   // parks spread over the middle micro-steps
                constexpr int NMID = NU - 2;
                if (u >= 1 && u <= NMID) park(P, cur ^ 1, rp, (u - 1) * NCH / NMID, u * NCH / NMID);
            } else {
                if (u == 0) park(P, cur ^ 1, rp, 0, NCH);
            }
            const int ks = u / NPB;
            if (u + 1 < NU) {
                const int ksn = (u + 1) / NPB, pn = (u + 1) % NPB;
                if (pn == 0) frag_a(cur, ksn, fa[ksn & 1]);
                frag_b(cur, ksn, pn, fb[(u + 1) & 1]);
            } else {
                __syncthreads();
                __builtin_amdgcn_sched_barrier(0);
                frag_a(cur ^ 1, 0, fa[0]);
                frag_b(cur ^ 1, 0, 0, fb[0]);
            }
            mfmas(fa[ks & 1], fb[u & 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto tile_any = [&](int i, const int cur, Regs& L, const Regs& P) {
        const int npb = tile_of(t_begin + i).npb;
        if (PB >= 3 && npb == 3) one_tile(i, cur, L, P, std::integral_constant<int, 3>{});
        else if (PB >= 2 && npb == 2) one_tile(i, cur, L, P, std::integral_constant<int, 2>{});
        else one_tile(i, cur, L, P, std::integral_constant<int, 1>{});
    };

    if (nt > 0) {
        const TileRef r = tile_of(t_begin);
        fetch(r0, t_begin);
        park(r0, 0, r, 0, NCH);
        fetch(r1, t_begin + 1);
        __syncthreads();
        frag_a(0, 0, fa[0]);
        frag_b(0, 0, 0, fb[0]);
        KURBM_STAMP(ts[1]);
        // unrolled by two: register sets and LDS buffers alternate statically.  Branch-free: past the
        // end of the slice the last tile is fetched / parked again (in bounds, never read).
        int i = 0;
        for (; i + 1 < nt; i += 2) {
            tile_any(i, 0, r0, r1);
            tile_any(i + 1, 1, r1, r0);
        }
        if (i < nt) tile_any(i, 0, r0, r1);
    }
    __syncthreads();
    KURBM_STAMP(ts[2]);

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

    // ---------------- epilogue: bias + activation + draw; planes leave through an LDS patch ----
    // (the epilogue of kurbm_bf16.hip for a WM x WN wave tile)
    constexpr int LPR = WN / 8;          // lanes per output row, 8 bf16 = 16 B each
    constexpr int RPI = 64 / LPR;
    constexpr int NPASS = (WM + RPI - 1) / RPI;
    constexpr int LPRT = WM / 8;         // transposed plane: lanes per row of outT
    constexpr int RPIT = 64 / LPRT;
    constexpr int NPASST = (WN + RPIT - 1) / RPIT;
    float* patch = reinterpret_cast<float*>(smem) + wave * (WM * LDE);
    const bool side = (g.prob_f32 != nullptr) || (g.out_u != nullptr);   // test planes

    float pv[TM][TN][4], sv[TM][TN][4], uv[TM][TN][4];
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
                    philox4x32_10((uint32_t)col, (uint32_t)(grow >> 2), g.rng.stream_id, g.rng.step, g.rng.seed_lo,
                                  g.rng.seed_hi, w);
                    if (NOISE == NOISE_GAUSSIAN)
                        philox4x32_10((uint32_t)col, (uint32_t)(grow >> 2), g.rng.stream_id | 0x80000000u, g.rng.step,
                                      g.rng.seed_lo, g.rng.seed_hi, w2);
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
                        sm = p + sqrtf(-2.0f * logf(1.0f - ua)) * cospif(2.0f * ub);
                    }
                    pv[mi][ni][r] = p; sv[mi][ni][r] = sm; uv[mi][ni][r] = ua;
                }
            }
        }
    };
    if (g.act == ACT_SIGMOID) elementwise(std::integral_constant<int, ACT_SIGMOID>{});
    else if (g.act == ACT_RELU) elementwise(std::integral_constant<int, ACT_RELU>{});
    else elementwise(std::integral_constant<int, ACT_LINEAR>{});
    KURBM_STAMP(ts[3]);

    float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bool want_diff = (g.ref32 != nullptr) || (g.ref16 != nullptr);
    const int prow = lane / LPR, pc8 = lane - prow * LPR;
    const bool lane_on = lane < RPI * LPR;
    const int gcol = n0 + wn * WN + 8 * pc8;
    const bool pieces3 = (g.outT_pieces == 3);

    auto flush = [&](const float (&val)[TM][TN][4], uint16_t* __restrict__ o16, uint16_t* __restrict__ o16T,
                     float* __restrict__ o32, bool diff) {
        __syncthreads();
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) patch[(mi * 16 + slot * 4 + r) * LDE + ni * 16 + l15] = val[mi][ni][r];
        __syncthreads();
        if (o16 || o32 || diff) {
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                const int lrow = ps * RPI + prow;
                const int grow = m0 + wm * WM + lrow;
                if (lane_on && lrow < WM && grow < g.M && gcol < g.ldo_cols) {
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(patch + lrow * LDE + 8 * pc8);
                    const f32x4 v1 = *reinterpret_cast<const f32x4*>(patch + lrow * LDE + 8 * pc8 + 4);
                    const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                    const int nval = g.N - gcol;   // <= 0 in the zero padding of the row
                    if (o16) {   // rows of the bf16 planes are padded to 8 elements: the 16-B store stays inside
                        u32x4 pk;
                        pk.x = pack_bf16x2(nval > 0 ? v[0] : 0.f, nval > 1 ? v[1] : 0.f);
                        pk.y = pack_bf16x2(nval > 2 ? v[2] : 0.f, nval > 3 ? v[3] : 0.f);
                        pk.z = pack_bf16x2(nval > 4 ? v[4] : 0.f, nval > 5 ? v[5] : 0.f);
                        pk.w = pack_bf16x2(nval > 6 ? v[6] : 0.f, nval > 7 ? v[7] : 0.f);
                        *reinterpret_cast<u32x4*>(o16 + (size_t)grow * g.ldo + gcol) = pk;
                    }
                    if (o32) {
                        float* o = o32 + (size_t)grow * g.ldo32 + gcol;
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (e < nval) o[e] = v[e];
                    }
                    if (diff) {
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (e < nval) {
                                const float rf = g.ref32 ? g.ref32[(size_t)grow * g.ldref32 + gcol + e]
                                                         : bf16_bits_to_f32(g.ref16[(size_t)grow * g.ldref16 + gcol + e]);
                                csum[e] += rf - v[e];
                            }
                    }
                }
            }
        }
        if (o16T) {
            const int trow = lane / LPRT, tc8 = lane - trow * LPRT;   // row of outT = output column
#pragma unroll
            for (int ps = 0; ps < NPASST; ++ps) {
                const int lcol = ps * RPIT + trow;
                const int gn = n0 + wn * WN + lcol;
                const int gb = m0 + wm * WM + 8 * tc8;
                if (lane < RPIT * LPRT && lcol < WN && gn < g.N && gb < g.ldoT) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (gb + e < g.M) ? patch[(8 * tc8 + e) * LDE + lcol] : 0.f;
                    uint16_t* dst = o16T + (size_t)gn * g.ldoT + gb;
                    const int np = pieces3 ? 3 : 1;
                    for (int j = 0; j < np; ++j) {
                        u32x4 pk;
                        pk.x = pack_bf16x2(v[0], v[1]); pk.y = pack_bf16x2(v[2], v[3]);
                        pk.z = pack_bf16x2(v[4], v[5]); pk.w = pack_bf16x2(v[6], v[7]);
                        *reinterpret_cast<u32x4*>(dst + j * g.outT_plane) = pk;
                        if (j + 1 < np) {   // residual of the piece just written: exact in fp32
                            const uint32_t w[4] = {pk.x, pk.y, pk.z, pk.w};
#pragma unroll
                            for (int e = 0; e < 8; ++e)
                                v[e] -= bf16_bits_to_f32((e & 1) ? (w[e >> 1] >> 16) : (w[e >> 1] & 0xFFFFu));
                        }
                    }
                }
            }
        }
    };
    const bool on_sample = (NOISE != NOISE_NONE);
    if (!on_sample) flush(pv, g.out, g.outT, g.out_f32, want_diff);
    else {
        if (side && g.prob_f32) flush(pv, nullptr, nullptr, g.prob_f32, false);
        flush(sv, g.out, g.outT, g.out_f32, want_diff);
        if (side && g.out_u) flush(uv, nullptr, nullptr, g.out_u, false);
    }
    KURBM_STAMP(ts[4]);

    if (want_diff) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);   // [WAVES_M * RPI][BN]
        if (lane_on) {
            float* dst = red + (wm * RPI + prow) * BN + wn * WN + 8 * pc8;
#pragma unroll
            for (int e = 0; e < 8; ++e) dst[e] = csum[e];
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
}

// ------------------------------------------------------------------------------------
// launcher: 128 x 128 tiles, 512 threads
// ------------------------------------------------------------------------------------
hipError_t launch_gemm_pb(int epi, const GemmArgsB& g_in, hipStream_t st) {
    GemmArgsB g = g_in;
    {   // diagnostic build: launch n after the buffer was set writes at base + n * 4096 entries
        static unsigned long long* last_base = nullptr;
        static int nth = 0;
        unsigned long long* base = get_stamp_buffer();
        if (base != last_base) { last_base = base; nth = 0; }
        g.stamps = base ? base + (size_t)(nth++) * 4096 : nullptr;
    }
    const int nblk = g.grid_m * g.grid_n * g.nsplit;
#define KURBM_PB(E, NZ)                                                                              \
    if (epi == E && (E != EPI_HALFSTEP || g.noise == NZ)) {                                          \
        hipLaunchKernelGGL((k_gemm_pb<128, 128, 2, 4, 64, 3, E, NZ>), dim3(nblk), dim3(512), 0, st, g); \
        return hipGetLastError();                                                                    \
    }
    KURBM_PB(EPI_HALFSTEP, NOISE_NONE)
    KURBM_PB(EPI_HALFSTEP, NOISE_BERNOULLI)
    KURBM_PB(EPI_SLAB, NOISE_NONE)
#undef KURBM_PB
    return hipErrorInvalidValue;
}

}  // namespace kurbm
