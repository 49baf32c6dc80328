// kurbm_bf16.hip -- bf16-storage / fp32-accumulate variant of the CD kernels (gfx950).
//
// BASELINE.json config 5 (4096 x 4096 RBM, bf16, CD-10 persistent chains) is 16x faster on the
// matrix cores in bf16 (v_mfma_f32_16x16x32_bf16) than the fp32 path, provided every GEMM operand is
// k-contiguous.  So this variant keeps bf16 MIRRORS in both orientations instead of re-staging:
//
//     W  [n_vis][n_hid]  (k = hidden)   -> B operand of the h->v half step
//     Wt [n_hid][n_vis]  (k = visible)  -> B operand of the v->h half step
//     activations: row-major [batch][units] (A operand of the next half step) AND transposed
//     [units][batch] (operands of the statistics GEMM, k = batch), both written by the epilogue
//
// Every GEMM is then "NT": A [M][K], B [N][K], one staging layout, one kernel.  All k extents are
// zero-padded to a multiple of 128 by the host side, so there is no masked tail path.  The fp32
// master weights stay authoritative: the slab reduction updates them (kurbm_kernels.hip) and the
// mirrors are re-quantised (round to nearest even) afterwards.
//
// Reference op sequences: the same as the fp32 kernels (reference ku/ebm/rbm.py:46-47, :52-53,
// :121-134); bf16 is an extension of this build, absent from the reference.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "kurbm_kernels.h"
#include "kurbm_device.h"

namespace kurbm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int NTH = 256;

// ------------------------------------------------------------------------------------
// fp32 [rows][ld_in] -> bf16 [rows_pad][ldo] (+ transposed bf16 [cols_pad][ldoT]); padding zeroed
// ------------------------------------------------------------------------------------
// One TR x 64 tile per workgroup through LDS (TR = 64, or 16 when the matrix is too small to fill the chip with
// 64-row tiles): 16-byte loads, 8-byte stores in both orientations.
// Out-of-range source elements read as zero, so the k padding of both mirrors is written here and
// nowhere else.  pieces = 3 writes the exact split x = hi + mid + lo (bf16_piece_bits) as three planes.
// colpart (nullable): column sums of each 64-row band of `in`.
constexpr int CVT = 64;
template <int TR>
__global__ __launch_bounds__(256) void k_f32_to_bf16(const float* __restrict__ in, int rows, int cols, int ld_in,
                                                     uint16_t* __restrict__ out, int ldo, int out_rows,
                                                     uint16_t* __restrict__ outT, int ldoT, int outT_rows,
                                                     int pieces, size_t out_plane, size_t outT_plane,
                                                     float* __restrict__ colpart, int ld_colpart) {
    __shared__ float tile[TR][CVT + 1];
    const int t = threadIdx.x;
    const int q4 = (t & 15) * 4, rq = t >> 4;           // 16 lanes x 4 elements across, 16 rows per pass
    const int r0 = blockIdx.y * TR, c0 = blockIdx.x * CVT;
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    auto store3 = [&](uint16_t* dst, size_t plane, float a, float b, float c, float d) {
        float v[4] = {a, b, c, d};
        for (int j = 0; j < pieces; ++j) {
            u32x2 pk;
            pk.x = pack_bf16x2(v[0], v[1]); pk.y = pack_bf16x2(v[2], v[3]);
            *reinterpret_cast<u32x2*>(dst + j * plane) = pk;
            if (j + 1 < pieces) {   // residual of the piece just written: exact in fp32
                v[0] -= bf16_bits_to_f32(pk.x & 0xFFFFu); v[1] -= bf16_bits_to_f32(pk.x >> 16);
                v[2] -= bf16_bits_to_f32(pk.y & 0xFFFFu); v[3] -= bf16_bits_to_f32(pk.y >> 16);
            }
        }
    };
#pragma unroll
    for (int j = 0; j < TR / 16; ++j) {
        const int r = r0 + rq + 16 * j, c = c0 + q4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r < rows && c < cols) {      // ld_in % 4 == 0 and c % 4 == 0: the 16-byte load stays inside the row
            v = *reinterpret_cast<const f32x4*>(in + (size_t)r * ld_in + c);
            if (c + 1 >= cols) v.y = 0.f;
            if (c + 2 >= cols) v.z = 0.f;
            if (c + 3 >= cols) v.w = 0.f;
        }
        tile[rq + 16 * j][q4 + 0] = v.x; tile[rq + 16 * j][q4 + 1] = v.y;
        tile[rq + 16 * j][q4 + 2] = v.z; tile[rq + 16 * j][q4 + 3] = v.w;
        if (out && r < out_rows && c < ldo) store3(out + (size_t)r * ldo + c, out_plane, v.x, v.y, v.z, v.w);   // ldo % 8 == 0
    }
    if (!outT && !colpart) return;
    __syncthreads();
    if (colpart && t < CVT && r0 < rows && c0 + t < cols) {   // out-of-range elements were staged as zeros
        float s = 0.f;
#pragma unroll 8
        for (int i = 0; i < TR; ++i) s += tile[i][t];
        colpart[(size_t)blockIdx.y * ld_colpart + c0 + t] = s;
    }
    if (!outT) return;
    // TR / 4 lanes x 4 rows across a column of the tile, 1024 / TR columns per pass
    constexpr int LPC = TR / 4, CPP = 256 / LPC;
    const int rr4 = (t % LPC) * 4, cq = t / LPC;
#pragma unroll
    for (int j = 0; j < CVT / CPP; ++j) {
        const int c = c0 + cq + CPP * j, r = r0 + rr4;    // outT[c][r .. r+3]
        if (c < outT_rows && r < ldoT)                    // ldoT % 8 == 0
            store3(outT + (size_t)c * ldoT + r, outT_plane, tile[rr4 + 0][cq + CPP * j], tile[rr4 + 1][cq + CPP * j],
                   tile[rr4 + 2][cq + CPP * j], tile[rr4 + 3][cq + CPP * j]);
    }
}

// ------------------------------------------------------------------------------------
// slab reduction + parameter update + weight-piece mirror in ONE launch (x3 / bf16 steps)
// ------------------------------------------------------------------------------------
// Blocks [0, tiles): a TR x 64 tile of W each (TR = 16 / 32 / 64) -- sum the split-K slabs in index order (bit-reproducible), W += lr * dW
// (and / or emit dW), then write the NEW weights as bf16 pieces row-major (8-byte stores) and, through an LDS transpose,
// transposed; the k padding of both mirrors is rewritten as zeros.  Remaining blocks: the bias column sums of
// k_reduce_apply (kurbm_kernels.hip).  Replaces k_reduce_apply + k_f32_to_bf16 on the fp32 master (one launch, one
// pass over W instead of three).
template <int TR>
__global__ __launch_bounds__(256) void k_reduce_apply_split(ReduceArgs a, int tiles_x, int tiles) {
    __shared__ float tile[TR][CVT + 1];
    const int t = threadIdx.x;
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    if ((int)blockIdx.x < tiles) {
        const int by = blockIdx.x / tiles_x, bx = blockIdx.x - by * tiles_x;
        const int q4 = (t & 15) * 4, rq = t >> 4;
        const int c = bx * CVT + q4;
        auto store3 = [&](uint16_t* dst, size_t plane, float v0, float v1, float v2, float v3) {
            float v[4] = {v0, v1, v2, v3};
            for (int j = 0; j < a.pieces; ++j) {
                u32x2 pk;
                pk.x = pack_bf16x2(v[0], v[1]); pk.y = pack_bf16x2(v[2], v[3]);
                *reinterpret_cast<u32x2*>(dst + j * plane) = pk;
                if (j + 1 < a.pieces) {
                    v[0] -= bf16_bits_to_f32(pk.x & 0xFFFFu); v[1] -= bf16_bits_to_f32(pk.x >> 16);
                    v[2] -= bf16_bits_to_f32(pk.y & 0xFFFFu); v[3] -= bf16_bits_to_f32(pk.y >> 16);
                }
            }
        };
#pragma unroll
        for (int j = 0; j < TR / 16; ++j) {
            const int r = by * TR + rq + 16 * j;
            f32x4 w = {0.f, 0.f, 0.f, 0.f};
            if (r < a.n_vis && c < a.n_hid) {
                f32x4 s = {0.f, 0.f, 0.f, 0.f};
                if (a.slab) {
                    const float* sp = a.slab + (size_t)r * a.ld_slab + c;    // ld_slab % 4 == 0: the group stays inside the row
                    int zz = 0;
                    for (; zz + 4 <= a.nslab; zz += 4) {
                        const f32x4 v0 = *reinterpret_cast<const f32x4*>(sp + (size_t)(zz + 0) * a.slab_stride);
                        const f32x4 v1 = *reinterpret_cast<const f32x4*>(sp + (size_t)(zz + 1) * a.slab_stride);
                        const f32x4 v2 = *reinterpret_cast<const f32x4*>(sp + (size_t)(zz + 2) * a.slab_stride);
                        const f32x4 v3 = *reinterpret_cast<const f32x4*>(sp + (size_t)(zz + 3) * a.slab_stride);
                        s += v0; s += v1; s += v2; s += v3;
                    }
                    for (; zz < a.nslab; ++zz) s += *reinterpret_cast<const f32x4*>(sp + (size_t)zz * a.slab_stride);
                }
                if (a.W) {
                    float* wp = a.W + (size_t)r * a.ldw + c;                  // ldw % 4 == 0
                    w = *reinterpret_cast<const f32x4*>(wp);
                    if (a.slab) w = w + s * a.lr;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (c + e >= a.n_hid) w[e] = 0.f;                     // the row padding stays zero
                    if (a.slab) *reinterpret_cast<f32x4*>(wp) = w;
                }
                if (a.delta_w) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (c + e < a.n_hid) a.delta_w[(size_t)r * a.n_hid + c + e] = s[e];
                }
            }
            tile[rq + 16 * j][q4 + 0] = w.x; tile[rq + 16 * j][q4 + 1] = w.y;
            tile[rq + 16 * j][q4 + 2] = w.z; tile[rq + 16 * j][q4 + 3] = w.w;
            if (a.Wb && r < a.n_vis && c < a.ldWb) store3(a.Wb + (size_t)r * a.ldWb + c, a.planeWb, w.x, w.y, w.z, w.w);
        }
        __syncthreads();
        if (a.Wtb) {
            // TR / 4 lanes x 4 rows down a column of the tile (a run of 2 TR bytes per column), 1024 / TR columns per pass
            constexpr int LPC = TR / 4, CPP = 256 / LPC;
            const int rr4 = (t % LPC) * 4, cq = t / LPC;
            const int rr = by * TR + rr4;
#pragma unroll
            for (int j = 0; j < CVT / CPP; ++j) {
                const int cl = cq + CPP * j, cc = bx * CVT + cl;
                if (cc < a.n_hid && rr < a.ldWtb)
                    store3(a.Wtb + (size_t)cc * a.ldWtb + rr, a.planeWtb, tile[rr4 + 0][cl], tile[rr4 + 1][cl], tile[rr4 + 2][cl],
                           tile[rr4 + 3][cl]);
            }
        }
        return;
    }
    // bias partials [row tiles][columns] -> column sums: eight loads in flight per lane, fixed order
    const int q = ((int)blockIdx.x - tiles) * 256 + t;
    // in double: the rows are +sum(h_pos) ... -sum(h_neg) per 64 batch rows, two large totals that cancel
    // (part2: n2 more rows that follow the n1 rows of part as if they were contiguous -- same additions, same order)
    auto colsum = [](const float* __restrict__ part, int n1, int ld, int col, const float* __restrict__ part2 = nullptr, int n2 = 0) {
        const int ntiles = n1 + (part2 ? n2 : 0);
        auto at = [&](int i) { return (double)(i < n1 ? part[(size_t)i * ld + col] : part2[(size_t)(i - n1) * ld + col]); };
        double u[8] = {0., 0., 0., 0., 0., 0., 0., 0.};
        int i = 0;
        for (; i + 8 <= ntiles; i += 8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) u[e] += at(i + e);
        }
        for (; i < ntiles; ++i) u[0] += at(i);
        return (float)(((u[0] + u[1]) + (u[2] + u[3])) + ((u[4] + u[5]) + (u[6] + u[7])));
    };
    if (q < a.n_hid) {
        if (a.part_h) {
            const float v = colsum(a.part_h, a.nrow_tiles_h, a.ld_part_h, q);
            if (a.delta_bh) a.delta_bh[q] = v;
            if (a.b_h) a.b_h[q] += a.lr * v;
        }
    } else if (q < a.n_hid + (a.n_vis_bias ? a.n_vis_bias : a.n_vis)) {
        const int col = q - a.n_hid;
        if (a.part_v) {
            const float v = colsum(a.part_v, a.nrow_tiles_v, a.ld_part_v, col, a.part_v2, a.nrow_tiles_v2);
            if (a.delta_bv) a.delta_bv[col] = v;
            if (a.b_v) a.b_v[col] += a.lr * v;
        }
    }
}

// flag := 1 if some element of `in` is not exactly representable in bf16 (the caller zeroes it)
__global__ __launch_bounds__(256) void k_bf16_exact_check(const float* __restrict__ in, int rows, int cols, int ld_in,
                                                          int* __restrict__ flag) {
    const int c4 = cols >> 2;   // ld % 4 == 0 and 16-B aligned rows: whole float4s, then the tail
    bool bad = false;
    for (int r = blockIdx.x; r < rows; r += gridDim.x) {
        const float* row = in + (size_t)r * ld_in;
        for (int c = threadIdx.x; c < c4; c += 256) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(row + 4 * c);
            bad |= ((__float_as_uint(v.x) | __float_as_uint(v.y) | __float_as_uint(v.z) | __float_as_uint(v.w)) & 0xFFFFu) != 0u;
        }
        for (int c = 4 * c4 + threadIdx.x; c < cols; c += 256) bad |= (__float_as_uint(row[c]) & 0xFFFFu) != 0u;
    }
    if (bad) *flag = 1;
}

// ------------------------------------------------------------------------------------
// the bf16 NT GEMM
// ------------------------------------------------------------------------------------
// BKB = k-tile in bf16 elements (2 * BKB bytes per LDS row, + 16 B pad -> conflict-free ds_read_b128).
// <128,128,BKB 128>: one workgroup per CU (139 KB LDS);  <128,64,BKB 64>: 55 KB, two per CU, whose
// MFMAs and staging instructions overlap across the two co-resident waves of a SIMD.
// WS ("wave specialised", 512 threads): waves 0-3 only read fragments and issue MFMAs, waves 4-7 only
// move tiles global -> registers -> LDS.  A bf16 MFMA leaves ~8 issue cycles per 16, far too few for the
// staging instructions of its own wave; with a loader wave beside every MFMA wave they issue in parallel.
template <int BM, int BN, int WAVES_M, int WAVES_N, int BKB, int EPI, int NOISE, bool WS = false>
__global__ __launch_bounds__(WS ? 2 * NTH : NTH) void k_gemm_bf16(GemmArgsB g) {
    static_assert(WAVES_M * WAVES_N == 4, "4 waves");
    constexpr int ROWB = 2 * BKB + 16;
    constexpr int CPR = BKB / 8;          // 16-B chunks per row
    constexpr int KS = BKB / 32;          // MFMA k-steps per tile
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int TM = WM / 16, TN = WN / 16;
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;
    constexpr int NA = BM * CPR / NTH, NB = BN * CPR / NTH;   // 16-B chunks per lane per tile
    static_assert((BM * CPR) % NTH == 0 && (BN * CPR) % NTH == 0, "whole chunks per lane");
    constexpr int LDE = WN + 4;
    constexpr int EPI_BYTES = 4 * WM * LDE * 4;
    constexpr int SMEM_BYTES = (2 * (A_BYTES + B_BYTES) > EPI_BYTES) ? 2 * (A_BYTES + B_BYTES) : EPI_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = WS && wave_all >= 4;      // wave-uniform role
    const int wave = loader ? wave_all - 4 : wave_all;
    const int stid = tid & (NTH - 1);             // staging thread id
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int l15 = lane & 15, slot = lane >> 4;

    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {   // XCD-aware bijective remap (see kurbm_kernels.hip)
        const int xcd = bid & 7, q = nwg >> 3, rr = nwg & 7;
        bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
    }
    const int tiles_mn = g.grid_m * g.grid_n;
    const int z = bid / tiles_mn;
    const int tmn = bid - z * tiles_mn;
    // walk the SHORTER grid dimension fastest: an XCD's contiguous run of tiles then spans the whole
    // short dimension and a slice of the long one, which minimises the operand bytes its L2 must hold
    const int bm = g.m_fastest ? tmn % g.grid_m : tmn / g.grid_n;
    const int bn = g.m_fastest ? tmn / g.grid_m : tmn - bm * g.grid_n;
    const int m0 = bm * BM, n0 = bn * BN;

    const int t_begin = z * g.kt_per_split;
    int t_end = t_begin + g.kt_per_split;
    if (t_end > g.kt_total) t_end = g.kt_total;
    const int nt = t_end > t_begin ? t_end - t_begin : 0;

    // per-lane staging map: chunk q -> (row q / CPR, 16-B chunk q % CPR); rows outside the matrix are
    // pointed at row 0 (they only feed outputs that are never stored)
    unsigned goffA[NA], goffB[NB];
    int soffA[NA], soffB[NB];
#pragma unroll
    for (int it = 0; it < NA; ++it) {
        const int q = it * NTH + stid, row = q / CPR, ch = q % CPR;
        const int x = (m0 + row < g.M) ? m0 + row : 0;
        goffA[it] = (unsigned)(x * g.lda + 8 * ch);
        soffA[it] = row * ROWB + 16 * ch;
    }
#pragma unroll
    for (int it = 0; it < NB; ++it) {
        const int q = it * NTH + stid, row = q / CPR, ch = q % CPR;
        const int x = (n0 + row < g.N) ? n0 + row : 0;
        goffB[it] = (unsigned)(x * g.ldb + 8 * ch);
        soffB[it] = row * ROWB + 16 * ch;
    }
    struct Regs { u32x4 a[NA], b[NB]; };
    Regs r0, r1;   // two tiles in flight between global memory and LDS (fetched two tiles ahead)
    unsigned char* sA0 = smem;
    unsigned char* sB0 = smem + 2 * A_BYTES;
    constexpr bool SIGNED = (EPI == EPI_SLAB);
    constexpr int NCH = NA + NB;

    auto tile_of = [&](int t, const uint16_t*& oa, const uint16_t*& ob, uint32_t& flip) {
        t = t < t_end ? t : t_end - 1;
        const int seg = g.inv_nkt ? (int)__umulhi((uint32_t)t, g.inv_nkt) : t;   // inv_nkt == 0: one k-tile per segment
        const uint32_t code = (uint32_t)(g.seg_codes >> (5 * seg)) & 31u;
        const bool neg = (code & 16u) != 0u;
        const size_t k0 = (size_t)(t - seg * g.nkt) * BKB;
        oa = (neg ? g.A1 : g.A0) + (code & 3u) * (neg ? g.a_plane1 : g.a_plane0) + k0;
        ob = (neg ? g.B1 : g.B0) + ((code >> 2) & 3u) * (neg ? g.b_plane1 : g.b_plane0) + k0;
        flip = neg ? 0x80008000u : 0u;   // sign bits of a bf16 pair: set 1 enters negated
    };
    auto fetch = [&](Regs& R, int t) {
        const uint16_t *oa, *ob;
        uint32_t flip;
        tile_of(t, oa, ob, flip);
#pragma unroll
        for (int it = 0; it < NA; ++it) R.a[it] = *reinterpret_cast<const u32x4*>(oa + goffA[it]);
#pragma unroll
        for (int it = 0; it < NB; ++it) R.b[it] = *reinterpret_cast<const u32x4*>(ob + goffB[it]);
    };
    // park chunks [c0, c1) of the tile held in R
    auto park = [&](const Regs& R, int buf, uint32_t flip, int c0, int c1) {
        unsigned char* a = sA0 + buf * A_BYTES;
        unsigned char* b = sB0 + buf * B_BYTES;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (c < c0 || c >= c1) continue;
            if (c < NA) {
                u32x4 v = R.a[c];
                if (SIGNED) { v.x ^= flip; v.y ^= flip; v.z ^= flip; v.w ^= flip; }
                *reinterpret_cast<u32x4*>(a + soffA[c]) = v;
            } else {
                *reinterpret_cast<u32x4*>(b + soffB[c - NA]) = R.b[c - NA];
            }
        }
    };
    if (WS && loader) {
        // ---- loader waves: their own loop and their own registers (the roles share no live values, so
        // the register allocator sees max(), not sum(), of the two roles' needs)
        if (nt > 0) {
            const uint16_t *oa, *ob;
            uint32_t flip;
            tile_of(t_begin, oa, ob, flip);
            fetch(r0, t_begin);
            park(r0, 0, flip, 0, NCH);
            fetch(r1, t_begin + 1);
            __syncthreads();
            int i = 0;
            for (; i + 1 < nt; i += 2) {
                tile_of(t_begin + i + 1, oa, ob, flip);
                fetch(r0, t_begin + i + 2);
                park(r1, 1, flip, 0, NCH);
                __syncthreads();
                tile_of(t_begin + i + 2, oa, ob, flip);
                fetch(r1, t_begin + i + 3);
                park(r0, 0, flip, 0, NCH);
                __syncthreads();
            }
            if (i < nt) {
                tile_of(t_begin + i + 1, oa, ob, flip);
                fetch(r0, t_begin + i + 2);
                park(r1, 1, flip, 0, NCH);
                __syncthreads();
            }
        }
        __syncthreads();
        if (EPI == EPI_SLAB) return;
        // keep the MFMA waves' epilogue barriers company (same counts as below)
        const bool diff = (g.ref32 != nullptr) || (g.ref16 != nullptr);
        const int nflush = (NOISE != NOISE_NONE) ? 1 + (g.prob_f32 ? 1 : 0) + (g.out_u ? 1 : 0) : 1;
        for (int f = 0; f < nflush; ++f) { __syncthreads(); __syncthreads(); }
        if (diff) { __syncthreads(); __syncthreads(); }
        return;
    }

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto frags = [&](int buf, int ks, u32x4 (&fa)[TM], u32x4 (&fb)[TN]) {
        const unsigned char* cA = sA0 + buf * A_BYTES + (wm * WM + l15) * ROWB + 16 * slot + 64 * ks;
        const unsigned char* cB = sB0 + buf * B_BYTES + (wn * WN + l15) * ROWB + 16 * slot + 64 * ks;
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) fa[mi] = *reinterpret_cast<const u32x4*>(cA + mi * 16 * ROWB);
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) fb[ni] = *reinterpret_cast<const u32x4*>(cB + ni * 16 * ROWB);
    };
    auto mfmas = [&](const u32x4 (&fa)[TM], const u32x4 (&fb)[TN]) {
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8, fa[mi]), __builtin_bit_cast(bf16x8, fb[ni]), acc[mi][ni], 0, 0, 0);
    };
    u32x4 fax[TM], fbx[TN], fay[TM], fby[TN];   // fragment double buffer: k-step s in x, s+1 in y, ...

    // One tile = KS k-steps of 32, software pipelined (same idea as the fp32 kernel):
    //   first k-step      request tile i+2 from global memory (registers L)
    //   middle k-steps    park tile i+1 (registers P, fetched a whole tile ago) in the other LDS buffer
    //   before the last   the tile's ONLY barrier (all fragments of the current buffer have been read)
    //   last k-step       read the NEXT tile's first fragments from the other buffer
    // Every k-step's fragment reads are issued one step ahead of the MFMAs that use them.
    auto one_tile = [&](int i, const int cur, Regs& L, const Regs& P) {
        const uint16_t *oa, *ob;
        uint32_t flip_p;
        tile_of(t_begin + i + 1, oa, ob, flip_p);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            __builtin_amdgcn_sched_barrier(0);
            if (!WS) {
                if (ks == 0) fetch(L, t_begin + i + 2);
                if (KS == 4) {
                    if (ks == 1) park(P, cur ^ 1, flip_p, 0, NCH / 2);
                    if (ks == 2) park(P, cur ^ 1, flip_p, NCH / 2, NCH);
                } else if (ks == 0) {
                    park(P, cur ^ 1, flip_p, 0, NCH);
                }
            }
            const bool last = (ks == KS - 1);
            if (ks & 1) {   // fragments alternate x, y, x, y
                if (!last) frags(cur, ks + 1, fax, fbx);
                if (last) { __syncthreads(); __builtin_amdgcn_sched_barrier(0); frags(cur ^ 1, 0, fax, fbx); }
                mfmas(fay, fby);
            } else {
                frags(cur, ks + 1, fay, fby);
                mfmas(fax, fbx);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    if (nt > 0) {
        const uint16_t *oa, *ob;
        uint32_t flip;
        tile_of(t_begin, oa, ob, flip);
        if (!WS) {
            fetch(r0, t_begin);
            park(r0, 0, flip, 0, NCH);
            fetch(r1, t_begin + 1);
        }
        __syncthreads();
        frags(0, 0, fax, fbx);
        // unrolled by two: register sets and LDS buffers alternate statically.  Branch-free: past the
        // end of the slice the last tile is fetched / parked again (in bounds, never read).
        int i = 0;
        for (; i + 1 < nt; i += 2) {
            one_tile(i, 0, r0, r1);
            one_tile(i + 1, 1, r1, r0);
        }
        if (i < nt) one_tile(i, 0, r0, r1);
    }
    __syncthreads();

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
        return;
    }

    // ---------------- epilogue: bias + activation + draw; planes leave through an LDS patch ----
    constexpr int LPR = WN / 8;          // lanes per output row, 8 bf16 = 16 B each
    constexpr int RPI = 64 / LPR;
    constexpr int NPASS = (WM + RPI - 1) / RPI;
    constexpr int LPRT = WM / 8;         // transposed plane: lanes per row of outT
    constexpr int RPIT = 64 / LPRT;
    constexpr int NPASST = (WN + RPIT - 1) / RPIT;
    float* patch = reinterpret_cast<float*>(smem) + wave * (WM * LDE);

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

    float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bool want_diff = (g.ref32 != nullptr) || (g.ref16 != nullptr);
    const int prow = lane / LPR, pc8 = lane - prow * LPR;
    const bool lane_on = lane < RPI * LPR;
    const int gcol = n0 + wn * WN + 8 * pc8;

    // registers -> patch; then (a) whole rows of the row-major planes, (b) whole rows of the
    // transposed bf16 plane, both 16 B per lane
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
        if (g.prob_f32) flush(pv, nullptr, nullptr, g.prob_f32, false);
        flush(sv, g.out, g.outT, g.out_f32, want_diff);
        if (g.out_u) flush(uv, nullptr, nullptr, g.out_u, false);
    }

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
}

// ------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------
hipError_t launch_f32_to_bf16(const float* in, int rows, int cols, int ld_in, uint16_t* out, int ldo, int out_rows,
                              uint16_t* outT, int ldoT, int outT_rows, int pieces, size_t out_plane, size_t outT_plane,
                              float* colpart, int ld_colpart, hipStream_t st) {
    // cover the padded extents of whichever mirrors are requested
    int r_ext = rows, c_ext = cols;
    if (out) { if (out_rows > r_ext) r_ext = out_rows; if (ldo > c_ext) c_ext = ldo; }
    if (outT) { if (ldoT > r_ext) r_ext = ldoT; if (outT_rows > c_ext) c_ext = outT_rows; }
    // colpart bands are 64 rows, so a launch that wants them keeps 64-row tiles
    const bool small = !colpart && ((c_ext + CVT - 1) / CVT) * ((r_ext + CVT - 1) / CVT) < 512;
    if (small) {
        dim3 grid((c_ext + CVT - 1) / CVT, (r_ext + 15) / 16);
        hipLaunchKernelGGL(k_f32_to_bf16<16>, grid, dim3(256), 0, st, in, rows, cols, ld_in, out, ldo, out_rows, outT, ldoT,
                           outT_rows, pieces, out_plane, outT_plane, colpart, ld_colpart);
    } else {
        dim3 grid((c_ext + CVT - 1) / CVT, (r_ext + CVT - 1) / CVT);
        hipLaunchKernelGGL(k_f32_to_bf16<64>, grid, dim3(256), 0, st, in, rows, cols, ld_in, out, ldo, out_rows, outT, ldoT,
                           outT_rows, pieces, out_plane, outT_plane, colpart, ld_colpart);
    }
    return hipGetLastError();
}

hipError_t launch_reduce_apply_split(const ReduceArgs& a, hipStream_t st) {
    // the tile grid covers the padded extents of both mirrors (their zero k padding is written here)
    int r_ext = a.n_vis, c_ext = a.n_hid;
    if (a.Wb && a.ldWb > c_ext) c_ext = a.ldWb;
    if (a.Wtb && a.ldWtb > r_ext) r_ext = a.ldWtb;
    const int nb = (a.n_hid + (a.n_vis_bias ? a.n_vis_bias : a.n_vis) + 255) / 256;
    const int tiles_x = (c_ext + CVT - 1) / CVT;
    // taller tiles = longer runs in the transposed mirror (2 TR bytes per column) but fewer workgroups
    static const int forced = getenv("KURBM_REDUCE_TR") ? atoi(getenv("KURBM_REDUCE_TR")) : 0;
    int tr = forced ? forced : (tiles_x * ((r_ext + 31) / 32) >= 384 ? 32 : 16);
    const int tiles_y = (r_ext + tr - 1) / tr;
    const dim3 grid(tiles_x * tiles_y + nb);
    if (tr == 64) hipLaunchKernelGGL(k_reduce_apply_split<64>, grid, dim3(256), 0, st, a, tiles_x, tiles_x * tiles_y);
    else if (tr == 32) hipLaunchKernelGGL(k_reduce_apply_split<32>, grid, dim3(256), 0, st, a, tiles_x, tiles_x * tiles_y);
    else hipLaunchKernelGGL(k_reduce_apply_split<16>, grid, dim3(256), 0, st, a, tiles_x, tiles_x * tiles_y);
    return hipGetLastError();
}

hipError_t launch_bf16_exact_check(const float* in, int rows, int cols, int ld_in, int* flag, hipStream_t st) {
    hipLaunchKernelGGL(k_bf16_exact_check, dim3(rows < 1024 ? rows : 1024), dim3(256), 0, st, in, rows, cols, ld_in, flag);
    return hipGetLastError();
}

hipError_t launch_gemm_bf16(int epi, const GemmArgsB& g, hipStream_t st) {
    const int nblk = g.grid_m * g.grid_n * g.nsplit;
#define KURBM_B(E, NZ)                                                                              \
    if (epi == E && (E != EPI_HALFSTEP || g.noise == NZ)) {                                         \
        if (g.cfg == 1) hipLaunchKernelGGL((k_gemm_bf16<128, 64, 2, 2, 64, E, NZ>), dim3(nblk), dim3(NTH), 0, st, g);   \
        else if (g.cfg == 2) hipLaunchKernelGGL((k_gemm_bf16<128, 128, 2, 2, 128, E, NZ, true>), dim3(nblk), dim3(2 * NTH), 0, st, g); \
        else hipLaunchKernelGGL((k_gemm_bf16<128, 128, 2, 2, 128, E, NZ>), dim3(nblk), dim3(NTH), 0, st, g);            \
        return hipGetLastError();                                                                   \
    }
    KURBM_B(EPI_HALFSTEP, NOISE_NONE)
    KURBM_B(EPI_HALFSTEP, NOISE_BERNOULLI)
    KURBM_B(EPI_HALFSTEP, NOISE_GAUSSIAN)
    KURBM_B(EPI_SLAB, NOISE_NONE)
#undef KURBM_B
    return hipErrorInvalidValue;
}

}  // namespace kurbm
