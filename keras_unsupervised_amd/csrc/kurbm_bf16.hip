// kurbm_bf16.hip -- the kernels AROUND the bf16 GEMM of the x3 and rounded-bf16 paths (gfx950): fp32 -> bf16 plane
// conversion (rounded, or the exact three-piece split), the slab reduction that also rewrites the weight-piece mirror, the
// bf16-exactness check.  The GEMM itself is k_gemm_pb (kurbm_x3.hip) for both paths: PB = 3 pieces per weight for x3,
// PB = 1 for the rounded-bf16 path of BASELINE.json config 5 (its first kernel, k_gemm_bf16, measured the same 1.20 ms per
// 4096 x 4096 PCD-10 step as k_gemm_pb<PB = 1> and was removed -- DESIGN.md section 4).
//
// bf16 operands run 16x faster on the matrix cores (v_mfma_f32_16x16x32_bf16) than fp32 ones, provided every GEMM
// operand is k-contiguous.  So these paths keep bf16 MIRRORS in both orientations instead of re-staging:
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
#include <string.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "kurbm_kernels.h"
#include "kurbm_device.h"

namespace kurbm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));


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
                                                     float* __restrict__ colpart, int ld_colpart, int outT_f8, int out_bytes) {
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
        if (out && out_bytes) {   // 0/1 data as a byte plane (0x40 = one, k-permuted: kperm64), ldo bytes between its rows
            if (r < out_rows && c < ldo)
                *reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(out) + (size_t)r * ldo + kperm64(c)) =   // (c % 4 == 0)
                    (v.x != 0.f ? 0x40u : 0u) | (v.y != 0.f ? 0x4000u : 0u) | (v.z != 0.f ? 0x400000u : 0u) | (v.w != 0.f ? 0x40000000u : 0u);
        } else if (out && r < out_rows && c < ldo) store3(out + (size_t)r * ldo + c, out_plane, v.x, v.y, v.z, v.w);   // ldo % 8 == 0
    }
    if (!outT && !colpart) return;
    __syncthreads();
    if (colpart) {   // out-of-range elements were staged as zeros.  Four lanes per column, a quarter of the rows each (one wave's
                     // chain of TR dependent LDS reads was 2 us of this launch), combined in a fixed order: (q0 + q1) + (q2 + q3)
        const int cc = t >> 2, part = t & 3;
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < TR / 4; ++i) s += tile[part * (TR / 4) + i][cc];
        s += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(s), 0xB1, 0xF, 0xF, true));   // lane ^ 1
        s += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(s), 0x4E, 0xF, 0xF, true));   // lane ^ 2
        if (part == 0 && r0 < rows && c0 + cc < cols) colpart[(size_t)blockIdx.y * ld_colpart + c0 + cc] = s;
    }
    if (!outT) return;
    // TR / 4 lanes x 4 rows across a column of the tile, 1024 / TR columns per pass
    constexpr int LPC = TR / 4, CPP = 256 / LPC;
    const int rr4 = (t % LPC) * 4, cq = t / LPC;
#pragma unroll
    for (int j = 0; j < CVT / CPP; ++j) {
        const int c = c0 + cq + CPP * j, r = r0 + rr4;    // outT[c][r .. r+3]
        if (outT_f8) {   // 0/1 data: four fp8 bytes (1.0 = 0x38), row stride as for bf16
            if (c < outT_rows && r < ldoT)
                *reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(outT) + (size_t)c * ldoT * 2 + r) =
                    (tile[rr4 + 0][cq + CPP * j] != 0.f ? 0x38u : 0u) | (tile[rr4 + 1][cq + CPP * j] != 0.f ? 0x3800u : 0u) |
                    (tile[rr4 + 2][cq + CPP * j] != 0.f ? 0x380000u : 0u) | (tile[rr4 + 3][cq + CPP * j] != 0.f ? 0x38000000u : 0u);
            continue;
        }
        if (c < outT_rows && r < ldoT)                    // ldoT % 8 == 0
            store3(outT + (size_t)c * ldoT + r, outT_plane, tile[rr4 + 0][cq + CPP * j], tile[rr4 + 1][cq + CPP * j],
                   tile[rr4 + 2][cq + CPP * j], tile[rr4 + 3][cq + CPP * j]);
    }
}

// ------------------------------------------------------------------------------------
// 0/1 data: fp32 [rows][ld_in] -> the k-permuted byte plane (0x40 = one; kperm64) and / or the transposed fp8 plane (0x38 = one)
// ------------------------------------------------------------------------------------
// The planes a 0/1 batch travels as on the x3 path (kurbm_x3_convert_rows, and every step of a first epoch).  64 x 64 elements per
// workgroup: a thread loads a 4 x 4 block (four 16-byte loads), writes its four row dwords into a row image and its four column
// dwords into a column image in LDS, and every global store is 16 bytes -- a row of the tile is one 64-byte run in either plane.
// (k_f32_to_bf16 wrote both planes in 4-byte stores and summed a column in one wave's chain of 64 LDS reads: 10 us for 19 MB.)
// Out-of-range source elements read as zero: the k padding of both planes is written here.  colpart (nullable): the column sums
// of each 64-row band -- counts of ones, exact in any order.
__global__ __launch_bounds__(256) void k_f32_to_planes01(const float* __restrict__ in, int rows, int cols, int ld_in,
                                                         unsigned char* __restrict__ out, int ldo, int out_rows,
                                                         unsigned char* __restrict__ outT, int ldoT, int outT_rows,
                                                         float* __restrict__ colpart, int ld_colpart) {
    constexpr int PR = 80, PC = 68;   // image pitches (bytes): 16-byte rows for the row image; 17 dwords for the column image, whose
                                      // writes then fall on 32 different banks (80: on two)
    __shared__ __attribute__((aligned(16))) unsigned char rimg[64 * PR];
    __shared__ __attribute__((aligned(16))) unsigned char cimg[64 * PC];
    const int t = threadIdx.x;
    const int q4 = (t & 15) * 4, r4 = (t >> 4) * 4;       // this thread's 4 x 4 block of the tile
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    uint32_t colw[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + r4 + i, c = c0 + q4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r < rows && c < cols) {      // ld_in % 4 == 0 and c % 4 == 0: the 16-byte load stays inside the row
            v = *reinterpret_cast<const f32x4*>(in + (size_t)r * ld_in + c);
            if (c + 1 >= cols) v.y = 0.f;
            if (c + 2 >= cols) v.z = 0.f;
            if (c + 3 >= cols) v.w = 0.f;
        }
        const bool b0 = v.x != 0.f, b1 = v.y != 0.f, b2 = v.z != 0.f, b3 = v.w != 0.f;
        *reinterpret_cast<uint32_t*>(rimg + (r4 + i) * PR + kperm64(q4)) =      // (q4 % 4 == 0: four consecutive k stay consecutive)
            (b0 ? 0x40u : 0u) | (b1 ? 0x4000u : 0u) | (b2 ? 0x400000u : 0u) | (b3 ? 0x40000000u : 0u);
        const uint32_t one = 0x38u << (8 * i);
        colw[0] |= b0 ? one : 0u; colw[1] |= b1 ? one : 0u; colw[2] |= b2 ? one : 0u; colw[3] |= b3 ? one : 0u;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) *reinterpret_cast<uint32_t*>(cimg + (q4 + k) * PC + r4) = colw[k];
    __syncthreads();
    const int line = t >> 2, part = t & 3;   // a row of the row image / a column of the column image, 16 bytes of it
    if (out) {
        const int r = r0 + line, c = c0 + 16 * part;
        if (r < out_rows && c < ldo)         // (ldo % 64 == 0, the launcher checks: whole k-permuted groups)
            *reinterpret_cast<u32x4*>(out + (size_t)r * ldo + c) = *reinterpret_cast<const u32x4*>(rimg + line * PR + 16 * part);
    }
    if (outT || colpart) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(cimg + line * PC + 16 * part);
        const u32x4 x = {src[0], src[1], src[2], src[3]};
        const int c = c0 + line, r = r0 + 16 * part;
        if (outT && c < outT_rows && r < ldoT)   // row stride as for bf16 (2 ldoT bytes), the first ldoT of them used
            *reinterpret_cast<u32x4*>(outT + (size_t)c * ldoT * 2 + r) = x;
        if (colpart) {
            int n = __popc(x.x & 0x08080808u) + __popc(x.y & 0x08080808u) + __popc(x.z & 0x08080808u) + __popc(x.w & 0x08080808u);
            n += __builtin_amdgcn_mov_dpp(n, 0xB1, 0xF, 0xF, true);   // lane ^ 1
            n += __builtin_amdgcn_mov_dpp(n, 0x4E, 0xF, 0xF, true);   // lane ^ 2
            if (part == 0 && r0 < rows && c < cols) colpart[(size_t)blockIdx.y * ld_colpart + c] = (float)n;
        }
    }
}

// ------------------------------------------------------------------------------------
// slab reduction + parameter update + weight-piece mirror in ONE launch (x3 / bf16 steps)
// ------------------------------------------------------------------------------------
// The first `nbias` blocks: the bias column sums (kurbm_kernels.h: bias_colsum_wave; first, so that they do not trail the
// launch).  Then `tiles` blocks, a TR x 64 tile of W each (TR = 16 / 32 / 64) -- sum the split-K slabs in index order (bit-reproducible), W += lr * dW
// (and / or emit dW), then write the NEW weights as bf16 pieces row-major (8-byte stores) and, through an LDS transpose,
// transposed; the k padding of both mirrors is rewritten as zeros.  Replaces k_reduce_apply + k_f32_to_bf16 on the fp32 master (one launch, one
// pass over W instead of three).
// PEER (a template flag, so that the ordinary launch carries none of it): the summed statistics come from the ranks' exchange
// buffers (kurbm_peer.hip) -- the rows of band q from rank q's `sum` region once its `summed` flag is up: the all-gather of the
// two-shot all-reduce IS this launch's read.  Workgroup 0 also waits for EVERY rank's flag (a rank with an empty band still read
// this rank's bias tail: nobody's `delta` may be rewritten before all are done).
template <int TR, bool PEER>
__device__ __forceinline__ void reduce_apply_split_body(ReduceArgs& a, int tiles_x, int tiles, int nbias, const PeerSrc& ps) {
    __shared__ __attribute__((aligned(16))) float tile[TR][CVT + 1];
    __shared__ int peer_ok;
    const int t = threadIdx.x;
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    if constexpr (PEER) {
        if (blockIdx.x == 0) {
            if (t == 0) peer_ok = 1;
            __syncthreads();
            if (t < ps.nranks && !wait_flag_ge(ps.summed[t], ps.epoch, ps.timeout_ticks)) peer_ok = 0;
            __syncthreads();
            if (!peer_ok) {
                if (t == 0) __hip_atomic_fetch_or(ps.status, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
        }
    }
    // the first nbias blocks: bias column sums (kurbm_kernels.h); then the tiles of W
    if ((int)blockIdx.x < nbias) {
        bias_colsum_wave(a, blockIdx.x * 4 + (t >> 6), t & 63);
        return;
    }
    const int tb = (int)blockIdx.x - nbias;
    if (tb < tiles) {
        const int by = tb / tiles_x, bx = tb - by * tiles_x;
        if constexpr (PEER) {
            // the band of this tile's rows (band_rows is a multiple of TR; tiles past the matrix only write mirror padding)
            int q = (by * TR) / ps.band_rows;
            if (q >= ps.nranks) q = ps.nranks - 1;
            if (t == 0) peer_ok = (by * TR >= a.n_vis || wait_flag_ge(ps.summed[q], ps.epoch, ps.timeout_ticks)) ? 1 : 0;
            __syncthreads();
            if (!peer_ok) {
                if (t == 0) __hip_atomic_fetch_or(ps.status, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
            a.slab = ps.sum[q];      // (nslab = 1, ld_slab = n_hid: the packed layout)
        }
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
        // Every load of the thread's TR / 16 rows goes out before the first sum: the rows' weights and their first four slabs (this launch
        // is a burst of ~17 MB of loads that every CU must keep in flight: with the rows one after the other a wave held 5 x 16 bytes per
        // lane in flight, now TR / 16 times that).  The slabs are still added in slab order, row by row: the same bits as before.
        constexpr int NJ = TR / 16;
        f32x4 wv[NJ], sv[NJ][4];
        bool live[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int r = by * TR + rq + 16 * j;
            live[j] = r < a.n_vis && c < a.n_hid;
            wv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int z = 0; z < 4; ++z) sv[j][z] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (live[j]) {
                if (a.W) wv[j] = *reinterpret_cast<const f32x4*>(a.W + (size_t)r * a.ldw + c);           // ldw % 4 == 0
                if (a.slab) {
                    const float* sp = a.slab + (size_t)r * a.ld_slab + c;    // ld_slab % 4 == 0: the group stays inside the row
#pragma unroll
                    for (int z = 0; z < 4; ++z)
                        if (z < a.nslab) sv[j][z] = *reinterpret_cast<const f32x4*>(sp + (size_t)z * a.slab_stride);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int r = by * TR + rq + 16 * j;
            f32x4 w = {0.f, 0.f, 0.f, 0.f};
            if (live[j]) {
                f32x4 s = {0.f, 0.f, 0.f, 0.f};
                if (a.slab) {
                    const float* sp = a.slab + (size_t)r * a.ld_slab + c;
#pragma unroll
                    for (int z = 0; z < 4; ++z)
                        if (z < a.nslab) s += sv[j][z];
                    int zz = 4;
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
                    float* wp = a.W + (size_t)r * a.ldw + c;
                    w = wv[j];
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
                if (cc < a.n_hid && rr < (a.wtb_k_ext ? a.wtb_k_ext : a.ldWtb))
                    store3(a.Wtb + (size_t)cc * a.ldWtb + rr, a.planeWtb, tile[rr4 + 0][cl], tile[rr4 + 1][cl], tile[rr4 + 2][cl],
                           tile[rr4 + 3][cl]);
            }
        }
        return;
    }
}
// (two entry points: the ordinary launch does not carry the 184 bytes of peer pointers in its argument segment)
template <int TR>
__global__ __launch_bounds__(256) void k_reduce_apply_split(ReduceArgs a, int tiles_x, int tiles, int nbias) {
    warm_kernel_arguments<sizeof(ReduceArgs) + 16>();   // (kurbm_device.h: one wait for the argument segment, not one per use)
    PeerSrc none;
    none.band_rows = 0;
    reduce_apply_split_body<TR, false>(a, tiles_x, tiles, nbias, none);
}
template <int TR>
__global__ __launch_bounds__(256) void k_reduce_apply_split_peer(ReduceArgs a, int tiles_x, int tiles, int nbias, PeerSrc ps) {
    warm_kernel_arguments<sizeof(ReduceArgs) + 16 + sizeof(PeerSrc)>();
    reduce_apply_split_body<TR, true>(a, tiles_x, tiles, nbias, ps);
}

// flag |= 1 if some element of `in` is not exactly representable in bf16, |= 2 if some element is neither 0.0 nor 1.0
// (the caller zeroes it)
__global__ __launch_bounds__(256) void k_bf16_exact_check(const float* __restrict__ in, int rows, int cols, int ld_in,
                                                          int* __restrict__ flag) {
    const int c4 = cols >> 2;   // ld % 4 == 0 and 16-B aligned rows: whole float4s, then the tail
    bool bad = false, other = false;
    auto look = [&](float x) {
        const uint32_t b = __float_as_uint(x);
        bad |= (b & 0xFFFFu) != 0u;
        other |= (b != 0u) && (b != 0x3F800000u);
    };
    for (int r = blockIdx.x; r < rows; r += gridDim.x) {
        const float* row = in + (size_t)r * ld_in;
        for (int c = threadIdx.x; c < c4; c += 256) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(row + 4 * c);
            look(v.x); look(v.y); look(v.z); look(v.w);
        }
        for (int c = 4 * c4 + threadIdx.x; c < cols; c += 256) look(row[c]);
    }
    const int bits = (bad ? 1 : 0) | (other ? 2 : 0);
    if (bits) atomicOr(flag, bits);
}

// test hook: one element per thread, no attempt at speed (kurbm_kernels.h: DumpArgs)
__global__ __launch_bounds__(256) void k_dump_plane(DumpArgs a) {
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
    if (q >= (long long)a.rows * a.units) return;
    const int r = (int)(q / a.units), u = (int)(q - (long long)r * a.units);
    const int major = a.transposed ? u : r, minor = a.transposed ? r : u;
    float v = 0.f;
    if (a.fmt == 0) {
        const uint16_t* s = static_cast<const uint16_t*>(a.src);
        for (int j = 0; j < a.pieces; ++j) v += bf16_bits_to_f32(s[(size_t)j * a.plane + (size_t)major * a.ld + minor]);
    } else if (a.fmt == 1) {
        const unsigned char b = static_cast<const unsigned char*>(a.src)[(size_t)major * a.ld + kperm64(minor)];
        v = (b == 0x40) ? 1.f : (b == 0 ? 0.f : __int_as_float(0x7FC00000));      // anything else is not a plane byte: NaN
    } else {
        // (a transposed fp8 plane keeps the bf16 plane's row stride, 2 ld bytes: kurbm_x3.hip outT_f8)
        const unsigned char b = static_cast<const unsigned char*>(a.src)[(size_t)major * a.ld * 2 + minor];
        v = (b == 0x38) ? 1.f : (b == 0 ? 0.f : __int_as_float(0x7FC00000));
    }
    a.out[(size_t)r * a.ld_out + u] = a.sign * v;
}

hipError_t launch_dump_plane(const DumpArgs& a, hipStream_t st) {
    const long long n = (long long)a.rows * a.units;
    hipLaunchKernelGGL(k_dump_plane, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------
hipError_t launch_f32_to_bf16(const float* in, int rows, int cols, int ld_in, uint16_t* out, int ldo, int out_rows,
                              uint16_t* outT, int ldoT, int outT_rows, int pieces, size_t out_plane, size_t outT_plane,
                              float* colpart, int ld_colpart, hipStream_t st, int outT_f8, int out_bytes) {
    if ((outT_f8 || out_bytes) && pieces != 1) return hipErrorInvalidValue;
    // cover the padded extents of whichever mirrors are requested
    int r_ext = rows, c_ext = cols;
    if (out) { if (out_rows > r_ext) r_ext = out_rows; if (ldo > c_ext) c_ext = ldo; }
    if (outT) { if (ldoT > r_ext) r_ext = ldoT; if (outT_rows > c_ext) c_ext = outT_rows; }
    // 0/1 data as a byte plane and / or an fp8 transposed plane: the 16-byte-store kernel
    if (pieces == 1 && (out || outT) && (!out || out_bytes) && (!outT || outT_f8 == 1) && (!out || ldo % 64 == 0) && (!outT || ldoT % 8 == 0)) {
        dim3 grid((c_ext + 63) / 64, (r_ext + 63) / 64);
        hipLaunchKernelGGL(k_f32_to_planes01, grid, dim3(256), 0, st, in, rows, cols, ld_in, reinterpret_cast<unsigned char*>(out), ldo,
                           out_rows, reinterpret_cast<unsigned char*>(outT), ldoT, outT_rows, colpart, ld_colpart);
        return hipGetLastError();
    }
    // colpart bands are 64 rows, so a launch that wants them keeps 64-row tiles
    const bool small = !colpart && ((c_ext + CVT - 1) / CVT) * ((r_ext + CVT - 1) / CVT) < 512;
    if (small) {
        dim3 grid((c_ext + CVT - 1) / CVT, (r_ext + 15) / 16);
        hipLaunchKernelGGL(k_f32_to_bf16<16>, grid, dim3(256), 0, st, in, rows, cols, ld_in, out, ldo, out_rows, outT, ldoT,
                           outT_rows, pieces, out_plane, outT_plane, colpart, ld_colpart, outT_f8, out_bytes);
    } else {
        dim3 grid((c_ext + CVT - 1) / CVT, (r_ext + CVT - 1) / CVT);
        hipLaunchKernelGGL(k_f32_to_bf16<64>, grid, dim3(256), 0, st, in, rows, cols, ld_in, out, ldo, out_rows, outT, ldoT,
                           outT_rows, pieces, out_plane, outT_plane, colpart, ld_colpart, outT_f8, out_bytes);
    }
    return hipGetLastError();
}

hipError_t launch_reduce_apply_split(const ReduceArgs& a, hipStream_t st, const PeerSrc* peer) {
    // the tile grid covers the padded extents of both mirrors (their zero k padding is written here)
    int r_ext = a.n_vis, c_ext = a.n_hid;
    if (a.Wb && a.ldWb > c_ext) c_ext = a.ldWb;
    if (a.Wtb && (a.wtb_k_ext ? a.wtb_k_ext : a.ldWtb) > r_ext) r_ext = a.wtb_k_ext ? a.wtb_k_ext : a.ldWtb;
    const int nb = bias_blocks(a);
    const int tiles_x = (c_ext + CVT - 1) / CVT;
    // taller tiles = longer runs in the transposed mirror (2 TR bytes per column) but fewer workgroups
    const int forced = (a.tile_rows == 16 || a.tile_rows == 32 || a.tile_rows == 64) ? a.tile_rows : 0;
    int tr = forced ? forced : (tiles_x * ((r_ext + 31) / 32) >= 384 ? 32 : 16);
    PeerSrc ps;
    memset(&ps, 0, sizeof ps);
    if (peer && peer->band_rows) {
        ps = *peer;
        if (tr > 32) tr = 32;                                   // (band boundaries are multiples of 32 rows)
        if (ps.band_rows % 32 != 0 || nb < 1 || a.nslab != 1) return hipErrorInvalidValue;
    }
    const int tiles_y = (r_ext + tr - 1) / tr;
    const dim3 grid(tiles_x * tiles_y + nb);
    if (ps.band_rows) {
        if (tr == 32) hipLaunchKernelGGL(k_reduce_apply_split_peer<32>, grid, dim3(256), 0, st, a, tiles_x, tiles_x * tiles_y, nb, ps);
        else hipLaunchKernelGGL(k_reduce_apply_split_peer<16>, grid, dim3(256), 0, st, a, tiles_x, tiles_x * tiles_y, nb, ps);
    } else if (tr == 64) hipLaunchKernelGGL(k_reduce_apply_split<64>, grid, dim3(256), 0, st, a, tiles_x, tiles_x * tiles_y, nb);
    else if (tr == 32) hipLaunchKernelGGL(k_reduce_apply_split<32>, grid, dim3(256), 0, st, a, tiles_x, tiles_x * tiles_y, nb);
    else hipLaunchKernelGGL(k_reduce_apply_split<16>, grid, dim3(256), 0, st, a, tiles_x, tiles_x * tiles_y, nb);
    return hipGetLastError();
}

hipError_t launch_bf16_exact_check(const float* in, int rows, int cols, int ld_in, int* flag, hipStream_t st) {
    hipLaunchKernelGGL(k_bf16_exact_check, dim3(rows < 1024 ? rows : 1024), dim3(256), 0, st, in, rows, cols, ld_in, flag);
    return hipGetLastError();
}

}  // namespace kurbm
