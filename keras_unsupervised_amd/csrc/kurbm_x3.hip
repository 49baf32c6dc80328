// kurbm_x3.hip -- the NT bf16 GEMM of the x3 path: fp32 operands as exact bf16 triples (gfx950).
//
// v_mfma_f32_16x16x32_bf16 runs 16x faster than v_mfma_f32_16x16x4_f32, and an fp32 value is exactly
// hi + mid + lo with three bf16 pieces (kurbm_device.h: bf16_piece_bits).  In a CD step one operand of
// every product is a 0/1 sample (one exact piece) and the other is real-valued (W, or the h_neg
// probabilities): A . B = A . B_hi + A . B_mid + A . B_lo, three bf16 MFMAs into the SAME fp32
// accumulators.  So this kernel's k-tile holds ONE tile of A and up to PB = 3 tiles of B (the pieces of
// the same rows), LDS-staged: per k-step a wave reads its A fragments once and multiplies them with the
// fragments of every piece.  Against running the three products as consecutive k ranges
// (kurbm_bf16.hip) that is a third fewer bytes from L2 and through LDS per MFMA.
//
// Segments (GemmArgsB::seg_codes): segment s multiplies piece `ia` of A (set 0 or 1) with pieces
// 0 .. npb-1 of B of the same set; set 1 is the negative phase of the statistics (its B planes are stored negated).  A
// real-valued A (grey-level data) is three segments (ia = 0, 1, 2 with npb = 3, 2, 1).
//
// One kernel, k_gemm_pb<BM, BN, WAVES_M, WAVES_N, 64, PB, EPI, NOISE, AB, RP>: 768 threads = 8 MFMA waves of 64 x 32 outputs (two per
// SIMD: while one waits for LDS or the barrier the other issues MFMAs) + 4 LOADER waves that stage by LDS-DMA; k-tile 64 in bare
// 128-byte LDS rows, chunk-swizzled; one barrier per tile; B fragments read two micro-steps ahead.  What runs where (round 4):
//   half steps on 0/1 states (AB)      256 x 64, the A operand a k-permuted BYTE plane (128-deep A blocks, v_perm_b32 expansion),
//                                      two B stages, no tile list: one tile body, running offsets
//   statistics, 0/1 data (ABS)         256 x 64, split-K 4, units of [fp8 tile | 3-piece | 3-piece] on byte planes, three A buffers,
//                                      every request 10 pieces per wave at least a tile ahead of its reader
//   half steps on a 3-piece batch,     128 x 128, the PAIRED walk (BSP): two tiles of 48 MFMAs per wave and k position, four A stages
//   negative statistics of Gaussian      beside two B stages (160 KB)
//   visibles
//   statistics of real-valued data     two launches: the positive half TRANSPOSED (A = h_pos^T as bytes, B = the pieces of v_pos^T;
//   (ABP: template slot RP on EPI_SLAB)  the tile leaves transposed into ordinary slabs), the negative half on bytes or on BSP
//   one-piece tiles (rounded bf16)     128 x 128, four B stages + three A blocks, tiles requested four ahead, fragments read a tile
//                                      ahead across the barrier (DEEP)
//   everything else                    the generic two-stage walk of a tile list (segments of one to three pieces: KURBM_X3_PAIR=0,
//                                      KURBM_X3_SPLIT_STATS=0, the free-energy GEMM of real-valued data, fp8 positive statistics on
//                                      bf16 planes)
// Entry: the argument segment warmed in one batch (kurbm_device.h), block mapping by multiply-high constants into XCD-aware 2-D
// blocks, the loaders at high priority until their first requests are out.
//
// Epilogue of a half step, from registers: bias + activation + Philox draw (the words of a lane's first columns drawn in the prologue
// and parked in LDS); column sums of the value plane (bias statistics); the transposed plane(s) in 16-byte stores after a 4 x 4
// transpose across lanes (v_permlane32_swap / v_permlane16_swap): fp8 or k-permuted bytes for a 0/1 sample, three negated bf16 pieces
// for h_neg; only the row-major plane (bytes of a 0/1 sample, else bf16 pieces) takes a trip through LDS.  RP: also the softplus row
// sums of F(v) from the same accumulators (the score), by DPP adds.
//
// Measured (MI355X, config 2, round 4, sustained loops; DESIGN.md section 4, profiles/r04_*): sampling half steps 20.1 / 22.4 us, v -> h
// prob 21.7, statistics 22.1-22.8 (0.40-0.42 of its MFMA time at the dense peaks), slab reduce + mirrors 9.3; per launch ~2.4 us to
// the first fragments, ~1 920 cycles per 1 536-cycle tile, 2-3 us of draws + sigmoid, 3-4 us of plane write-back.  Everything that
// was tried and dropped: LABBOOK.md.
//
// Reference op sequences: ku/ebm/rbm.py:46-47 (v->h), :52-53 / :121-123 (h->v), :124 (h_neg), :125-134
// (statistics); the split is an implementation choice of this build.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "kurbm_kernels.h"
#include "kurbm_device.h"

namespace kurbm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

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

#ifndef KURBM_PRIO_MFMA
#define KURBM_PRIO_MFMA 2     // s_setprio of the MFMA waves / of the loader waves (A/B builds)
#endif
#ifndef KURBM_PRIO_LOADER
#define KURBM_PRIO_LOADER 0
#endif
#ifndef KURBM_PRIO_ENTRY
#define KURBM_PRIO_ENTRY 3    // the loader waves' priority from the kernel's first instruction to their first barrier (0: A/B builds)
#endif
#ifndef KURBM_DEEP_FENCE
#define KURBM_DEEP_FENCE 1   // the deep schedule's barrier: 1 = __syncthreads (every fragment read has RETURNED), 0 = bare s_barrier
#endif

// x / d through the launcher's ceil(2^32 / d) (x, d < 2^16: exact; inv == 0 stands for d = 1)
__device__ __forceinline__ int div_magic(int x, uint32_t inv) { return inv ? (int)__umulhi((uint32_t)x, inv) : x; }
// (slow = 1: a grid beyond the constants' exact range -- launch_gemm_pb -- pays for the division)
__device__ __forceinline__ int div_map(int x, int d, uint32_t inv, int slow) { return slow ? x / d : div_magic(x, inv); }

// Wave-specialised: four LOADER waves beside the eight MFMA waves (768 threads, three waves per SIMD).  The loaders do all
// the staging in a loop of their own -- LDS-DMA, `buffer_load_dwordx4 ... lds`: no register and no ds_write between memory
// and LDS -- and the MFMA waves only read fragments, issue MFMAs and meet the loaders at the tile's barrier.  A stall of a
// staging instruction then never sits in the instruction stream of a wave that has MFMAs to issue: with staging in the
// MFMA waves the k loop took twice the time of its MFMAs alone (round 1), wherever in the tile the staging was put.
//
// AB: the A operand is a BYTE plane -- a 0/1 sample or 0/1 data, one byte per element, 0x40 for one: behind a zero low byte
// that is the bf16 value 2.0, and the epilogue halves the sum (exact).  Half the bytes of the A operand from L2 and through
// LDS (with DMA staging a tile's time follows the 128-byte lines it takes in).  So that the lines stay whole, an A row in
// LDS is 128 BYTES = 128 k, an A BLOCK that serves two 64-deep k-tiles: two block buffers beside the two B stages, half a
// block requested per tile.  A byte plane is stored K-PERMUTED inside every group of 64 elements: element 32 ks + 8 s + j
// (k-step ks, lane group s of the MFMA operand, j < 8) sits at byte 16 s + 8 ks + j, so that the 16 bytes a lane needs for
// BOTH k-steps of a tile are one ds_read_b128 -- the very chunk / swizzle / bank pattern of the bf16 fragments (two
// 8-byte reads per tile fused into ds_read2_b64 and cost 27 % of the LDS cycles in bank conflicts).  The MFMA waves
// expand a k-step's 8 bytes with four v_perm_b32.
// RP: a half step that ALSO leaves the row sums of softplus(x + bias) over each tile's columns in g.rowpart [column tiles][rows]
// -- the free energy F(v) (rbm.py:73-75) of the rows whose hidden states it samples, from the same accumulators: the score of
// fit(verbose = 1) then needs three GEMMs, not four (kurbm_score_x3).
template <int BM, int BN, int WAVES_M, int WAVES_N, int BKB, int PB, int EPI, int NOISE, bool AB = false, bool RP = false>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N + 256, (64 * WAVES_M * WAVES_N + 256) / 256) void k_gemm_pb(GemmArgsB g) {
    constexpr int NT = 64 * WAVES_M * WAVES_N;          // threads of the MFMA waves (and of every epilogue loop)
    constexpr int NTS = 256;                            // threads of the loader waves
    // LDS rows are the bare 128-byte k-tile, their eight 16-byte chunks XOR-swizzled with (row >> 1) & 7.
    // ds_read_b128 is served in groups of 16 lanes that are NOT consecutive ({0-3, 12-15, 20-27}, ...;
    // MI355X_MICROARCH.md, LDS): for an MFMA fragment read (lane = row l15, k chunk `slot`) a group is 8 rows at
    // chunk c plus the 8 other rows at chunk c ^ 1, and this swizzle sends the 16 of them to 16 different
    // bank quads.  (A 144-byte padded row is conflict-free only for 16 CONSECUTIVE lanes: it cost 35 % of
    // the LDS cycles in bank conflicts.)  A row written by 8 consecutive lanes stays one 128-byte line.
    constexpr int ROWB = 2 * BKB;
    static_assert(BKB == 64, "swizzle of eight 16-byte chunks per row");
    constexpr int CPR = BKB / 8;          // 16-B chunks per row
    constexpr int KS = BKB / 32;          // MFMA k-steps per tile
    static_assert(KS == 2, "fragment buffers alternate with the k-step");
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int TM = WM / 16, TN = WN / 16;
    // ABS: the statistics GEMM of 0/1 data with BOTH A operands one byte per element -- v_pos^T fp8 (as before), v_neg^T a byte
    // plane like the half steps' A operand (32 KB per 128 k where the bf16 plane is 64: this launch is bound by the bytes a CU
    // can take in, ~34 per cycle).  Only the fixed walk fp8 / 3-piece / 3-piece (g.walk3); the fp8 tile is scaled x 2 through
    // its E8M0 exponent so that the slab epilogue's halving (0x40 bytes are 2.0) leaves it alone.
    constexpr bool ABS = AB && EPI == EPI_SLAB && !RP;
    static_assert(!ABS || PB == 3, "byte-plane statistics: three-piece negative half");
    // ABP (template slot RP on EPI_SLAB): a statistics GEMM whose A operand is ONE byte plane and whose k range is one segment of
    // three-piece tiles -- the plain byte-plane walk of the half steps with the slab epilogue, any number of k slices of whole
    // 128-deep blocks.  Two launches of real-valued data use it: the NEGATIVE statistics of Bernoulli visibles (A = v_neg^T, 0/1;
    // B = the pieces of -h_neg^T) and the POSITIVE statistics as the transposed problem (A = h_pos^T, 0/1; B = the three pieces of
    // v_pos^T: one 40-KB tile per k position where the untransposed walk staged three 40-KB tiles of one piece each), whose tile
    // leaves TRANSPOSED (g.slab_t) so that its slabs look like everyone else's.
    constexpr bool ABP = AB && EPI == EPI_SLAB && RP;
    static_assert(!ABP || PB == 3, "plain byte-plane statistics: three-piece tiles");
    constexpr int A_BYTES = BM * ROWB, B1_BYTES = BN * ROWB, B_BYTES = PB * B1_BYTES;   // (AB: A_BYTES is a BLOCK, 128 k deep)
    // DEEP: one-piece tiles of bytes (the rounded-bf16 path's half steps on 0/1 states) -- 16 MFMAs per wave and tile, a third
    // of a DMA round trip -- run a deeper pipeline than the three-piece tiles have room for: FOUR B stages and THREE A blocks,
    // tiles requested four ahead, fragments read two micro-steps (= one tile) ahead ACROSS the tile's barrier (see `deep_tile`)
    constexpr bool DEEP = AB && PB == 1;
    constexpr int NAB = (DEEP || ABS) ? 3 : 2;     // A block buffers (AB)
    // LDS: two stages [A tile | B pieces]; AB: the A blocks, then the stages of B pieces
    constexpr int STAGE = AB ? B_BYTES : A_BYTES + B_BYTES;
    constexpr int B_OFF = AB ? NAB * A_BYTES : A_BYTES;
    constexpr int NA = BM * CPR / NTS, NB1 = BN * CPR / NTS;   // 1-KiB pieces per loader wave: A tile, ONE piece of B
    static_assert((BM * CPR) % NTS == 0 && (BN * CPR) % NTS == 0 && NA % 2 == 0, "whole pieces per loader wave");
    constexpr int PROW16 = 2 * BN + 16;   // bf16 patch row (bytes)
    constexpr int PROW32 = 4 * BN + 16;   // fp32 patch row (bytes)
    constexpr int PROWT = 4 * BM + 16;    // fp32 patch row of a tile that leaves transposed (ABP, g.slab_t): BN rows of BM floats
    constexpr int PATCH_BYTES0 = (EPI == EPI_SOFTPLUS) ? WAVES_N * BM * 4 : BM * (EPI == EPI_SLAB ? PROW32 : PROW16);
    constexpr int PATCH_BYTES = (ABP && BN * PROWT > PATCH_BYTES0) ? BN * PROWT : PATCH_BYTES0;
    // One-piece tiles (the rounded-bf16 path) are 16 MFMAs per wave, shorter than the round trip of a tile's DMA: THREE
    // stages, tiles requested two ahead.  (Three-piece tiles: two stages fill the LDS.)
    constexpr int NSTG = DEEP ? 4 : (PB == 1) ? 3 : 2;
    // (AB: NAB A blocks of 128 k beside NSTG stages of B pieces, the A blocks requested half a block per tile)
    // (byte-plane statistics: three A blocks, a B stage for each of the two 3-piece tiles of a unit and one PIECE for its fp8 tile)
    constexpr int STAGES_BYTES = ABS ? NAB * A_BYTES + 2 * B_BYTES + B1_BYTES
                               : AB ? NAB * A_BYTES + NSTG * B_BYTES : NSTG * (A_BYTES + B_BYTES);
    // BSP (g.bshare == 2; 128 x 128 tiles only): the PAIRED walk of a real-valued A operand -- per k position TWO tiles of 48 MFMAs per
    // wave each, (A piece 0) x (B pieces 0-2) and then (piece 1) x (0-1) together with (piece 2) x (0) -- instead of three tiles of
    // 48 / 32 / 16: a tile costs its barrier and its restart whatever is in it (one-piece tiles of 512 matrix-pipe
    // cycles took ~1 500).  The second tile reads TWO A tiles, so the A tiles ring through FOUR stages (16 KB each on this tiling)
    // beside the two B stages: 160 KB.  The half steps on grey-level data / Gaussian visibles, and the negative statistics of
    // Gaussian visibles (EPI_SLAB) walk it.
    constexpr bool BSP = (BM == 128) && !AB && PB == 3 && (EPI == EPI_HALFSTEP || EPI == EPI_SLAB);
    constexpr int BSP_BYTES = 4 * A_BYTES + 2 * B_BYTES, BSP_BOFF = 4 * A_BYTES;
    constexpr int SMEM_BYTES00 = (STAGES_BYTES > PATCH_BYTES) ? STAGES_BYTES : PATCH_BYTES;
    constexpr int SMEM_BYTES = (BSP && BSP_BYTES > SMEM_BYTES00) ? BSP_BYTES : SMEM_BYTES00;
    // the Philox words of each lane's first NI_LDS output columns are drawn in the prologue (the MFMA waves idle there
    // until the first tile has landed) and wait in LDS behind the stage buffers, 16 bytes per lane and 4-row group
    // (none beside the paired walk's layout: all 160 KB)
    constexpr int NI_LDS = (EPI == EPI_HALFSTEP && NOISE == NOISE_BERNOULLI && !BSP) ? 1 : 0;
    constexpr int DRAW_LDS_BYTES = NI_LDS * TM * NT * 16;
    static_assert(SMEM_BYTES + DRAW_LDS_BYTES <= 160 * 1024, "LDS per workgroup");
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_BYTES + DRAW_LDS_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave >= NT / 64;                // wave-uniform role
    // Every wave runs the same ~600 instructions of set-up (block mapping, operand offsets, descriptors) before the roles part,
    // three waves to a SIMD; what the launch waits for is the loaders' first requests.  They go first until those are out.
    if (KURBM_PRIO_ENTRY && loader) __builtin_amdgcn_s_setprio(KURBM_PRIO_ENTRY);
    const int stid = tid & 255;                         // loader thread id
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int l15 = lane & 15, slot = lane >> 4;
#ifdef KURBM_STAMPS
    unsigned long long ts[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long tu[6] = {0, 0, 0, 0, 0, 0};   // cycles per micro-step of the 3-piece tiles, summed over tiles
    unsigned long long tk[3] = {0, 0, 0};            // cycles per KIND of tile, summed: fp8, one piece, three pieces
    KURBM_STAMP(ts[0]);
#define KURBM_STAMP_OUT()                                                                         \
    do {                                                                                          \
        KURBM_STAMP(ts[5]);                                                                       \
        if (g.stamps && lane == 0 && wave != 7) {                                                 \
            unsigned long long* o = g.stamps + ((size_t)blockIdx.x * (NT / 64) + wave) * 16;      \
            for (int q = 0; q < 6; ++q) o[q] = ts[q];                                             \
            for (int q = 0; q < 6; ++q) o[8 + q] = tu[q];                                         \
            o[6] = tk[0]; o[7] = tk[1]; o[14] = tk[2];                                            \
        }                                                                                         \
    } while (0)
    // the first loader wave's loop, summed over its tiles: issue of a tile's DMA / wait for it to land / wait at the barrier;
    // written where MFMA wave 7 would write (that wave's stamps are dropped)
    unsigned long long tl[4] = {0, 0, 0, 0}, tls[3] = {0, 0, 0};
    // ... and its way to the first barrier (absolute): setup done / the prologue's requests issued / tile 0 has landed
    unsigned long long tp[6] = {0, 0, 0, 0, 0, 0};   // (3..5: arguments warm / block mapped / the role branch reached)
#define KURBM_PSTAMP(q) KURBM_STAMP(tp[q])
#define KURBM_LSTAMP(q)                                                   \
    do {                                                                  \
        KURBM_STAMP(tl[q]);                                               \
        if ((q) >= 1) tls[(q) - 1] += tl[q] - tl[(q) - 1];                \
    } while (0)
#define KURBM_LSTAMP_OUT()                                                                        \
    do {                                                                                          \
        if (g.stamps && lane == 0 && wave == NT / 64) {                                           \
            unsigned long long* o = g.stamps + ((size_t)blockIdx.x * (NT / 64) + 7) * 16;         \
            o[0] = 1; o[1] = tls[0]; o[2] = tls[1]; o[3] = tls[2];                                \
            o[4] = ts[0]; o[5] = tp[0]; o[6] = tp[1]; o[7] = tp[2];                               \
            o[8] = tp[3]; o[9] = tp[4]; o[10] = tp[5];                                            \
        }                                                                                         \
    } while (0)
#else
#define KURBM_STAMP_OUT() do { } while (0)
#define KURBM_LSTAMP(q) do { } while (0)
#define KURBM_PSTAMP(q) do { } while (0)
#define KURBM_LSTAMP_OUT() do { } while (0)
#endif

    warm_kernel_arguments<sizeof(GemmArgsB)>();
    KURBM_PSTAMP(3);
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    int z, bm, bn;
    if (g.xcd_r) {
        // workgroup i runs on XCD i % 8: XCD (xz, xr, xc) owns k slices [xz zl, +zl), row tiles [xr rl, +rl), column tiles [xc cl, +cl)
        const int xcd = bid & 7, idx = bid >> 3;
        const int xz = xcd >> g.map_lrc, xr = (xcd & ((1 << g.map_lrc) - 1)) >> g.map_lxc, xc = xcd & ((1 << g.map_lxc) - 1);
        const int rl = g.map_rl, cl = g.map_cl;
        const int zi = div_map(idx, rl * cl, g.map_inv_a, g.map_slow), t2 = idx - zi * (rl * cl);
        const int q2 = div_map(t2, g.m_fastest ? rl : cl, g.map_inv_b, g.map_slow);   // t2 / rl (m fastest) or t2 / cl
        const int ri = g.m_fastest ? t2 - q2 * rl : q2;
        const int ci = g.m_fastest ? q2 : t2 - q2 * cl;
        z = xz * g.map_zl + zi; bm = xr * rl + ri; bn = xc * cl + ci;
    } else {
        {   // XCD-aware bijective remap (see kurbm_kernels.hip)
            const int xcd = bid & 7, q = nwg >> 3, rr = nwg & 7;
            bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
        }
        const int tiles_mn = g.grid_m * g.grid_n;
        z = div_map(bid, tiles_mn, g.map_inv_a, g.map_slow);
        const int tmn = bid - z * tiles_mn;
        const int q2 = div_map(tmn, g.m_fastest ? g.grid_m : g.grid_n, g.map_inv_b, g.map_slow);   // tmn / grid_m (m fastest) or tmn / grid_n
        bm = g.m_fastest ? tmn - q2 * g.grid_m : q2;
        bn = g.m_fastest ? q2 : tmn - q2 * g.grid_n;
    }
    const int m0 = bm * BM, n0 = bn * BN;
    KURBM_PSTAMP(4);

    const int t_begin = z * g.kt_per_split;
    int t_end = t_begin + g.kt_per_split;
    if (t_end > g.kt_total) t_end = g.kt_total;
    // (a half-step tile that lies wholly in the column padding of its row-major plane has nothing to multiply: it only
    //  writes that plane's zeros)
    const int nt = (t_end > t_begin && !(EPI == EPI_HALFSTEP && n0 >= g.N)) ? t_end - t_begin : 0;

    // Staging map of the loader waves.  A wave instruction of LDS-DMA writes 1 KiB of LDS linearly (lane l -> bytes
    // 16 l ...): piece q = 8 rows of a tile, lane l fills chunk slot l % 8 of row 8 q + l / 8.  The swizzle therefore
    // sits on the SOURCE: the lane that fills slot ch of a row fetches chunk ch ^ key(row) of that row, the same
    // involution as the fragment reads.  Rows outside the matrix are pointed at row 0 (they only feed outputs that
    // are never stored).  Loads are buffer loads: a per-lane BYTE offset that never changes (voffset) plus a per-tile
    // scalar offset (soffset) against one descriptor per operand -- no per-tile vector address arithmetic at all.
    // ATR (g.a_tr; the paired walk of the statistics GEMM only): the A operand is read from a ROW-MAJOR plane [k][M] -- the pieces of
    // v_neg as the h -> v half step leaves them for the next half step -- so that no transposed copy of them need be written (19 MB per
    // Gaussian step).  The A tile then sits in LDS as 64 k rows of 256 bytes (128 m), chunk c of row r at c ^ (((r & 3) << 2) |
    // ((r >> 2) & 3)) (cdna_hip_programming.md T10, image (b)), and the MFMA waves fetch their fragments with ds_read_b64_tr_b16: a lane
    // needs 8 consecutive k of ONE m, i.e. a column of that image.
    constexpr bool ATR = BSP && EPI == EPI_SLAB;
    const bool a_tr = ATR && g.a_tr != 0;
    unsigned goffA[NA], goffB[NB1];
#pragma unroll
    for (int it = 0; it < NA; ++it) {
        const int q = it * NTS + stid, row = q / CPR, ch = q % CPR;
        const int x = (m0 + row < g.M) ? m0 + row : 0;
        // (a byte plane has lda BYTES between its rows; its 128-byte row piece is 128 k)
        goffA[it] = AB ? (unsigned)(x * g.lda + 16 * (ch ^ ((row >> 1) & 7))) : 2u * (unsigned)(x * g.lda + 8 * (ch ^ ((row >> 1) & 7)));
        if (a_tr) {   // row = k index inside the tile (16 chunks of 8 m per row); rows and columns past the matrix are the plane's zero padding
            const int rt = q >> 4, ct = q & 15;
            goffA[it] = 2u * (unsigned)(rt * g.lda + m0 + 8 * (ct ^ (((rt & 3) << 2) | ((rt >> 2) & 3))));
        }
    }
#pragma unroll
    for (int it = 0; it < NB1; ++it) {
        const int q = it * NTS + stid, row = q / CPR, ch = q % CPR;
        const int x = (n0 + row < g.N) ? n0 + row : 0;
        goffB[it] = 2u * (unsigned)(x * g.ldb + 8 * (ch ^ ((row >> 1) & 7)));
    }
    typedef __amdgpu_buffer_rsrc_t rsrc_t;
    // B: one descriptor for both operand sets (they live in one workspace), a set is a scalar offset.  A: one descriptor
    // PER SET -- set 0 of the statistics GEMM (the v_pos planes) may sit in the caller's resident data planes, any distance
    // from the workspace that holds set 1 -- picked per tile by a scalar select.
    [[maybe_unused]] const rsrc_t dA0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(g.A0), 0, 0xFFFFFFFF, 0x00020000);
    [[maybe_unused]] const rsrc_t dA1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(g.A1 ? g.A1 : g.A0), 0, 0xFFFFFFFF, 0x00020000);
    [[maybe_unused]] const rsrc_t dB = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(g.baseB), 0, 0xFFFFFFFF, 0x00020000);

    // a tile: the scalar byte offsets of its A rows and of piece 0 of its B rows
    struct TileRef { uint32_t oa, ob, bplane, oa_piece; int npb; bool neg, f8; };
    constexpr bool F8 = (EPI == EPI_SLAB && PB == 3);   // the kernel that may meet fp8 tiles (g.f8pos)
    // The tiles of a k slice are walked in order, so a tile is a POSITION: `a` inside a period of `wl_len` tiles, `b` = which
    // period.  Three walks: segment-major (half steps: a = k-tile, b = segment), segment-fastest (statistics: a = segment,
    // b = k-tile), and with fp8 positive statistics units of 128 k (a = 0: the fp8 tile of segment 0, then two 64-deep tiles of
    // every other segment; b = unit).  What a segment contributes -- plane offsets, operand set, piece count -- sits in a table
    // with one LANE per segment (four registers) and comes out by v_readlane: a step of the walk is a compare and an add, a tile
    // reference five scalar instructions.  (Until round 3 every tile was decoded from its index in BOTH loops -- two multiply-
    // highs, 64-bit shifts, a dozen kernel arguments kept live in scalar registers that then spilled: a sixth of the k loop.)
    struct Walk { int a, b; };
    const bool wl_f8 = F8 && g.f8pos, wl_sf = g.seg_fastest != 0;
    const int wl_len = wl_f8 ? 2 * g.nseg - 1 : wl_sf ? g.nseg : g.nkt;
    uint32_t t_code, t_oa, t_ob, t_bp;
    {
        const int sl = lane < g.nseg ? lane : 0;
        t_code = (uint32_t)(g.seg_codes >> (5 * sl)) & 31u;
        const bool ng = (t_code & 16u) != 0u;
        t_oa = 2u * (t_code & 3u) * (uint32_t)(ng ? g.a_plane1 : g.a_plane0);
        t_ob = ng ? g.offB1 : g.offB0;
        t_bp = 2u * (uint32_t)(ng ? g.b_plane1 : g.b_plane0);
    }
    auto walk_at = [&](int t) __attribute__((always_inline)) {   // (one multiply-high: outside the loops)
        const uint32_t inv = (wl_f8 || wl_sf) ? g.inv_nseg : g.inv_nkt;      // (0: the period is one tile long)
        Walk w;
        w.b = inv ? (int)__umulhi((uint32_t)t, inv) : t;
        w.a = t - w.b * wl_len;
        return w;
    };
    auto walk_next = [&](Walk& w) __attribute__((always_inline)) {
        if (++w.a == wl_len) { w.a = 0; ++w.b; }
    };
    auto walk_ref = [&](const Walk& w) __attribute__((always_inline)) {
        TileRef r;
        int seg, kt;
        bool f8 = false;
        if (wl_f8) {
            f8 = (w.a == 0);
            seg = f8 ? 0 : 1 + ((w.a - 1) >> 1);
            kt = f8 ? w.b : 2 * w.b + ((w.a - 1) & 1);
        } else if (wl_sf) {
            seg = w.a; kt = w.b;
        } else {
            kt = w.a; seg = w.b;
        }
        const uint32_t code = __builtin_amdgcn_readlane(t_code, seg);
        const uint32_t k0 = 2u * (uint32_t)(kt * BKB);   // (128 bytes per k-tile, bf16 or fp8)
        r.oa = __builtin_amdgcn_readlane(t_oa, seg) + k0;                    // (AB: unused)
        r.oa_piece = code & 3u;                                              // (diagnostic builds)
        r.neg = (code & 16u) != 0u;
        r.f8 = f8;
        r.ob = __builtin_amdgcn_readlane(t_ob, seg) + k0;
        r.bplane = __builtin_amdgcn_readlane(t_bp, seg);
        r.npb = (int)((code >> 2) & 3u);
        return r;
    };
    auto tile_of = [&](int t) __attribute__((always_inline)) { return walk_ref(walk_at(t)); };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    typedef u32x4 afrag;   // (AB: fa[0] = the 16 bytes of BOTH k-steps of a tile, expanded per k-step into fx; fa[1] unused)
    afrag fa[2][TM];
    u32x4 fb[3][TN];   // fragment buffers: A by k-step; B by micro-step (three: read two steps ahead)
    // fragment rows are l15 + a multiple of 16, so their swizzle key is (l15 >> 1) & 7
    const int swz = (l15 >> 1) & 7;
    // AB: `ablk` = the block buffer (0 / 1) of the tile pair in work; tile `buf` (= its parity) is half `buf` of the block's k
    int ablk = 0;
    const bool bsp = BSP && g.bshare == 2;     // (wave-uniform) the paired walk: A stages and B stages apart
    auto frag_a = [&](int buf, int ks, afrag (&f)[TM], int blk_step = 0) __attribute__((always_inline)) {
        if constexpr (AB) {   // tile `buf` of the block, lane group `slot`: chunk 4 buf + slot of the 128-byte row (k-permuted plane)
            if (ks == 0) {
                const unsigned char* c = smem + ((ablk + blk_step) & 1) * A_BYTES + (wm * WM + l15) * ROWB + 16 * ((4 * buf + slot) ^ swz);
#pragma unroll
                for (int mi = 0; mi < TM; ++mi) f[mi] = *reinterpret_cast<const u32x4*>(c + mi * 16 * ROWB);
            }
        } else {
            if constexpr (ATR) {
                if (a_tr) {
#if defined(__HIP_DEVICE_COMPILE__)
                    // lane i of a 16-lane group supplies the address of row q = i >> 2, columns 4 (i & 3) .. + 3 of a 4-row block and
                    // receives column i of the four rows: two blocks (k = 32 ks + 8 slot + 0..3, + 4..7) are a lane's 8 k
                    typedef short v4s __attribute__((ext_vector_type(4)));
                    typedef __attribute__((address_space(3))) v4s* lds_v4s;
                    const int q = l15 >> 2, pp = l15 & 3;
                    const unsigned char* tb = smem + buf * A_BYTES;
#pragma unroll
                    for (int mi = 0; mi < TM; ++mi) {
                        uint32_t w[4];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int row = 32 * ks + 8 * slot + 4 * h + q;
                            const int key = (q << 2) | ((2 * slot + h) & 3);
                            const int ch = (2 * (wm * TM + mi) + (pp >> 1)) ^ key;
                            const v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(tb + 256 * row + 16 * ch + 8 * (pp & 1)));
                            w[2 * h] = ((const uint32_t*)&r)[0]; w[2 * h + 1] = ((const uint32_t*)&r)[1];
                        }
                        f[mi] = u32x4{w[0], w[1], w[2], w[3]};
                    }
#endif
                    return;
                }
            }
            const unsigned char* c = smem + (bsp ? buf * A_BYTES : buf * STAGE) + (wm * WM + l15) * ROWB + 16 * ((4 * ks + slot) ^ swz);
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) f[mi] = *reinterpret_cast<const u32x4*>(c + mi * 16 * ROWB);
        }
    };
    auto frag_b = [&](int buf, int ks, int p, u32x4 (&f)[TN]) __attribute__((always_inline)) {
        const unsigned char* c = smem + (bsp ? BSP_BOFF + buf * B_BYTES : buf * STAGE + B_OFF) + p * B1_BYTES + (wn * WN + l15) * ROWB +
                                 16 * ((4 * ks + slot) ^ swz);
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) f[ni] = *reinterpret_cast<const u32x4*>(c + ni * 16 * ROWB);
    };
    // AB: the fragments of the k-step in use, expanded (byte b -> bf16 b << 8: v_perm_b32 picks {byte, 0} pairs) once per
    // k-step, behind the last MFMAs of the step before (so the permutes run while the matrix pipe drains those); raw
    // double buffer + this = the registers of the bf16 fragments
    [[maybe_unused]] u32x4 fx[AB ? TM : 1];
    auto expand_a = [&](int ks) __attribute__((always_inline)) {
        if constexpr (AB) {
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) {
                const uint32_t lo = ks ? fa[0][mi].z : fa[0][mi].x, hi = ks ? fa[0][mi].w : fa[0][mi].y;
                fx[mi] = u32x4{__builtin_amdgcn_perm(0u, lo, 0x010C000Cu), __builtin_amdgcn_perm(0u, lo, 0x030C020Cu),
                               __builtin_amdgcn_perm(0u, hi, 0x010C000Cu), __builtin_amdgcn_perm(0u, hi, 0x030C020Cu)};
            }
        }
    };
    auto mfmas = [&](const afrag (&a)[TM], const u32x4 (&b)[TN]) __attribute__((always_inline)) {
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
            u32x4 am;
            if constexpr (AB) am = fx[mi];
            else am = a[mi];
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8, am), __builtin_bit_cast(bf16x8, b[ni]), acc[mi][ni], 0, 0, 0);
        }
    };

    // One tile with NPB pieces = KS * NPB micro-steps (k-step ks = u / NPB, piece p = u % NPB).  Every micro-step's
    // fragment reads are issued one step (three-piece tiles: two steps) ahead of the MFMAs that use them; the last
    // micro-step holds the tile's ONLY barrier (behind it the loaders have the next tile in the other LDS buffer), then
    // reads the NEXT tile's first fragments.
    // (cur = the tile's B stage; half = its half of the A block, AB only -- with two stages the two coincide)
    // (bcur / anext_ / bnext_ >= 0: the B stage of this tile and the A / B stages of the next one, where they are not `cur` and
    //  `cur + 1` -- the shared-B walk)
    auto one_tile = [&](const int cur, const int half, auto npb_tag, const int bcur_ = -1, const int anext_ = -1,
                        const int bnext_ = -1) __attribute__((always_inline)) {
        const int acur = AB ? half : cur, anext = anext_ >= 0 ? anext_ : AB ? (half ^ 1) : (cur + 1) % NSTG;
        const int bcur = bcur_ >= 0 ? bcur_ : cur, bnext = bnext_ >= 0 ? bnext_ : (cur + 1) % NSTG;
        constexpr int NPB = decltype(npb_tag)::value;
        constexpr int NU = KS * NPB;
#ifdef KURBM_STAMPS
        unsigned long long tq[7];
        KURBM_STAMP(tq[0]);
#endif
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            __builtin_amdgcn_sched_barrier(0);
            const int ks = u / NPB;
            if constexpr (NPB == 3) {
                // fragments are read TWO micro-steps ahead (an LDS read under load takes longer than the 8 MFMAs
                // of one micro-step); the tile is entered with fa[0], fb[0] loaded, so u = 0 catches up
                if (u == 0) { frag_b(bcur, 0, 1, fb[1]); frag_b(bcur, 0, 2, fb[2]); }
                if (u == 1) { frag_a(acur, 1, fa[1]); frag_b(bcur, 1, 0, fb[0]); }
                if (u == 2) frag_b(bcur, 1, 1, fb[1]);
                if (u == 3) frag_b(bcur, 1, 2, fb[2]);
                // The tile's ONLY barrier sits in front of its last micro-step (behind it the loaders may refill this
                // stage), and the next tile's first fragments are read right behind it, under the MFMAs of micro-step 5
                // instead of in front of an idle matrix pipe.  (The barrier behind micro-step 3: no faster, LABBOOK.)
                if (u == NU - 1) {
                    __syncthreads();
                    __builtin_amdgcn_sched_barrier(0);
                    frag_a(anext, 0, fa[0], acur);   // (AB: after the second tile of a pair comes the other block)
                    frag_b(bnext, 0, 0, fb[0]);
                }
                mfmas(fa[ks & 1], fb[u % 3]);
                if (AB && (u + 1) % NPB == 0) {   // the NEXT k-step's, behind this step's MFMAs (same registers: not among them)
                    __builtin_amdgcn_sched_barrier(0);
                    expand_a(((u + 1) / NPB) & 1);
                }
#ifdef KURBM_STAMPS
                KURBM_STAMP(tq[u + 1]);
                tu[u] += tq[u + 1] - tq[u];
#endif
                continue;
            }
            if (u + 1 < NU) {
                const int ksn = (u + 1) / NPB, pn = (u + 1) % NPB;
                if (pn == 0) frag_a(acur, ksn, fa[ksn & 1]);
                frag_b(bcur, ksn, pn, fb[(u + 1) & 1]);
            } else {
                __syncthreads();
                __builtin_amdgcn_sched_barrier(0);
                frag_a(anext, 0, fa[0], acur);   // (AB: after the second tile of a pair comes the other block)
                frag_b(bnext, 0, 0, fb[0]);
            }
            mfmas(fa[ks & 1], fb[u & 1]);
            if (AB && (u + 1) % NPB == 0) {
                __builtin_amdgcn_sched_barrier(0);
                expand_a(((u + 1) / NPB) & 1);
            }
#ifdef KURBM_STAMPS
            KURBM_STAMP(tq[u + 1]);
            if (NPB == 3) tu[u] += tq[u + 1] - tq[u];
#endif
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // The second tile of a k position on the paired walk (BSP): (A piece 1) x (B pieces 0, 1) and (A piece 2) x (B piece 0), 2 x 3 micro-
    // steps of 8 MFMAs like a three-piece tile.  Entered with fa[0] = A1 and fb[0] = B0 of k-step 0 (the tile in front read them
    // behind its barrier); A1 alternates between fa[0] / fa[1] with the k-step, A2 has a set of its own (the k loop has the
    // registers: accumulators 32 + fragments 56 + 16), B rotates through the three fb sets.  Every read is two micro-steps ahead of
    // its MFMAs except the two that catch up at u = 0; the tile's ONE barrier in front of the last micro-step, as in one_tile.
    [[maybe_unused]] afrag fa2[BSP ? TM : 1];
    auto pair_tile = [&](const int a1, const int a2, const int bcur, const int anext, const int bnext) __attribute__((always_inline)) {
        if constexpr (BSP) {
            __builtin_amdgcn_sched_barrier(0);
            frag_b(bcur, 0, 1, fb[1]); frag_a(a2, 0, fa2);
            mfmas(fa[0], fb[0]);                                   // A1 . B0, k-step 0
            __builtin_amdgcn_sched_barrier(0);
            frag_a(a1, 1, fa[1]); frag_b(bcur, 1, 0, fb[2]);
            mfmas(fa[0], fb[1]);                                   // A1 . B1
            __builtin_amdgcn_sched_barrier(0);
            frag_b(bcur, 1, 1, fb[1]);
            mfmas(fa2, fb[0]);                                     // A2 . B0
            __builtin_amdgcn_sched_barrier(0);
            frag_a(a2, 1, fa2);
            mfmas(fa[1], fb[2]);                                   // A1 . B0, k-step 1
            __builtin_amdgcn_sched_barrier(0);
            mfmas(fa[1], fb[1]);                                   // A1 . B1
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
            frag_a(anext, 0, fa[0]);
            frag_b(bnext, 0, 0, fb[0]);
            mfmas(fa2, fb[2]);                                     // A2 . B0
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // An fp8 tile (statistics of 0/1 data, segment 0): the same 128-byte rows hold 128 k.  Lane group g = lane >> 4 of
    // v_mfma_scale_f32_16x16x128_f8f6f4 supplies 32 bytes of its row; which 32 is free as long as A and B agree (k is a
    // summation index), so it takes chunks g and 4 + g -- exactly the two fragment reads of a bf16 tile, and the tile is
    // entered like any other with fa[0], fb[0] loaded.  Scale operands 0x7F = 2^0 (E8M0).
    auto f8_tile = [&](const int cur) __attribute__((always_inline)) {
        if constexpr (F8 && !AB) {
            typedef int i32x8 __attribute__((ext_vector_type(8)));
            __builtin_amdgcn_sched_barrier(0);
            frag_a(cur, 1, fa[1]);
            frag_b(cur, 1, 0, fb[1]);
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni) {
                    const i32x8 a = {(int)fa[0][mi].x, (int)fa[0][mi].y, (int)fa[0][mi].z, (int)fa[0][mi].w,
                                     (int)fa[1][mi].x, (int)fa[1][mi].y, (int)fa[1][mi].z, (int)fa[1][mi].w};
                    const i32x8 b = {(int)fb[0][ni].x, (int)fb[0][ni].y, (int)fb[0][ni].z, (int)fb[0][ni].w,
                                     (int)fb[1][ni].x, (int)fb[1][ni].y, (int)fb[1][ni].z, (int)fb[1][ni].w};
                    acc[mi][ni] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[mi][ni], 0, 0, 0, 0x7F, 0, 0x7F);
                }
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
            frag_a((cur + 1) % NSTG, 0, fa[0], cur);
            frag_b((cur + 1) % NSTG, 0, 0, fb[0]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto tile_any = [&](const int cur, const int half, const TileRef& r) __attribute__((always_inline)) {
#ifdef KURBM_STAMPS
        unsigned long long tk0, tk1;
        KURBM_STAMP(tk0);
#endif
        if (F8 && r.f8) f8_tile(cur);   // (never with byte A planes)
        else if (PB >= 3 && r.npb == 3) one_tile(cur, half, std::integral_constant<int, 3>{});
        else if (PB >= 2 && r.npb == 2) one_tile(cur, half, std::integral_constant<int, 2>{});
        else one_tile(cur, half, std::integral_constant<int, 1>{});
#ifdef KURBM_STAMPS
        KURBM_STAMP(tk1);
        tk[(F8 && r.f8) ? 0 : (r.npb == 3 ? 2 : 1)] += tk1 - tk0;
#endif
    };

    // The Philox words of this lane's outputs do not depend on the GEMM: those of its first NI_LDS output columns are drawn
    // in the prologue and parked in LDS (three waves per SIMD leave 168 registers: none to hold draws over the loop), the
    // rest in the epilogue.
    uint32_t draws[(NOISE != NOISE_NONE) ? TM * TN : 1][4];
    auto draw_cols = [&](int ni0, int ni1) __attribute__((always_inline)) {
        if (NOISE == NOISE_NONE) return;
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) {
                if (ni < ni0 || ni >= ni1) continue;
                const uint64_t grow = g.rng.row0 + (uint64_t)(m0 + wm * WM + slot * 4 + mi * 16);
                uint32_t (&w)[4] = draws[(NOISE != NOISE_NONE) ? ni * TM + mi : 0];
                philox4x32_10((uint32_t)(n0 + wn * WN + l15 + ni * 16), (uint32_t)(grow >> 2), g.rng.stream_id, g.rng.step,
                              g.rng.seed_lo, g.rng.seed_hi, w);
            }
    };
    float biasv[TN];   // loaded here too: in the epilogue its latency would be exposed
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
        const int c = n0 + wn * WN + l15 + ni * 16;
        biasv[ni] = (EPI == EPI_HALFSTEP && c < g.N) ? g.bias[c] : 0.f;
    }
    if (loader) {   // (NOT marked likely: the loaders would reach their first request ~1 000 cycles sooner, but hipcc then treats the
                    //  MFMA waves' path as cold and their k loop takes 1.5 x the time)
        KURBM_PSTAMP(5);
        // ---- loader waves: a loop of their own; one barrier per tile, like the MFMA waves.  `buffer_load_dwordx4 ... lds`,
        // one 1-KiB piece (8 rows of a tile) per wave instruction, straight into the stage that the barrier before has
        // freed; the wait for a tile's pieces and then the tile's barrier make them visible to the MFMA waves
        // (cdna_hip_programming.md 5, "Read a staged buffer one phase AFTER the wait that retires it").
        if (nt > 0) {
            // part: bit 0 = the A tile (AB: pieces [a_lo, a_hi) of A block `ablk_`, whose k offset is r.oa), bit 1 = the B pieces
            auto dma_part = [&](int buf, const TileRef& r, int part, int ablk_ = 0, int a_lo = 0, int a_hi = 1 << 20, int pb_lo = 0,
                                int pb_hi = PB) __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)   // (the host pass drops a kernel's launch stub over the LDS address-space cast)
                typedef __attribute__((address_space(3))) void* lds_ptr;
                const int lw = wave - NT / 64;
                if (part & 1) {
                    unsigned char* a = smem + (AB ? ablk_ * A_BYTES : bsp ? buf * A_BYTES : buf * STAGE) + lw * 8 * ROWB;   // piece it * 4 + lw
                    if (r.neg) {   // (wave-uniform)
#pragma unroll
                        for (int it = 0; it < NA; ++it)
                            if (it >= a_lo && it < a_hi)
                                __builtin_amdgcn_raw_ptr_buffer_load_lds(dA1, (lds_ptr)(a + it * 32 * ROWB), 16, goffA[it], r.oa, 0, 0);
                    } else {
#pragma unroll
                        for (int it = 0; it < NA; ++it)
                            if (it >= a_lo && it < a_hi)
                                __builtin_amdgcn_raw_ptr_buffer_load_lds(dA0, (lds_ptr)(a + it * 32 * ROWB), 16, goffA[it], r.oa, 0, 0);
                    }
                }
                if (part & 2) {
                    unsigned char* b = smem + (bsp ? BSP_BOFF + buf * B_BYTES : buf * STAGE + B_OFF) + lw * 8 * ROWB;
#pragma unroll
                    for (int p = 0; p < PB; ++p) {
                        if (p >= r.npb) break;   // (wave-uniform)
                        if (p < pb_lo || p >= pb_hi) continue;
                        const uint32_t so = r.ob + (uint32_t)p * r.bplane;
#pragma unroll
                        for (int it = 0; it < NB1; ++it)
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(dB, (lds_ptr)(b + p * B1_BYTES + it * 32 * ROWB), 16, goffB[it], so, 0, 0);
                    }
                }
#endif
            };
            auto dma_tile = [&](int buf, const TileRef& r) __attribute__((always_inline)) { dma_part(buf, r, 3); };
            // raw barriers and explicit waits: `s_waitcnt` with expcnt / lgkmcnt not waited for, vmcnt in bits 3:0 and 15:14
            constexpr int VM0 = 0x0F70;
            auto vm = [](int n) constexpr { return VM0 | (n & 15) | ((n >> 4) << 14); };
            if constexpr (DEEP) {
                // Tile t = B tile t in stage t % 4 + its share of the A blocks: block j (tiles 2 j, 2 j + 1) in buffer j % 3, its
                // first half of rows requested with tile 2 j - 1, the second with tile 2 j -- in FRONT of that tile's B pieces, so
                // "tile t has landed" (a counted vmcnt: NPT pieces per wave and tile) covers everything tile t reads.
                // Iteration i runs behind barrier i - 1, where the MFMA waves have READ all of tile i (they read one tile ahead):
                // tile i + 4 goes into tile i's stage, and block (i + 5) / 2 into the buffer of the block tile i was the last of.
                // Barrier i promises tile i + 2; tiles i + 3 and i + 4 stay in flight across it.
                constexpr int NPT = NA / 2 + NB1;
                static_assert(3 * NPT <= 63, "vmcnt");
                // ONE segment (launch_pb checks): tile t lies 128 t bytes along k in the B operand, block j 128 j bytes in the A
                // plane -- running scalars, no tile_of in the loop: the loaders' issue is this loop's critical path
                TileRef rb = tile_of(t_begin);
                TileRef ra = rb;
                ra.neg = false;
                const uint32_t ob0 = rb.ob, oa0 = 128u * (uint32_t)(t_begin >> 1);
                const int jb_last = (nt - 1) >> 1;
                // tile t: half (t odd: first, t even: second) of the rows of block jb = (t + 1) / 2 into A buffer ab = jb % 3,
                // then the B pieces into stage t % 4
                auto issue = [&](int t, int jb, int ab) __attribute__((always_inline)) {
                    // (past the last block: the same count of pieces from the last block -- the waits stay exact)
                    ra.oa = oa0 + 128u * (uint32_t)(jb < jb_last ? jb : jb_last);
                    if (t & 1) dma_part(0, ra, 1, ab, 0, NA / 2);
                    else dma_part(0, ra, 1, ab, NA / 2, NA);
                    rb.ob = ob0 + 128u * (uint32_t)t;
                    dma_part(t & 3, rb, 2);
                };
                ra.oa = oa0;
                KURBM_PSTAMP(0);
                dma_part(0, ra, 3, 0);                                  // block 0 whole + B tile 0
                const int npro = nt < 4 ? nt : 4;
                if (npro > 1) issue(1, 1, 1);
                if (npro > 2) issue(2, 1, 1);
                if (npro > 3) issue(3, 2, 2);
                KURBM_PSTAMP(1);
                // barrier P: tile 0 has landed (the MFMA waves read it all, then meet again); barrier -1: tile 1
                if (npro == 4) __builtin_amdgcn_s_waitcnt(vm(3 * NPT));
                else if (npro == 3) __builtin_amdgcn_s_waitcnt(vm(2 * NPT));
                else if (npro == 2) __builtin_amdgcn_s_waitcnt(vm(NPT));
                else __builtin_amdgcn_s_waitcnt(VM0);
                KURBM_PSTAMP(2);
                __builtin_amdgcn_s_setprio(KURBM_PRIO_LOADER);   // (the first requests are out)
                __builtin_amdgcn_s_barrier();
                if (npro == 4) __builtin_amdgcn_s_waitcnt(vm(2 * NPT));
                else if (npro == 3) __builtin_amdgcn_s_waitcnt(vm(NPT));
                else __builtin_amdgcn_s_waitcnt(VM0);
                __builtin_amdgcn_s_barrier();
                int jb = 2, ab = 2;                                     // of tile t = 4
                int i = 0;
                for (; i + 4 < nt; ++i) {
                    KURBM_LSTAMP(0);
                    const int t = i + 4;
                    issue(t, jb, ab);
                    if (!(t & 1)) { ++jb; ab = (ab == 2) ? 0 : ab + 1; }
                    KURBM_LSTAMP(1);
                    __builtin_amdgcn_s_waitcnt(vm(2 * NPT));
                    KURBM_LSTAMP(2);
                    __builtin_amdgcn_s_barrier();
                    KURBM_LSTAMP(3);
                }
                for (; i < nt; ++i) {
                    if (i + 3 < nt) __builtin_amdgcn_s_waitcnt(vm(NPT));
                    else __builtin_amdgcn_s_waitcnt(VM0);
                    __builtin_amdgcn_s_barrier();
                }
                KURBM_LSTAMP_OUT();
            } else if constexpr (ABS) {
                // Unit u = tiles F(u) (fp8), Na(u), Nb(u) (three pieces each, the two halves of the unit's byte block).  The A
                // blocks rotate through THREE buffers (block 2 u = fp8, 2 u + 1 = bytes; buffer = block % 3); the B pieces of Na go
                // to stage 0, of Nb to stage 1, the fp8 tile's one piece to "stage" 2 (8 KB).  Every buffer is requested the
                // moment its last reader has passed its barrier, every request is 40 KB = 10 pieces per wave, and each lands at
                // least one whole tile before its first reader:
                //   while F(u) is multiplied:   first half of fp8 block u + 1,  B of Nb(u)
                //   while Na(u):                first half of byte block u + 1, B of F(u + 1), second half of fp8 block u + 1
                //   while Nb(u):                B of Na(u + 1),                 second half of byte block u + 1
                // so the tile's barrier waits for everything but the 10 pieces just requested (a counted vmcnt).
                static_assert(NA / 2 + 3 * NB1 == 10 && NA + NB1 == 10, "10 pieces per wave and tile");
                const int nu = nt / 3;
                TileRef rp = tile_of(t_begin), rn = tile_of(t_begin + 1);
                TileRef ra = rn;                                 // the byte block: its plane holds 128 k in the 128 bytes a
                ra.oa = rn.oa >> 1;                              // bf16 plane spends on 64 (piece 0: no plane offset in oa)
                int bf = 0, bb = 1;                              // A buffers of fp8 block u / byte block u (u = 0)
                auto nxt = [](int b, int by) constexpr { return (b + by) % 3; };
                dma_part(2, rp, 3, bf);                          // F(0): fp8 block 0 + its one B piece
                dma_part(0, ra, 1, bb);                          // byte block 0
                dma_part(0, rn, 2);                              // B of Na(0)
                __builtin_amdgcn_s_waitcnt(vm(NA + 3 * NB1));    // F(0) has landed
                __builtin_amdgcn_s_setprio(KURBM_PRIO_LOADER);   // (the first requests are out)
                __builtin_amdgcn_s_barrier();
                rn.ob += 128u;                                   // -> Nb(0)
                for (int u = 0; u < nu; ++u) {
                    const bool more = u + 1 < nu;
                    const int bf1 = nxt(bf, 2), bb1 = nxt(bb, 2);   // buffers of the next unit's blocks
                    // ---- F(u) in work
                    KURBM_LSTAMP(0);
                    if (more) { rp.oa += 128u; rp.ob += 128u; dma_part(0, rp, 1, bf1, 0, NA / 2); }
                    dma_part(1, rn, 2);                          // B of Nb(u)
                    rn.ob += 128u;
                    KURBM_LSTAMP(1);
                    if (more) __builtin_amdgcn_s_waitcnt(vm(10)); else __builtin_amdgcn_s_waitcnt(vm(3 * NB1));
                    KURBM_LSTAMP(2);
                    __builtin_amdgcn_s_barrier();
                    KURBM_LSTAMP(3);
                    // ---- Na(u) in work
                    if (more) {
                        ra.oa += 128u;
                        dma_part(0, ra, 1, bb1, 0, NA / 2);
                        dma_part(2, rp, 2);
                        dma_part(0, rp, 1, bf1, NA / 2, NA);
                        __builtin_amdgcn_s_waitcnt(vm(10));
                    } else {
                        __builtin_amdgcn_s_waitcnt(VM0);
                    }
                    __builtin_amdgcn_s_barrier();
                    // ---- Nb(u) in work
                    if (more) {
                        dma_part(0, rn, 2);                      // B of Na(u + 1)
                        rn.ob += 128u;
                        dma_part(0, ra, 1, bb1, NA / 2, NA);
                        __builtin_amdgcn_s_waitcnt(vm(10));
                    } else {
                        __builtin_amdgcn_s_waitcnt(VM0);
                    }
                    __builtin_amdgcn_s_barrier();
                    bf = bf1; bb = bb1;
                }
                KURBM_LSTAMP_OUT();
            } else if constexpr (AB) {
                // A block j (128 k of bytes) serves tiles 2 j and 2 j + 1; while tile i is multiplied, the B pieces of tile
                // i + 1 and HALF of block i / 2 + 1 are requested (its buffer was freed by tile 2 (i / 2) - 1)
                // (ONE segment -- launch_pb checks: B tile t lies 128 t bytes along k; no tile_of in the loop)
                TileRef ra = tile_of(t_begin);
                TileRef rb = ra;
                const uint32_t ob0 = ra.ob;
                ra.neg = false; ra.oa = 128u * (uint32_t)(t_begin >> 1);   // (a k slice starts on a whole block: launch_pb checks)
                KURBM_PSTAMP(0);
                dma_part(0, ra, 3, 0);
                KURBM_PSTAMP(1);
                __builtin_amdgcn_s_waitcnt(VM0);
                KURBM_PSTAMP(2);
                __builtin_amdgcn_s_setprio(KURBM_PRIO_LOADER);   // (the first requests are out)
                __builtin_amdgcn_s_barrier();
                for (int i = 0; i < nt; ++i) {
                    KURBM_LSTAMP(0);
                    if (i + 1 < nt) { rb.ob = ob0 + 128u * (uint32_t)(i + 1); dma_part((i + 1) & 1, rb, 2); }
                    const int jb = (i >> 1) + 1;
                    if (2 * jb < nt) {
                        ra.oa = __builtin_amdgcn_readfirstlane(128u * (uint32_t)((t_begin >> 1) + jb));
                        if (i & 1) dma_part(0, ra, 1, jb & 1, NA / 2, NA);
                        else dma_part(0, ra, 1, jb & 1, 0, NA / 2);
                    }
                    KURBM_LSTAMP(1);
                    __builtin_amdgcn_s_waitcnt(VM0);
                    KURBM_LSTAMP(2);
                    __builtin_amdgcn_s_barrier();
                    KURBM_LSTAMP(3);
                }
                KURBM_LSTAMP_OUT();
            } else if constexpr (NSTG == 3) {
                // ring of three stages: while tile i is multiplied, tile i + 1 is landing and tile i + 2 is requested into the
                // stage tile i - 1 has left; a counted vmcnt leaves the youngest tile's pieces in flight across the barrier
                constexpr int NP = NA + NB1;   // (one-piece tiles: pieces per loader wave and tile)
                Walk w = walk_at(t_begin);
                dma_tile(0, walk_ref(w)); walk_next(w);
                if (nt > 1) { dma_tile(1, walk_ref(w)); walk_next(w); }
                if (nt > 1) __builtin_amdgcn_s_waitcnt(vm(NP)); else __builtin_amdgcn_s_waitcnt(VM0);
                __builtin_amdgcn_s_setprio(KURBM_PRIO_LOADER);   // (the first requests are out)
                __builtin_amdgcn_s_barrier();
                int nb = 2;   // stage of tile i + 2
                for (int i = 0; i < nt; ++i) {
                    if (i + 2 < nt) {
                        dma_tile(nb, walk_ref(w)); walk_next(w);
                        __builtin_amdgcn_s_waitcnt(vm(NP));
                    } else {
                        __builtin_amdgcn_s_waitcnt(VM0);
                    }
                    nb = nb == 2 ? 0 : nb + 1;
                    __builtin_amdgcn_s_barrier();
                }
            } else if (bsp) {
                // The paired walk: k position kt = tile T0(kt) [A piece 0 x B pieces 0-2] and tile T1(kt) [A1 x B0-1, A2 x B0].  A tiles
                // in the order they are consumed, a = 3 kt + piece, live in A stage a & 3; the three B pieces of a position in B
                // stage kt & 1.  A stage may be refilled once the tile that read it is behind its barrier, so while T0(kt) is
                // multiplied the stages of A1 / A2(kt - 1) and the B stage of kt - 1 are free, while T1(kt) that of A0(kt):
                //   slot T0(kt):  A2(kt) -- needed behind this tile's barrier, so it goes FIRST --, B0(kt + 1), B1(kt + 1)
                //   slot T1(kt):  B2(kt + 1), A0(kt + 1), A1(kt + 1)
                // 12 pieces per wave and slot; the barrier of T0(kt) waits for everything but the 8 B pieces just requested, that
                // of T1(kt) for everything but A1(kt + 1): every request but A2's has a whole tile to land.
                static_assert(!BSP || (NA == 4 && NB1 == 4), "paired walk: 128 x 128 tiles");
                const int np = nt / 3;                           // (launch_gemm_pb: whole positions per slice)
                Walk w = walk_at(t_begin);
                TileRef ra0 = walk_ref(w); walk_next(w);
                TileRef ra1 = walk_ref(w); walk_next(w);
                TileRef ra2 = walk_ref(w);
                TileRef rb = ra0;                                // (B of the position: all three pieces)
                rb.npb = 3;
                // (a_tr: a k position is 64 ROWS of the row-major A plane, not 128 bytes along a row)
                const uint32_t astep = a_tr ? 128u * (uint32_t)g.lda : 128u;
                if (a_tr) { const uint32_t fix = (astep - 128u) * (uint32_t)w.b; ra0.oa += fix; ra1.oa += fix; ra2.oa += fix; }
                dma_part(0, ra0, 1); dma_part(0, rb, 2);         // A0(0) -> A stage 0, B(0) -> B stage 0
                dma_part(1, ra1, 1);                             // A1(0) -> A stage 1
                __builtin_amdgcn_s_waitcnt(vm(NA));              // A0(0), B(0) have landed
                __builtin_amdgcn_s_setprio(KURBM_PRIO_LOADER);   // (the first requests are out)
                __builtin_amdgcn_s_barrier();
                int sa = 2;                                      // A stage of tile a = 3 kt + 2
                for (int kt = 0; kt < np; ++kt) {
                    const bool more = kt + 1 < np;
                    KURBM_LSTAMP(0);
                    // ---- T0(kt) in work
                    dma_part(sa, ra2, 1);                        // A2(kt)
                    ra0.oa += astep; ra1.oa += astep; ra2.oa += astep; rb.ob += 128u;   // -> position kt + 1
                    if (more) {
                        dma_part((kt + 1) & 1, rb, 2, 0, 0, 0, 0, 2);               // B0, B1(kt + 1)
                        KURBM_LSTAMP(1);
                        __builtin_amdgcn_s_waitcnt(vm(2 * NB1));
                    } else {
                        KURBM_LSTAMP(1);
                        __builtin_amdgcn_s_waitcnt(VM0);
                    }
                    KURBM_LSTAMP(2);
                    __builtin_amdgcn_s_barrier();
                    KURBM_LSTAMP(3);
                    // ---- T1(kt) in work
                    if (more) {
                        dma_part((kt + 1) & 1, rb, 2, 0, 0, 0, 2, 3);               // B2(kt + 1)
                        dma_part((sa + 1) & 3, ra0, 1);                              // A0(kt + 1): a = 3 kt + 3
                        dma_part((sa + 2) & 3, ra1, 1);                              // A1(kt + 1): a = 3 kt + 4
                        __builtin_amdgcn_s_waitcnt(vm(NA));
                    } else {
                        __builtin_amdgcn_s_waitcnt(VM0);
                    }
                    __builtin_amdgcn_s_barrier();
                    sa = (sa + 3) & 3;
                }
                KURBM_LSTAMP_OUT();
            } else {
            // both stages are free at the start: tiles 0 and 1 are requested back to back, and the first barrier waits for
            // tile 0's pieces only (a counted vmcnt leaves tile 1's in flight)
            Walk w = walk_at(t_begin);
            dma_tile(0, walk_ref(w)); walk_next(w);
            if (nt > 1) {
                const TileRef t1 = walk_ref(w);
                walk_next(w);                                   // (w: tile 2)
                dma_tile(1, t1);
                if (t1.npb >= 3) __builtin_amdgcn_s_waitcnt(vm(NA + 3 * NB1));
                else if (t1.npb == 2) __builtin_amdgcn_s_waitcnt(vm(NA + 2 * NB1));
                else __builtin_amdgcn_s_waitcnt(vm(NA + NB1));
            } else {
                __builtin_amdgcn_s_waitcnt(VM0);
            }
            __builtin_amdgcn_s_setprio(KURBM_PRIO_LOADER);   // (the first requests are out)
            __builtin_amdgcn_s_barrier();
            if (F8 && g.walk3) {
                // fp8, 3-piece, 3-piece, ...: the fp8 tile of unit u lies 128 u bytes along k in its planes, the two other tiles
                // 128 (2 u), 128 (2 u + 1) in theirs -- two running tile references, no tile list
                TileRef rp = tile_of(t_begin), rn = tile_of(t_begin + 1);
                int kind = 2;                        // of tile i + 1 = 2: the second 3-piece tile of unit 0
                rn.oa += 128u; rn.ob += 128u;
                rp.oa += 128u; rp.ob += 128u;        // (the fp8 tile of unit 1)
                for (int i = 0; i < nt; ++i) {
                    KURBM_LSTAMP(0);
                    if (i >= 1 && i + 1 < nt) {
                        if (kind == 0) { dma_tile((i + 1) & 1, rp); rp.oa += 128u; rp.ob += 128u; }
                        else { dma_tile((i + 1) & 1, rn); rn.oa += 128u; rn.ob += 128u; }
                    }
                    if (i >= 1) kind = (kind == 2) ? 0 : kind + 1;
                    KURBM_LSTAMP(1);
                    __builtin_amdgcn_s_waitcnt(VM0);
                    KURBM_LSTAMP(2);
                    __builtin_amdgcn_s_barrier();
                    KURBM_LSTAMP(3);
                }
            } else {
            for (int i = 0; i < nt; ++i) {
                KURBM_LSTAMP(0);
                if (i >= 1 && i + 1 < nt) { dma_tile((i + 1) & 1, walk_ref(w)); walk_next(w); }
                KURBM_LSTAMP(1);
                __builtin_amdgcn_s_waitcnt(VM0);
                KURBM_LSTAMP(2);
                __builtin_amdgcn_s_barrier();
                KURBM_LSTAMP(3);
            }
            }
            KURBM_LSTAMP_OUT();
            }
        }
        __syncthreads();
        // keep the MFMA waves' epilogue barriers company: one per patch write, one between the pieces of a plane
        if (EPI == EPI_SLAB || EPI == EPI_SOFTPLUS) __syncthreads();
        else {
            if (RP) { __syncthreads(); __syncthreads(); }   // (the row partials' trip through LDS)
            if (g.out) {
                const int nb = (g.out_pieces == 3) ? 5 : 1;
                for (int q = 0; q < nb; ++q) __syncthreads();
            }
        }
        return;
    }
    if constexpr (NI_LDS > 0) {
        if (nt > 0) {
#pragma unroll
            for (int ni = 0; ni < NI_LDS; ++ni)
#pragma unroll
                for (int mi = 0; mi < TM; ++mi) {
                    const uint64_t grow = g.rng.row0 + (uint64_t)(m0 + wm * WM + slot * 4 + mi * 16);
                    uint32_t w[4];
                    philox4x32_10((uint32_t)(n0 + wn * WN + l15 + ni * 16), (uint32_t)(grow >> 2), g.rng.stream_id, g.rng.step,
                                  g.rng.seed_lo, g.rng.seed_hi, w);
                    *reinterpret_cast<u32x4*>(smem + SMEM_BYTES + ((ni * TM + mi) * NT + tid) * 16) = u32x4{w[0], w[1], w[2], w[3]};
                }
        }
    }
    __builtin_amdgcn_s_setprio(KURBM_PRIO_MFMA);   // the MFMA waves go first wherever a loader wave competes for issue (the reverse: no difference)
    if (nt > 0) {
        __syncthreads();
        frag_a(0, 0, fa[0]);
        frag_b(ABS ? 2 : 0, 0, 0, fb[0]);  // (ABS: the fp8 tile's B piece has a stage of its own)
        if constexpr (!ABS) expand_a(0);   // (ABS enters an fp8 tile: raw fragments)
        KURBM_STAMP(ts[1]);
        // unrolled by two: the LDS buffers alternate statically
        int i = 0;
        if constexpr (ABS) {
            // fp8 / 3-piece / 3-piece in whole units (g.walk3: launch_pb checks); the unit's fp8 block and its byte block (whose two
            // halves are the two 3-piece tiles) sit in two of three A buffers (the loaders' comment has the ring).  The expanded
            // fragments of a byte tile live in fa[1], which only the fp8 tile uses as raw fragments.
            const uint32_t arow = (uint32_t)((wm * WM + l15) * ROWB);
            uint32_t of8 = 0u, oby = A_BYTES;     // A buffers of the unit's fp8 block / byte block (blocks 2 u, 2 u + 1; buffer = block % 3)
            auto rd_f8 = [&](uint32_t off, int ks, afrag (&f)[TM]) __attribute__((always_inline)) {
                const unsigned char* c = smem + (arow + 16 * ((4 * ks + slot) ^ swz) + off);
#pragma unroll
                for (int mi = 0; mi < TM; ++mi) f[mi] = *reinterpret_cast<const u32x4*>(c + mi * 16 * ROWB);
            };
            auto rd_by = [&](uint32_t off, int half, afrag (&f)[TM]) __attribute__((always_inline)) {
                const unsigned char* c = smem + (arow + 16 * ((4 * half + slot) ^ swz) + off);
#pragma unroll
                for (int mi = 0; mi < TM; ++mi) f[mi] = *reinterpret_cast<const u32x4*>(c + mi * 16 * ROWB);
            };
            auto adv = [](uint32_t o) __attribute__((always_inline)) {   // two blocks on in the ring of three
                return o + 2 * A_BYTES >= 3 * A_BYTES ? o + 2 * A_BYTES - 3 * A_BYTES : o + 2 * A_BYTES;
            };
            auto expand0 = [&](int ks) __attribute__((always_inline)) {   // fa[0] (16 bytes = both k-steps) -> fa[1] (one k-step as bf16)
#pragma unroll
                for (int mi = 0; mi < TM; ++mi) {
                    const uint32_t lo = ks ? fa[0][mi].z : fa[0][mi].x, hi = ks ? fa[0][mi].w : fa[0][mi].y;
                    fa[1][mi] = u32x4{__builtin_amdgcn_perm(0u, lo, 0x010C000Cu), __builtin_amdgcn_perm(0u, lo, 0x030C020Cu),
                                      __builtin_amdgcn_perm(0u, hi, 0x010C000Cu), __builtin_amdgcn_perm(0u, hi, 0x030C020Cu)};
                }
            };
            auto mf = [&](const u32x4 (&b)[TN]) __attribute__((always_inline)) {
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8, fa[1][mi]), __builtin_bit_cast(bf16x8, b[ni]), acc[mi][ni], 0, 0, 0);
            };
            // B stages: 0 = Na, 1 = Nb, 2 = the fp8 tile's piece.  The fp8 tile is entered with fa[0] = chunks 0-3 of its block,
            // fb[0] = those of its B piece
            auto f8t = [&]() __attribute__((always_inline)) {
                typedef int i32x8 __attribute__((ext_vector_type(8)));
                __builtin_amdgcn_sched_barrier(0);
                rd_f8(of8, 1, fa[1]);
                frag_b(2, 1, 0, fb[1]);
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni) {
                        const i32x8 a = {(int)fa[0][mi].x, (int)fa[0][mi].y, (int)fa[0][mi].z, (int)fa[0][mi].w,
                                         (int)fa[1][mi].x, (int)fa[1][mi].y, (int)fa[1][mi].z, (int)fa[1][mi].w};
                        const i32x8 b = {(int)fb[0][ni].x, (int)fb[0][ni].y, (int)fb[0][ni].z, (int)fb[0][ni].w,
                                         (int)fb[1][ni].x, (int)fb[1][ni].y, (int)fb[1][ni].z, (int)fb[1][ni].w};
                        // (A scaled by 2^1: the slab epilogue halves every sum)
                        acc[mi][ni] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[mi][ni], 0, 0, 0, 0x80, 0, 0x7F);
                    }
                __builtin_amdgcn_sched_barrier(0);
                __syncthreads();
                __builtin_amdgcn_sched_barrier(0);
                rd_by(oby, 0, fa[0]);
                frag_b(0, 0, 0, fb[0]);
                __builtin_amdgcn_sched_barrier(0);
                expand0(0);
                __builtin_amdgcn_sched_barrier(0);
            };
            // a 3-piece tile of half `half` of the byte block, B pieces in stage `half`: entered with fa[0] raw, fa[1] = k-step 0
            // expanded, fb[0] = piece 0 of k-step 0; micro-step u = 3 ks + piece, fragments read two micro-steps ahead.  `nf8`:
            // the A buffer of the NEXT unit's fp8 block (behind the second tile)
            auto negt = [&](const int half, const uint32_t nf8) __attribute__((always_inline)) {
#pragma unroll
                for (int u = 0; u < 6; ++u) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (u == 0) { frag_b(half, 0, 1, fb[1]); frag_b(half, 0, 2, fb[2]); }
                    if (u == 1) frag_b(half, 1, 0, fb[0]);
                    if (u == 2) frag_b(half, 1, 1, fb[1]);
                    if (u == 3) frag_b(half, 1, 2, fb[2]);
                    if (u == 5) {
                        __syncthreads();
                        __builtin_amdgcn_sched_barrier(0);
                        if (half == 0) { rd_by(oby, 1, fa[0]); frag_b(1, 0, 0, fb[0]); }
                        else { rd_f8(nf8, 0, fa[0]); frag_b(2, 0, 0, fb[0]); }
                    }
                    mf(fb[u % 3]);
                    if (u == 2) { __builtin_amdgcn_sched_barrier(0); expand0(1); }
                    if (u == 5 && half == 0) { __builtin_amdgcn_sched_barrier(0); expand0(0); }
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            for (; i + 2 < nt; i += 3) {
                const uint32_t nf8 = adv(of8);
                f8t();
                negt(0, nf8);
                negt(1, nf8);
                of8 = nf8; oby = adv(oby);
            }
        } else if constexpr (DEEP) {
            // (the pre-loop reads above took tile 0's k-step 0; behind barrier P the rest of tile 0, then barrier -1)
            const uint32_t la = (uint32_t)((wm * WM + l15) * ROWB + 16 * (slot ^ swz));            // + block; ^ 64: second tile of the block
            const uint32_t lb = (uint32_t)(B_OFF + (wn * WN + l15) * ROWB + 16 * (slot ^ swz));    // + stage; ^ 64: k-step 1
            auto rd_a = [&](uint32_t blk_off, int half, afrag (&f)[TM]) __attribute__((always_inline)) {
                const unsigned char* c = smem + ((la ^ (64u * (uint32_t)half)) + blk_off);
#pragma unroll
                for (int mi = 0; mi < TM; ++mi) f[mi] = *reinterpret_cast<const u32x4*>(c + mi * 16 * ROWB);
            };
            auto rd_b = [&](uint32_t stg_off, int ks, u32x4 (&f)[TN]) __attribute__((always_inline)) {
                const unsigned char* c = smem + ((lb ^ (64u * (uint32_t)ks)) + stg_off);
#pragma unroll
                for (int ni = 0; ni < TN; ++ni) f[ni] = *reinterpret_cast<const u32x4*>(c + ni * 16 * ROWB);
            };
            auto expand_from = [&](const afrag (&f)[TM], int ks) __attribute__((always_inline)) {
#pragma unroll
                for (int mi = 0; mi < TM; ++mi) {
                    const uint32_t lo = ks ? f[mi].z : f[mi].x, hi = ks ? f[mi].w : f[mi].y;
                    fx[mi] = u32x4{__builtin_amdgcn_perm(0u, lo, 0x010C000Cu), __builtin_amdgcn_perm(0u, lo, 0x030C020Cu),
                                   __builtin_amdgcn_perm(0u, hi, 0x010C000Cu), __builtin_amdgcn_perm(0u, hi, 0x030C020Cu)};
                }
            };
            auto deep_barrier = [&]() __attribute__((always_inline)) {
                __builtin_amdgcn_sched_barrier(0);
#if KURBM_DEEP_FENCE
                __syncthreads();
#else
                __builtin_amdgcn_s_barrier();
#endif
                __builtin_amdgcn_sched_barrier(0);
            };
            u32x4 fb4[4][TN];                      // B fragments of micro-steps m, m + 1, m + 2 (m = 2 i + k-step), index m % 4
            uint32_t sbn = B_BYTES;                // stage of tile i + 1
            uint32_t oab = 0u, oan = A_BYTES;      // A block of the tile pair in work, of the next pair
            // Tile i of parity P, entered with fx = A(i, k-step 0) expanded, fa[P] = A(i) raw, fb4[2 P], fb4[2 P + 1] = B(i):
            // every read is for tile i + 1, two micro-steps ahead of the MFMAs that use it -- no read waits behind the barrier.
            // (The permutes stay BEHIND a k-step's MFMAs: two MFMAs / four permutes alternating measured 3 % slower.)
            auto deep_tile = [&](auto par_tag) __attribute__((always_inline)) {
                constexpr int P = decltype(par_tag)::value;
#ifdef KURBM_STAMPS
                unsigned long long tq0, tq1, tq2;   // tk[0]: the tiles up to their barrier, tk[1]: at the barrier (tu[] belongs to the epilogue)
                KURBM_STAMP(tq0);
#endif
                __builtin_amdgcn_sched_barrier(0);
                if (P == 0) rd_a(oab, 1, fa[1]); else rd_a(oan, 0, fa[0]);
                rd_b(sbn, 0, fb4[(2 * P + 2) & 3]);
                mfmas(fa[0], fb4[2 * P]);
                __builtin_amdgcn_sched_barrier(0);
                expand_from(fa[P], 1);
                __builtin_amdgcn_sched_barrier(0);
                rd_b(sbn, 1, fb4[(2 * P + 3) & 3]);
                mfmas(fa[0], fb4[2 * P + 1]);
                __builtin_amdgcn_sched_barrier(0);
                expand_from(fa[P ^ 1], 0);
#ifdef KURBM_STAMPS
                KURBM_STAMP(tq1);
#endif
                deep_barrier();
#ifdef KURBM_STAMPS
                KURBM_STAMP(tq2);
                tk[0] += tq1 - tq0; tk[1] += tq2 - tq1;
#endif
                sbn = (sbn + B_BYTES == NSTG * B_BYTES) ? 0u : sbn + B_BYTES;
                if (P == 1) { oab = oan; oan = (oan + A_BYTES == NAB * A_BYTES) ? 0u : oan + A_BYTES; }
            };
            // tile 0: fa[0] was read above (k-step 0 = half 0 of block 0: the same 16 bytes hold both k-steps), fb[0] = B(0, 0)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) fb4[0][ni] = fb[0][ni];
            rd_b(0u, 1, fb4[1]);
            deep_barrier();
            for (; i + 1 < nt; i += 2) {
                deep_tile(std::integral_constant<int, 0>{});
                deep_tile(std::integral_constant<int, 1>{});
            }
            if (i < nt) deep_tile(std::integral_constant<int, 0>{});
        } else if constexpr (NSTG == 3) {
            Walk w = walk_at(t_begin);
            for (; i + 2 < nt; i += 3) {
                tile_any(0, 0, walk_ref(w)); walk_next(w);
                tile_any(1, 0, walk_ref(w)); walk_next(w);
                tile_any(2, 0, walk_ref(w)); walk_next(w);
            }
            if (i < nt) { tile_any(0, 0, walk_ref(w)); walk_next(w); }
            if (i + 1 < nt) tile_any(1, 0, walk_ref(w));
        } else {
            if constexpr (AB) {
                // one segment of PB pieces (launch_pb checks): every tile is the same tile -- no tile list, no dispatch
                for (; i + 1 < nt; i += 2) {
                    one_tile(0, 0, std::integral_constant<int, PB>{});
                    one_tile(1, 1, std::integral_constant<int, PB>{});
                    ablk ^= 1;
                }
                if (i < nt) one_tile(0, 0, std::integral_constant<int, PB>{});
            } else {
                if (bsp) {
                    // (see the loaders) k position kt: one three-piece tile on A stage (3 kt) & 3, then the paired tile on the next two
                    const int np = nt / 3;
                    int s0 = 0;
                    for (int kt = 0; kt < np; ++kt) {
                        const int s1 = (s0 + 1) & 3, s2 = (s0 + 2) & 3, s3 = (s0 + 3) & 3, b = kt & 1;
#ifdef KURBM_STAMPS
                        unsigned long long tp0, tp1, tp2;
                        KURBM_STAMP(tp0);
#endif
                        one_tile(s0, 0, std::integral_constant<int, 3>{}, b, s1, b);
#ifdef KURBM_STAMPS
                        KURBM_STAMP(tp1);
#endif
                        pair_tile(s1, s2, b, s3, b ^ 1);
#ifdef KURBM_STAMPS
                        KURBM_STAMP(tp2);
                        tk[2] += tp1 - tp0; tk[1] += tp2 - tp1;     // (three-piece tile / paired tile, summed over the positions)
#endif
                        s0 = s3;
                    }
                    i = nt;
                } else if (F8 && g.walk3) {
                    // whole units of fp8, 3-piece, 3-piece; the stages alternate: six tiles per trip
                    for (; i + 5 < nt; i += 6) {
                        f8_tile(0);
                        one_tile(1, 1, std::integral_constant<int, 3>{});
                        one_tile(0, 0, std::integral_constant<int, 3>{});
                        f8_tile(1);
                        one_tile(0, 0, std::integral_constant<int, 3>{});
                        one_tile(1, 1, std::integral_constant<int, 3>{});
                    }
                    if (i < nt) {
                        f8_tile(0);
                        one_tile(1, 1, std::integral_constant<int, 3>{});
                        one_tile(0, 0, std::integral_constant<int, 3>{});
                    }
                } else {
                Walk w = walk_at(t_begin);
                for (; i + 1 < nt; i += 2) {
                    tile_any(0, 0, walk_ref(w)); walk_next(w);
                    tile_any(1, 1, walk_ref(w)); walk_next(w);
                    ablk ^= 1;
                }
                if (i < nt) tile_any(0, 0, walk_ref(w));
                }
            }
        }
    }
    __syncthreads();
    KURBM_STAMP(ts[2]);

    // ---------------- epilogue: raw partial sums to a slab (statistics GEMM), through an fp32 patch ----
    // C layout: acc[mi][ni][r] is row m0 + wm WM + 16 mi + 4 slot + r, column n0 + wn WN + 16 ni + l15.
    if (EPI == EPI_SLAB) {
        float* slab = g.slab + (size_t)z * g.slab_stride;
        if constexpr (ABP) {
            if (g.slab_t) {
                // The transposed problem: this tile is rows m0.. of B^T A... i.e. element (r, c) of the tile belongs at slab row
                // n0 + c, column m0 + r.  A lane holds four consecutive r of its column c: one 16-byte write into a [BN][BM] patch
                // (8 consecutive lanes = 8 rows of the patch, 260 dwords apart: all 32 banks), then whole patch rows leave as
                // 1-KB runs along the slab's rows.  Tile rows past M (A rows the loaders pointed at row 0) are zeros.
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni) {
                        const int r = wm * WM + mi * 16 + slot * 4, c = wn * WN + ni * 16 + l15;
                        f32x4 v = acc[mi][ni] * 0.5f;                           // (the A bytes read as 2.0)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (m0 + r + e >= g.M) v[e] = 0.f;
                        *reinterpret_cast<f32x4*>(smem + c * PROWT + 4 * r) = v;
                    }
                __syncthreads();
                constexpr int CHT = BM / 4;   // 16-B chunks per patch row
#pragma unroll
                for (int j = 0; j < BN * CHT / NT; ++j) {
                    const int q = j * NT + tid, row = q / CHT, c4 = q % CHT;
                    const int gr = n0 + row, gc = m0 + 4 * c4;
                    if (gr < g.N && gc < g.ld_slab)
                        *reinterpret_cast<f32x4*>(slab + (size_t)gr * g.ld_slab + gc) = *reinterpret_cast<const f32x4*>(smem + row * PROWT + 16 * c4);
                }
                KURBM_STAMP_OUT();
                return;
            }
        }
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    *reinterpret_cast<float*>(smem + (wm * WM + mi * 16 + slot * 4 + r) * PROW32 +
                                              4 * (wn * WN + ni * 16 + l15)) =
                        (n0 + wn * WN + ni * 16 + l15 < g.N) ? ((ABS || ABP) ? 0.5f * acc[mi][ni][r] : acc[mi][ni][r]) : 0.f;
        __syncthreads();
        constexpr int CH = BN / 4;   // 16-B chunks per row
#pragma unroll
        for (int j = 0; j < BM * CH / NT; ++j) {
            const int q = j * NT + tid, row = q / CH, c = q % CH;
            const int gr = m0 + row, gc = n0 + 4 * c;
            if (gr < g.M && gc < g.ld_slab) {   // ld_slab = N rounded up to 4: the chunk stays inside the row
                const f32x4 v = *reinterpret_cast<const f32x4*>(smem + row * PROW32 + 16 * c);
                *reinterpret_cast<f32x4*>(slab + (size_t)gr * g.ld_slab + gc) = v;
            }
        }
        KURBM_STAMP_OUT();
        return;
    }

    // ---------------- epilogue: row sums of softplus(x + b) over the tile's columns (free energy, rbm.py:73-75) ----
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
                    if (cok) rsum[mi][r] += softplusf((AB ? 0.5f * acc[mi][ni][r] : acc[mi][ni][r]) + bias);   // (AB: the A bytes read as 2.0)
        }
        float* red = reinterpret_cast<float*>(smem);   // [WAVES_N][BM]
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float t = row16_sum(rsum[mi][r]);
                if (l15 == 0) red[wn * BM + wm * WM + mi * 16 + slot * 4 + r] = t;
            }
        __syncthreads();
        if (tid < BM) {
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < WAVES_N; ++i) t += red[i * BM + tid];
            if (m0 + tid < g.M) g.rowpart[(size_t)bn * g.ld_rowpart + m0 + tid] = t;
        }
        KURBM_STAMP_OUT();
        return;
    }

    // ---------------- epilogue of a half step ------------------------------------------------
    if constexpr (RP) {
        // row sums of softplus(pre-activation) over the tile's columns, as the free-energy GEMM's epilogue forms them
        float rsum[TM][4];
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) rsum[mi][r] = 0.f;
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            const bool cok = n0 + wn * WN + ni * 16 + l15 < g.N;
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (cok) rsum[mi][r] += softplusf((AB ? 0.5f * acc[mi][ni][r] : acc[mi][ni][r]) + biasv[ni]);
        }
        float* red = reinterpret_cast<float*>(smem);   // [WAVES_N][BM]: the stages are free behind the k loop's last barrier
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float t = row16_sum(rsum[mi][r]);
                if (l15 == 0) red[wn * BM + wm * WM + mi * 16 + slot * 4 + r] = t;
            }
        __syncthreads();
        if (tid < BM && g.rowpart) {
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < WAVES_N; ++i) t += red[i * BM + tid];
            if (m0 + tid < g.M) g.rowpart[(size_t)bn * g.ld_rowpart + m0 + tid] = t;
        }
        __syncthreads();                               // (the plane patches reuse this LDS)
    }
    float xv[TM][TN][4];   // the value plane: the sample, or the probability when nothing is drawn
    int colb = n0 + wn * WN + l15;
    asm volatile("" : "+v"(colb));   // opaque: keeps the epilogue's address arithmetic out of the k loop's registers
    int rowq = m0 + wm * WM + slot * 4;
    asm volatile("" : "+v"(rowq));
    {
        if constexpr (NI_LDS > 0) {
            if (nt > 0) {
#pragma unroll
                for (int ni = 0; ni < NI_LDS; ++ni)
#pragma unroll
                    for (int mi = 0; mi < TM; ++mi) {
                        const u32x4 w = *reinterpret_cast<const u32x4*>(smem + SMEM_BYTES + ((ni * TM + mi) * NT + tid) * 16);
                        draws[ni * TM + mi][0] = w.x; draws[ni * TM + mi][1] = w.y;
                        draws[ni * TM + mi][2] = w.z; draws[ni * TM + mi][3] = w.w;
                    }
            }
            draw_cols(nt > 0 ? NI_LDS : 0, TN);
        } else {
            draw_cols(0, TN);
        }
        // (SIDE = the fp32 test planes are wanted: a compile-time tag, because a per-element test of g.side, however
        //  uniform, puts a branch between every two of the 32 sigmoids of a lane and serialises their latencies)
        auto elementwise = [&](auto act_tag, auto side_tag) __attribute__((always_inline)) {
            constexpr bool SIDE = decltype(side_tag)::value;
            constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                const int col = colb + ni * 16;
                const bool col_ok = col < g.N;
                const float bias = biasv[ni];
#pragma unroll
                for (int mi = 0; mi < TM; ++mi) {
                    const int rowb = rowq + mi * 16;
                    const uint32_t (&w)[4] = draws[(NOISE != NOISE_NONE) ? ni * TM + mi : 0];
                    uint32_t w2[4] = {0u, 0u, 0u, 0u};
                    if (NOISE == NOISE_GAUSSIAN) {
                        const uint64_t grow = g.rng.row0 + (uint64_t)rowb;
                        philox4x32_10((uint32_t)col, (uint32_t)(grow >> 2), g.rng.stream_id | 0x80000000u, g.rng.step,
                                      g.rng.seed_lo, g.rng.seed_hi, w2);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float x = (AB ? 0.5f * acc[mi][ni][r] : acc[mi][ni][r]) + bias;   // (AB: the A bytes read as 2.0)
                        float p;
                        if (ACT == ACT_SIGMOID) p = sigmoidf_fast(x);
                        else if (ACT == ACT_RELU) p = fmaxf(x, 0.f);
                        else p = x;
                        const float ua = (NOISE != NOISE_NONE) ? u32_to_unit(w[r]) : 0.f;
                        if (NOISE == NOISE_GAUSSIAN)   // N(p, 1) by Box-Muller from the site's two planes (oracle/philox.py normal())
                            xv[mi][ni][r] = p + box_muller(ua, u32_to_unit(w2[r]));
                        else
                            xv[mi][ni][r] = (NOISE == NOISE_BERNOULLI) ? ((ua < p) ? 1.0f : 0.0f) : p;
                        // (Gaussian: the run-time test stays -- 32 Box-Muller chains in flight at once spill)
                        if (NOISE != NOISE_NONE && (NOISE == NOISE_GAUSSIAN ? g.side : SIDE) && col_ok && rowb + r < g.M) {   // test planes
                            if (g.prob_f32) g.prob_f32[(size_t)(rowb + r) * g.ldo32 + col] = p;
                            if (g.out_u) g.out_u[(size_t)(rowb + r) * g.ldo32 + col] = ua;
                        }
                    }
                }
            }
        };
        const bool side = (NOISE == NOISE_BERNOULLI) && g.side;
        if (!side) {
            if (g.act == ACT_SIGMOID) elementwise(std::integral_constant<int, ACT_SIGMOID>{}, std::false_type{});
            else if (g.act == ACT_RELU) elementwise(std::integral_constant<int, ACT_RELU>{}, std::false_type{});
            else elementwise(std::integral_constant<int, ACT_LINEAR>{}, std::false_type{});
        } else {
            if (g.act == ACT_SIGMOID) elementwise(std::integral_constant<int, ACT_SIGMOID>{}, std::true_type{});
            else if (g.act == ACT_RELU) elementwise(std::integral_constant<int, ACT_RELU>{}, std::true_type{});
            else elementwise(std::integral_constant<int, ACT_LINEAR>{}, std::true_type{});
        }
    }
    KURBM_STAMP(ts[3]);

    // (a) column sums of the value plane over this wave's WM rows of the tile: row (bm * WAVES_M + wm) of colpart
    if (g.colpart) {
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            float cs = 0.f;
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (rowq + mi * 16 + r < g.M) cs += xv[mi][ni][r];
            cs += __shfl_xor(cs, 16);
            cs += __shfl_xor(cs, 32);
            const int col = colb + ni * 16;
            // (the host lays colpart out as one row per 64 rows of the tile: a wave that covers more writes its sum to the
            //  first of its rows and zeros to the others)
            constexpr int RPW = WM / 64;
            static_assert(WM % 64 == 0, "column partials: one row per 64 tile rows");
            if (slot == 0 && col < g.N) {
                float* cp = g.colpart + (size_t)((bm * WAVES_M + wm) * RPW) * g.ld_colpart + col;
                cp[0] = g.colsign * cs;
#pragma unroll
                for (int q = 1; q < RPW; ++q) cp[(size_t)q * g.ld_colpart] = 0.f;
            }
        }
    }

#ifdef KURBM_STAMPS
    unsigned long long te[5] = {0, 0, 0, 0, 0};
    KURBM_STAMP(te[0]);
#endif
    // (b) transposed bf16 plane(s) [N][ldoT]: 4 consecutive rows of this lane's column = one 8-byte store;
    //     rows past M (k padding of the statistics GEMM) are written as zeros
    static_assert(TM == 4, "the transposed planes' 4 x 4 lane transpose: four 16-row blocks per wave");
    if (g.outT && g.outT_f8) {
        // A 0/1 sample as ONE byte per element (fp8 1.0 = 0x38, or 0x40 of a k-permuted byte plane).  A lane holds four bytes
        // (four rows) per 16-row block mi; a 4 x 4 transpose between the block index and the lane's row group (two
        // v_permlane32_swap + two v_permlane16_swap) gives it 16 CONSECUTIVE rows of its column instead: per column of the
        // wave's tile one 64-byte run per store instruction where there were four 16-byte pieces in four instructions -- a
        // quarter of the partial cache lines this epilogue touches (its stores were issue-bound on exactly that: 3 600 cycles).
        const uint32_t one = (g.outT_f8 == 2) ? 0x40u : 0x38u;
        const int rq16 = rowq + slot * 12;             // m0 + wm WM + 16 slot
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            const int col = colb + ni * 16;
            uint32_t pk[4];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const int rb = rowq + mi * 16;
                pk[mi] = 0u;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (rb + r < g.M && xv[mi][ni][r] != 0.f) pk[mi] |= one << (8 * r);
            }
            {   // (every lane takes part: the swaps sit outside the bounds checks)
                auto s0 = __builtin_amdgcn_permlane32_swap(pk[0], pk[2], false, false);
                auto s1 = __builtin_amdgcn_permlane32_swap(pk[1], pk[3], false, false);
                auto s2 = __builtin_amdgcn_permlane16_swap(s0[0], s1[0], false, false);
                auto s3 = __builtin_amdgcn_permlane16_swap(s0[1], s1[1], false, false);
                pk[0] = s2[0]; pk[1] = s2[1]; pk[2] = s3[0]; pk[3] = s3[1];
            }
            if (col < g.N && rq16 < g.ldoT) {
                unsigned char* dst = reinterpret_cast<unsigned char*>(g.outT) + (size_t)col * g.ldoT * 2;
                if (g.outT_f8 == 2) {   // (kperm64: eight consecutive k from a multiple of eight stay consecutive)
                    *reinterpret_cast<u32x2*>(dst + kperm64(rq16)) = u32x2{pk[0], pk[1]};
                    *reinterpret_cast<u32x2*>(dst + kperm64(rq16 + 8)) = u32x2{pk[2], pk[3]};
                } else {
                    *reinterpret_cast<u32x4*>(dst + rq16) = u32x4{pk[0], pk[1], pk[2], pk[3]};
                }
            }
        }
    } else if (g.outT) {
        // bf16 pieces: the same transpose on the two dwords of a lane's four rows -- 16 consecutive rows = 32 bytes per lane and
        // piece, a whole 128-byte line per column of the wave's tile in two 16-byte stores where there were four 8-byte ones
        const int np = (g.outT_pieces == 3) ? 3 : 1;
        const float tsign = g.outT_neg ? -1.f : 1.f;   // (the pieces of -x are minus the pieces of x)
        const int rq16 = rowq + slot * 12;             // m0 + wm WM + 16 slot
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            const int col = colb + ni * 16;
            float v[4][4];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[mi][r] = (rowq + mi * 16 + r < g.M) ? tsign * xv[mi][ni][r] : 0.f;
            for (int j = 0; j < np; ++j) {
                uint32_t px[4], py[4];
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    px[mi] = pack_bf16x2(v[mi][0], v[mi][1]); py[mi] = pack_bf16x2(v[mi][2], v[mi][3]);
                    if (j + 1 < np) {   // residual of the piece just packed: exact in fp32
                        v[mi][0] -= bf16_bits_to_f32(px[mi] & 0xFFFFu); v[mi][1] -= bf16_bits_to_f32(px[mi] >> 16);
                        v[mi][2] -= bf16_bits_to_f32(py[mi] & 0xFFFFu); v[mi][3] -= bf16_bits_to_f32(py[mi] >> 16);
                    }
                }
                {   // (every lane takes part: the swaps sit outside the bounds checks)
                    auto a0 = __builtin_amdgcn_permlane32_swap(px[0], px[2], false, false);
                    auto a1 = __builtin_amdgcn_permlane32_swap(px[1], px[3], false, false);
                    auto a2 = __builtin_amdgcn_permlane16_swap(a0[0], a1[0], false, false);
                    auto a3 = __builtin_amdgcn_permlane16_swap(a0[1], a1[1], false, false);
                    px[0] = a2[0]; px[1] = a2[1]; px[2] = a3[0]; px[3] = a3[1];
                    auto b0 = __builtin_amdgcn_permlane32_swap(py[0], py[2], false, false);
                    auto b1 = __builtin_amdgcn_permlane32_swap(py[1], py[3], false, false);
                    auto b2 = __builtin_amdgcn_permlane16_swap(b0[0], b1[0], false, false);
                    auto b3 = __builtin_amdgcn_permlane16_swap(b0[1], b1[1], false, false);
                    py[0] = b2[0]; py[1] = b2[1]; py[2] = b3[0]; py[3] = b3[1];
                }
                if (col < g.N && rq16 < g.ldoT) {
                    uint16_t* dst = g.outT + (size_t)j * g.outT_plane + (size_t)col * g.ldoT + rq16;
                    *reinterpret_cast<u32x4*>(dst) = u32x4{px[0], py[0], px[1], py[1]};
                    *reinterpret_cast<u32x4*>(dst + 8) = u32x4{px[2], py[2], px[3], py[3]};
                }
            }
        }
    }

#ifdef KURBM_STAMPS
    KURBM_STAMP(te[1]);
#endif
    // (c) fp32 copy of the value plane (persistent chain, test hooks): 16 lanes x 4 B per row
    if (g.out_f32) {
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = rowq + mi * 16 + r, col = colb + ni * 16;
                    if (row < g.M && col < g.N) g.out_f32[(size_t)row * g.ldo32 + col] = xv[mi][ni][r];
                }
    }

    // (d) row-major bf16 plane [M][ldo]: neighbouring lanes hold neighbouring columns; pair them up (even
    //     lane takes rows r = 0, 2, odd lane r = 1, 3), 4-byte writes into a bf16 patch of the whole tile,
    //     then whole rows leave as 16-byte chunks.  Columns past N (k padding of the next GEMM) are zeros.
    if (g.out && g.out_bytes) {
        // a 0/1 sample as a BYTE plane (0x40 = one; the next half step's A operand, k-permuted: kurbm_device.h kperm64): bytes
        // into a patch of the tile, whole rows out as 16-byte chunks; columns past N are zeros
        constexpr int PROWB = BN + 16;
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            const bool col_ok = colb + ni * 16 < g.N;
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    smem[(wm * WM + mi * 16 + slot * 4 + r) * PROWB + kperm64(wn * WN + ni * 16 + l15)] =
                        (col_ok && xv[mi][ni][r] != 0.f) ? (unsigned char)0x40 : (unsigned char)0;
        }
        __syncthreads();
        constexpr int CHB = BN / 16;   // 16-B chunks per row
        unsigned char* plane = reinterpret_cast<unsigned char*>(g.out);
#pragma unroll
        for (int q8 = 0; q8 < (BM * CHB + NT - 1) / NT; ++q8) {
            const int q = q8 * NT + tid, row = q / CHB, c = q % CHB;
            const int gr = m0 + row, gc = n0 + 16 * c;
            if (row < BM && gr < g.M && gc < g.ldo_cols)
                *reinterpret_cast<u32x4*>(plane + (size_t)gr * g.ldo + gc) = *reinterpret_cast<const u32x4*>(smem + row * PROWB + 16 * c);
        }
    } else if (g.out) {
        // Each lane rounds its own four rows of a column to bf16 first (two packed dwords: rows 0|1 and 2|3), trades them with the
        // neighbouring column's lane (DPP) and picks the two halves of one row with a v_perm_b32: three instructions per dword of
        // the patch.  The residual for the next piece comes from the same packed dwords.  (As float selects around the swap and a
        // pack per dword this phase was ~20 instructions per dword: 14 800 cycles of the Gaussian h -> v launch, stamps of round 4.)
        const int odd = l15 & 1;
        const uint32_t sel = odd ? 0x03020706u : 0x05040100u;   // {neighbour | own}: odd lane (nb.hi, own.hi), even lane (own.lo, nb.lo)
        const int npc = (g.out_pieces == 3) ? 3 : 1;     // a real-valued plane leaves as its three exact pieces
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)                  // columns past N (k padding of the next GEMM): zeros
            if (!(colb + ni * 16 < g.N)) {
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int r = 0; r < 4; ++r) xv[mi][ni][r] = 0.f;
            }
        for (int j = 0; j < npc; ++j) {
            if (j > 0) __syncthreads();                   // the previous piece has left the patch
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
#pragma unroll
                for (int mi = 0; mi < TM; ++mi) {
                    float (&x)[4] = xv[mi][ni];
                    const uint32_t p01 = pack_bf16x2(x[0], x[1]), p23 = pack_bf16x2(x[2], x[3]);
                    const uint32_t n01 = (uint32_t)__builtin_amdgcn_mov_dpp((int)p01, 0xB1, 0xF, 0xF, true);
                    const uint32_t n23 = (uint32_t)__builtin_amdgcn_mov_dpp((int)p23, 0xB1, 0xF, 0xF, true);
                    unsigned char* dst = smem + (wm * WM + mi * 16 + slot * 4 + odd) * PROW16 + 2 * (wn * WN + ni * 16 + (l15 & ~1));
                    *reinterpret_cast<uint32_t*>(dst) = __builtin_amdgcn_perm(n01, p01, sel);                // row 0 (even lane) / 1 (odd)
                    *reinterpret_cast<uint32_t*>(dst + 2 * PROW16) = __builtin_amdgcn_perm(n23, p23, sel);   // row 2 / 3
                    if (j + 1 < npc) {   // residual of the piece just written: exact in fp32
                        x[0] -= __uint_as_float(p01 << 16); x[1] -= __uint_as_float(p01 & 0xFFFF0000u);
                        x[2] -= __uint_as_float(p23 << 16); x[3] -= __uint_as_float(p23 & 0xFFFF0000u);
                    }
                }
            }
            __syncthreads();
            constexpr int CH = BN / 8;   // 16-B chunks per row
            uint16_t* plane = g.out + (size_t)j * g.out_plane;
#pragma unroll
            for (int q8 = 0; q8 < BM * CH / NT; ++q8) {
                const int q = q8 * NT + tid, row = q / CH, c = q % CH;
                const int gr = m0 + row, gc = n0 + 8 * c;
                if (gr < g.M && gc < g.ldo_cols)
                    *reinterpret_cast<u32x4*>(plane + (size_t)gr * g.ldo + gc) =
                        *reinterpret_cast<const u32x4*>(smem + row * PROW16 + 16 * c);
                else if (gr < g.out_rows_pad && gc < g.ldo_cols)      // (rows past the batch that a k = batch consumer will read: zeros)
                    *reinterpret_cast<u32x4*>(plane + (size_t)gr * g.ldo + gc) = u32x4{0u, 0u, 0u, 0u};
            }
        }
    }
    KURBM_STAMP(ts[4]);
#if defined(KURBM_STAMPS) && !defined(KURBM_STAMPS_LOOP)   // (KURBM_STAMPS_LOOP: keep the k loop's micro-step sums)
    tu[0] = te[0] - ts[3]; tu[1] = te[1] - te[0]; tu[2] = te[2] ? te[2] - te[1] : 0; tu[3] = te[3] ? te[3] - te[2] : 0;
    tu[4] = te[3] ? ts[4] - te[3] : ts[4] - te[1]; tu[5] = 0;
#endif
    KURBM_STAMP_OUT();
}

// ------------------------------------------------------------------------------------
// launcher
// ------------------------------------------------------------------------------------
// one (pieces, epilogue, noise) combination: tile configuration by g.cfg (0: 128 x 128; 2: 256 x 64, half steps only); only
// the combinations the host plans are instantiated
#define KURBM_LAUNCH(kern, grid, block, shmem, st, g) hipLaunchKernelGGL(kern, grid, block, shmem, st, g)
template <int PBN, int E, int NZ>
static hipError_t launch_pb(const GemmArgsB& g, int nblk, hipStream_t st) {
    // byte A planes: ONE segment of PBN pieces, one k slice -- those kernels' loops do not walk a tile list
    // (the statistics GEMM on byte planes: its own fixed walk, any number of k slices -- checked where it is launched)
    if (g.a_bytes && E != EPI_SLAB && (g.nseg != 1 || g.nsplit != 1 || (int)((g.seg_codes >> 2) & 3u) != PBN)) return hipErrorInvalidValue;
    if (g.slab_t && !(g.a_bytes && E == EPI_SLAB)) return hipErrorInvalidValue;   // (only the plain byte-plane statistics kernel writes transposed)
    if constexpr (E == EPI_HALFSTEP && PBN == 3 && (NZ == NOISE_BERNOULLI || NZ == NOISE_NONE)) {
        if (g.rp) {   // the score's half steps: (samples AND) the softplus row sums of their rows
            if (!g.rowpart) return hipErrorInvalidValue;
            if (g.cfg == 2 && g.a_bytes) KURBM_LAUNCH((k_gemm_pb<256, 64, 4, 2, 64, PBN, E, NZ, true, true>), dim3(nblk), dim3(768), 0, st, g);
            else if (g.cfg == 2) KURBM_LAUNCH((k_gemm_pb<256, 64, 4, 2, 64, PBN, E, NZ, false, true>), dim3(nblk), dim3(768), 0, st, g);
            else if (g.cfg == 0 && g.a_bytes) KURBM_LAUNCH((k_gemm_pb<128, 128, 2, 4, 64, PBN, E, NZ, true, true>), dim3(nblk), dim3(768), 0, st, g);
            else if (g.cfg == 0) KURBM_LAUNCH((k_gemm_pb<128, 128, 2, 4, 64, PBN, E, NZ, false, true>), dim3(nblk), dim3(768), 0, st, g);
            else return hipErrorInvalidValue;
            return hipGetLastError();
        }
    }
    if (g.a_bytes) {   // byte A planes: the half steps of the x3 path, and the free-energy GEMM of a 0/1 plane
        if constexpr (E == EPI_SOFTPLUS && PBN == 3) {
            if (g.cfg != 0) return hipErrorInvalidValue;
            KURBM_LAUNCH((k_gemm_pb<128, 128, 2, 4, 64, PBN, E, NZ, true>), dim3(nblk), dim3(768), 0, st, g);
            return hipGetLastError();
        } else if constexpr (E == EPI_HALFSTEP) {
            if (g.cfg == 2) KURBM_LAUNCH((k_gemm_pb<256, 64, 4, 2, 64, PBN, E, NZ, true>), dim3(nblk), dim3(768), 0, st, g);
            else if (g.cfg == 0) KURBM_LAUNCH((k_gemm_pb<128, 128, 2, 4, 64, PBN, E, NZ, true>), dim3(nblk), dim3(768), 0, st, g);
            else return hipErrorInvalidValue;
            return hipGetLastError();
        } else if constexpr (E == EPI_SLAB && PBN == 3) {
            if (g.cfg != 2) return hipErrorInvalidValue;
            if (g.walk3) {
                if (g.slab_t) return hipErrorInvalidValue;
                KURBM_LAUNCH((k_gemm_pb<256, 64, 4, 2, 64, PBN, E, NZ, true>), dim3(nblk), dim3(768), 0, st, g);
            } else {
                // the plain byte-plane walk (ABP): one segment of three pieces, k slices of whole 128-deep blocks, no in-launch reduction
                if (g.nseg != 1 || (int)((g.seg_codes >> 2) & 3u) != 3 || (g.kt_per_split & 1) || g.f8pos) return hipErrorInvalidValue;
                KURBM_LAUNCH((k_gemm_pb<256, 64, 4, 2, 64, PBN, E, NZ, true, true>), dim3(nblk), dim3(768), 0, st, g);
            }
            return hipGetLastError();
        } else {
            return hipErrorInvalidValue;
        }
    }
    if (g.cfg == 2) {
        if constexpr (E == EPI_HALFSTEP || (E == EPI_SLAB && PBN == 3))
            KURBM_LAUNCH((k_gemm_pb<256, 64, 4, 2, 64, PBN, E, NZ>), dim3(nblk), dim3(768), 0, st, g);
        else return hipErrorInvalidValue;
    } else if (g.cfg == 0) {
        KURBM_LAUNCH((k_gemm_pb<128, 128, 2, 4, 64, PBN, E, NZ>), dim3(nblk), dim3(768), 0, st, g);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_gemm_pb(int epi, const GemmArgsB& g_in, hipStream_t st) {
    GemmArgsB g = g_in;
    {   // diagnostic build: launch n after the buffer was set writes at base + n * 4096 entries
        static unsigned long long* last_base = nullptr;
        static int nth = 0;
        unsigned long long* base = get_stamp_buffer();
        if (base != last_base) { last_base = base; nth = 0; }
        g.stamps = base ? base + (size_t)(nth++) * 65536 : nullptr;
    }
    {   // B: one descriptor for both sets (base = the lower pointer; both lie in one workspace).  A: a descriptor per set
        const uint16_t* b1 = g.B1 ? g.B1 : g.B0;
        g.baseA = g.A0;
        g.baseB = g.B0 < b1 ? g.B0 : b1;
        const size_t ob0 = (size_t)(g.B0 - g.baseB) * 2, ob1 = (size_t)(b1 - g.baseB) * 2;
        const size_t lim = 0x7FFFFFFFull;   // scalar offset + plane + k must stay below 4 GiB
        if (ob0 > lim || ob1 > lim || 4 * g.a_plane0 > lim || 4 * g.a_plane1 > lim) return hipErrorInvalidValue;
        g.offA0 = g.offA1 = 0u; g.offB0 = (uint32_t)ob0; g.offB1 = (uint32_t)ob1;
    }
    g.side = (g.prob_f32 != nullptr) || (g.out_u != nullptr);
    const int nblk = g.grid_m * g.grid_n * g.nsplit;
    {   // XCD blocks: the factorisation 8 = xz * xr * xc (xz | nsplit, xr | grid_m, xc | grid_n) with the fewest operand rows
        // per XCD, weighted by the tiles a k-tile loads (one A tile per segment, npb B tiles)
        g.xcd_r = g.xcd_c = 0;
        if (g.xcd2d && nblk % 8 == 0) {
            int wa = 0, wb = 0;
            for (int sgm = 0; sgm < g.nseg; ++sgm) { wa += 1; wb += (int)((g.seg_codes >> (5 * sgm + 2)) & 3u); }
            const int bmr = (g.cfg == 2) ? 256 : 128, bnr = (g.cfg == 2) ? 64 : 128;
            long long best = -1;
            for (int xz = 1; xz <= 8; xz *= 2)
                for (int xr = 1; xr * xz <= 8; xr *= 2) {
                    const int xc = 8 / (xz * xr);
                    if (g.nsplit % xz || g.grid_m % xr || g.grid_n % xc) continue;
                    const long long cost = (long long)(g.nsplit / xz) * ((long long)(g.grid_m / xr) * bmr * wa + (long long)(g.grid_n / xc) * bnr * wb);
                    if (best < 0 || cost < best) { best = cost; g.xcd_r = xr; g.xcd_c = xc; }
                }
        }
    }
    // shared B staging: a three-piece A operand against three-piece weights, (piece p) x (pieces 0 .. 2 - p), one k slice
    {
        // the PAIRED walk (k_gemm_pb, "BSP"): a three-piece A operand against three-piece B rows -- segments (A piece p) x (B pieces
        // 0 .. 2 - p) of ONE operand set, walked segment-fastest -- on 128 x 128 tiles as two tiles per k position: the half steps on a
        // real-valued batch, and the statistics GEMM whose only segments are these three (the negative phase of Gaussian visibles),
        // any number of k slices of whole positions
        const bool tri = g.pb_max == 3 && !g.a_bytes && g.nseg == 3 && g.seg_fastest && ((g.seg_codes >> 4) & 1ull) == ((g.seg_codes >> 9) & 1ull) &&
                         ((g.seg_codes >> 4) & 1ull) == ((g.seg_codes >> 14) & 1ull) &&
                         (g.seg_codes & 0x7FFFull & ~0x4210ull) == ((0ull | (3ull << 2)) | ((1ull | (2ull << 2)) << 5) | ((2ull | (1ull << 2)) << 10));
        g.bshare = (tri && g.pair_ok && g.cfg == 0 && (epi == EPI_HALFSTEP ? (g.nsplit == 1 && !((g.seg_codes >> 4) & 1ull)) : (epi == EPI_SLAB && !g.f8pos)) &&
                    g.kt_per_split % 3 == 0 && g.kt_total % 3 == 0) ? 2 : 0;
    }
    if (g.a_tr && !(g.bshare == 2 && epi == EPI_SLAB)) return hipErrorInvalidValue;   // (only that walk reads a row-major A operand)
    g.walk3 = (epi == EPI_SLAB && g.pb_max == 3 && g.f8pos && g.nseg == 2 && ((g.seg_codes >> 7) & 3u) == 3u &&
               g.kt_per_split % 3 == 0 && g.kt_total % 3 == 0) ? 1 : 0;
    {   // the block mapping's divisors as multiply-high constants: exact while dividend x divisor < 2^32 (every grid of the
        // BASELINE configs by orders of magnitude); beyond that the kernel divides
        if (nblk <= 0) return hipErrorInvalidValue;
        {
            const unsigned long long lim = 0xFFFFFFFFull;
            unsigned long long worst;
            if (g.xcd_r) {
                const unsigned long long rl = (unsigned long long)(g.grid_m / g.xcd_r), cl = (unsigned long long)(g.grid_n / g.xcd_c);
                worst = (unsigned long long)(nblk / 8 + 1) * (rl * cl);
                const unsigned long long w2 = (rl * cl) * (rl > cl ? rl : cl);
                if (w2 > worst) worst = w2;
            } else {
                const unsigned long long tm = (unsigned long long)g.grid_m * (unsigned long long)g.grid_n;
                worst = (unsigned long long)nblk * tm;
                const unsigned long long w2 = tm * (unsigned long long)(g.grid_m > g.grid_n ? g.grid_m : g.grid_n);
                if (w2 > worst) worst = w2;
            }
            g.map_slow = (worst >= lim || g.map_force) ? 1 : 0;
        }
        auto inv = [](int d) { return (uint32_t)((0x100000000ull + (uint64_t)d - 1) / (uint64_t)d); };   // d = 1: 2^32 wraps to 0, which div_magic reads as "divisor 1"
        if (g.xcd_r) {
            const int rc = g.xcd_r * g.xcd_c;
            g.map_lrc = __builtin_ctz(rc); g.map_lxc = __builtin_ctz(g.xcd_c);
            g.map_rl = g.grid_m / g.xcd_r; g.map_cl = g.grid_n / g.xcd_c; g.map_zl = g.nsplit / (8 / rc);
            g.map_inv_a = inv(g.map_rl * g.map_cl);
            g.map_inv_b = inv(g.m_fastest ? g.map_rl : g.map_cl);
        } else {
            g.map_lrc = g.map_lxc = 0; g.map_rl = g.map_cl = g.map_zl = 0;
            g.map_inv_a = inv(g.grid_m * g.grid_n);
            g.map_inv_b = inv(g.m_fastest ? g.grid_m : g.grid_n);
        }
    }
    // cfg 2: 256 x 64 tiles for the half steps (A tile 32 KB + three 8-KB pieces of B = 56 KB per k-tile instead of 64 KB
    // for the same MFMAs: the k loop moves with the bytes a CU takes in).  pb_max = 1: every segment has ONE piece of B (the
    // rounded-bf16 path, kurbm_cd_step_bf16): the same kernel with a k-tile of A + one B tile.
#define KURBM_PB_ANY(PBN, E, NZ) \
    if (epi == E && (E != EPI_HALFSTEP || g.noise == NZ)) return launch_pb<PBN, E, NZ>(g, nblk, st);
    if (g.pb_max == 1) {
        KURBM_PB_ANY(1, EPI_HALFSTEP, NOISE_NONE)
        KURBM_PB_ANY(1, EPI_HALFSTEP, NOISE_BERNOULLI)
        KURBM_PB_ANY(1, EPI_HALFSTEP, NOISE_GAUSSIAN)
        KURBM_PB_ANY(1, EPI_SLAB, NOISE_NONE)
        return hipErrorInvalidValue;
    }
    KURBM_PB_ANY(3, EPI_HALFSTEP, NOISE_NONE)
    KURBM_PB_ANY(3, EPI_HALFSTEP, NOISE_BERNOULLI)
    KURBM_PB_ANY(3, EPI_HALFSTEP, NOISE_GAUSSIAN)
    KURBM_PB_ANY(3, EPI_SLAB, NOISE_NONE)
    KURBM_PB_ANY(3, EPI_SOFTPLUS, NOISE_NONE)
#undef KURBM_PB_ANY
    return hipErrorInvalidValue;
}

}  // namespace kurbm
