// kurbm_kernels.h -- argument blocks shared by the kernels and the C-ABI host layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kurbm {

enum { ACT_SIGMOID = 0, ACT_RELU = 1, ACT_LINEAR = 2 };
enum { NOISE_NONE = 0, NOISE_BERNOULLI = 1, NOISE_GAUSSIAN = 2 };
enum { EPI_HALFSTEP = 0, EPI_SLAB = 1, EPI_SOFTPLUS = 2 };
// operand layouts: VH  A=[m][k] B=[k][n];  HV  A=[m][k] B=[n][k];  OUTER  A=[k][m] B=[k][n]
enum { LAYOUT_VH = 0, LAYOUT_HV = 1, LAYOUT_OUTER = 2 };
// tile configurations (BM x BN, waves as WAVES_M x WAVES_N); see tile_shape()
enum { CFG_128x128 = 0, CFG_128x112 = 1, CFG_112x128 = 2, CFG_128x64 = 3, CFG_64x128 = 4, CFG_64x64 = 5,
       CFG_64x112 = 6, CFG_112x64 = 7, CFG_COUNT = 8 };

struct RngArgs {
    uint32_t seed_lo, seed_hi;
    uint32_t stream_id, step;
    uint64_t row0;
};

struct GemmArgs {
    // operands; segment 1 (A1,B1) enters with sign -1 (statistics GEMM only)
    const float* A0;
    const float* B0;
    const float* A1;
    const float* B1;
    int lda, ldb;
    int M, N, K;        // K = extent of k per segment
    int nseg;           // 1, or 2 for the statistics GEMM
    int nkt;            // FULL k-tiles per segment = K / 32 (the K % 32 tail is handled apart)
    int kt_total;       // full k-tiles over all segments = nseg * nkt
    int kt_per_split;   // full k-tiles per split-K slice
    int nsplit;         // slices; the last one also takes the tail tiles
    int grid_m, grid_n;
    int m_fastest;      // tile order inside a k-slice: 1 = m index fastest (see k_gemm)
    int tile_major;     // 1: k-slices of a tile are adjacent in the remapped block order (one XCD)
    // half-step epilogue
    const float* bias;
    float* out_sample;
    float* out_prob;
    float* out_u;       // debug: the uniforms used (nullable)
    int ldo;
    const float* ref;   // nullable: column partials of (ref - out) are written to colpart
    int ldref;
    float* colpart;     // [grid_m][ld_colpart]
    int ld_colpart;
    int act, noise;
    RngArgs rng;
    // slab epilogue
    float* slab;        // [nsplit][M][ld_slab]
    size_t slab_stride;
    int ld_slab;
    // softplus epilogue
    float* rowpart;     // [grid_n][ld_rowpart]
    int ld_rowpart;
    // diagnostic build only (KURBM_STAMPS): 16 x u64 per wave
    unsigned long long* stamps;
    int dbg_off;        // ablation mask: 1 fetches, 2 parks, 4 fragment reads, 8 barrier (results are garbage)
};

// Diagnostic hook: the next GEMM launches write their s_memtime stamps here (null = off).
void set_stamp_buffer(unsigned long long* p);
void set_debug_off(int mask);

struct ReduceArgs {
    const float* slab;
    size_t slab_stride;
    int nslab, ld_slab;
    int n_vis, n_hid, ldw;   // n_vis: rows of the slabs / of W handled by this launch
    int n_vis_bias;          // visible units of the bias partials when that differs (row-range launch); 0 = n_vis
    int nblk_w;
    // tile-ordered visit (tile_bm > 0): nblk_w = grid_m * grid_n * parts blocks, same XCD order as the GEMM
    int tile_bm, tile_bn, grid_m, grid_n, parts, m_fastest;
    float lr;
    float* W;           // nullable: W += lr * dW
    float* delta_w;     // nullable: dense [n_vis][n_hid]
    const float* part_h;
    int nrow_tiles_h, ld_part_h;
    float* b_h;
    float* delta_bh;
    const float* part_v;
    int nrow_tiles_v, ld_part_v;
    const float* part_v2;    // nullable: nrow_tiles_v2 more rows of visible partials, summed AFTER those of part_v as if
    int nrow_tiles_v2;       // they followed them in memory (the positive rows may live in resident data planes)
    float* b_v;
    float* delta_bv;
    // k_reduce_apply_split (kurbm_bf16.hip): the updated weights also leave as bf16 pieces, both orientations
    uint16_t* Wb;         // [pieces][n_vis][ldWb]   nullable
    uint16_t* Wtb;        // [pieces][n_hid][ldWtb]
    int ldWb, ldWtb, pieces;
    int wtb_k_ext;        // k_reduce_apply_split: k columns of Wtb (from its pointer) this launch may write; 0 = ldWtb.  A launch on a
                          // row RANGE of W writes its own rows' columns only (the last range also the zero k padding behind them)
    int tile_rows;        // k_reduce_apply_split: forced tile height 16 / 32 / 64 (ctx knob KURBM_REDUCE_TR); 0 = by the grid it makes
    size_t planeWb, planeWtb;
};

// Bias partials [row tiles][columns] -> column sums (k_reduce_apply, k_reduce_apply_split, and the prologue of the statistics
// GEMM when it reduces its own slabs): one WAVE per 8 columns, lane = 8 rg + column.  A column's rows are spread over 8 lanes
// (row groups), each with up to 8 loads in flight per pass -- one thread per column walked 128 rows of partials in 16
// dependent passes of ~0.6 us, the longest chain of the whole launch, and it began last.  Group sums and their total in
// double, the total by an xor-shuffle tree ((g0 + g1) + (g2 + g3)) + ((g4 + g5) + (g6 + g7)): a fixed order, bit-reproducible,
// no LDS and no barrier.  Callers place this work FIRST in their grid.
constexpr int BIAS_COLS = 32;     // columns per 256-thread block (4 waves)
__host__ __device__ static inline int bias_waves(const ReduceArgs& a) { return (a.n_hid + (a.n_vis_bias ? a.n_vis_bias : a.n_vis) + 7) / 8; }
static inline int bias_blocks(const ReduceArgs& a) { return (bias_waves(a) + 3) / 4; }
#if defined(__HIPCC__)
__device__ __forceinline__ void bias_colsum_wave(const ReduceArgs& a, int group, int lane) {
    const int cl = lane & 7, rg = lane >> 3;
    const int q = group * 8 + cl;
    const int nvb = a.n_vis_bias ? a.n_vis_bias : a.n_vis;
    const bool hid = q < a.n_hid, vis = !hid && q < a.n_hid + nvb;
    const float* part = hid ? a.part_h : (vis ? a.part_v : nullptr);
    const float* part2 = vis ? a.part_v2 : nullptr;
    const int n1 = hid ? a.nrow_tiles_h : a.nrow_tiles_v, n2 = part2 ? a.nrow_tiles_v2 : 0;
    const int ld = hid ? a.ld_part_h : a.ld_part_v, col = hid ? q : q - a.n_hid;
    double u = 0.;
    if (part) {
        const int ntiles = n1 + n2;
        for (int j0 = 0; j0 < ntiles; j0 += 64) {
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int i = j0 + rg + 8 * e;
                x[e] = (i < ntiles) ? (i < n1 ? part[(size_t)i * ld + col] : part2[(size_t)(i - n1) * ld + col]) : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) u += (double)x[e];
        }
    }
    u += __shfl_xor(u, 8);
    u += __shfl_xor(u, 16);
    u += __shfl_xor(u, 32);
    if (rg == 0 && part) {
        const float v = (float)u;
        if (hid) {
            if (a.delta_bh) a.delta_bh[col] = v;
            if (a.b_h) a.b_h[col] += a.lr * v;
        } else {
            if (a.delta_bv) a.delta_bv[col] = v;
            if (a.b_v) a.b_v[col] += a.lr * v;
        }
    }
}
#endif

// The peer exchange (kurbm_peer.hip): where the reduce / apply launch reads the SUMMED packed statistics when they come from the
// ranks' exchange buffers instead of slabs -- rows [q band_rows, (q + 1) band_rows) of dW from rank q's `sum` region, once that
// rank's `summed` flag has reached `epoch` (bounded wait; a timeout sets bit 1 of *status and skips the tile).
constexpr int PEER_MAX = 8;
struct PeerSrc {
    const float* sum[PEER_MAX];
    const unsigned* summed[PEER_MAX];
    const float* own_sum;              // this rank's own region: the bias tail, summed by every rank itself
    unsigned* status;
    unsigned long long timeout_ticks;  // of the 100 MHz constant clock
    unsigned epoch;
    int band_rows;                     // 0: no peer source (the ordinary launch)
    int nranks;
};

struct ApplyArgs {
    const float* delta;
    float* W;
    float* b_h;
    float* b_v;
    int n_vis, n_hid, ldw;
    float lr;
};

// The score of fit(verbose = 1), rbm.py:225-233: F(v) and F(v') per row from the softplus row partials of two GEMMs, then
// mean |F(v) - F(v')| into ONE device float (k_score_rows, k_score_mean: kurbm_kernels.hip)
struct ScoreArgs {
    const float* v;        // [rows][ldv]  the batch
    const float* v1;       // [rows][ldv1] its one-step reconstruction v' (fp32 plane), or:
    const unsigned char* v1b;   // v' as a byte plane of 0/1 values (0x40 = one, k-permuted: kperm64), ldv1b bytes per row; then v1 is unused
    int ldv1b;
    const float* b_v;
    const float* rowpart;  // [ncol_tiles][ld_rowpart] of v
    const float* rowpart1; // ... of v'
    float* F;              // nullable [2 * rows]: F(v) then F(v')
    float* absdiff;        // [rows]
    float* score;          // [1]
    int rows, n_vis, ldv, ldv1, ncol_tiles, ncol_tiles1, ld_rowpart;   // (column tiles of v's partials, of v''s)
};
hipError_t launch_score(const ScoreArgs& a, hipStream_t st);

struct FinishArgs {
    const float* v;
    const float* b_v;
    const float* rowpart;
    float* F;
    int rows, n_vis, ldv, ncol_tiles, ld_rowpart;
};

// bf16 NT GEMM (k_gemm_pb, kurbm_x3.hip): A [M][lda], B [N][ldb] bf16, k contiguous and zero-padded to 128.
// The k range is a list of up to MAX_SEG SEGMENTS of K elements each.  Segment s multiplies piece ia of
// operand set 0 or 1 of A with pieces 0 .. npb-1 of the same set of B (pieces = the bf16 hi / mid / lo parts of an
// fp32 plane, `*_plane` elements apart; one piece on the rounded-bf16 path); set 1 is the negative phase of the
// statistics, whose B planes (h_neg, transposed) are stored negated by the half step that makes them (outT_neg).
// seg_codes packs one 5-bit code per segment: bits 0-1 ia, bits 2-3 npb, bit 4 set.
constexpr int MAX_SEG = 12;
struct GemmArgsB {
    // ---- the WALK: everything a wave reads between its first instruction and its first request for data comes first and
    // together (three 64-byte lines of the argument segment): the scalar loads of the kernel's entry are then a few wide
    // ones, not a chain of narrow ones -- each costs its latency even when it hits the scalar cache (DESIGN.md section 4,
    // round 3, "the entry of a launch")
    const uint16_t* A0;
    const uint16_t* A1;
    // filled by launch_gemm_pb: one buffer descriptor per operand (base = the lower of the two sets'
    // pointers) and each set's byte offset from it, so that a tile is a scalar offset, never a pointer
    const uint16_t* baseB;
    size_t a_plane0, a_plane1, b_plane0, b_plane1;
    unsigned long long seg_codes;
    uint32_t inv_nkt;     // floor(2^32 / nkt) + 1: segment of k-tile t = umulhi(t, inv_nkt)
    uint32_t inv_nseg;
    uint32_t offB0, offB1;
    // (launcher) the block mapping's divisions as shifts and multiply-high (divisors < 2^16, dividends < 2^16: exact): a wave's
    // way to its first request for data was ~600 instructions of set-up, and eight integer divisions were a third of them
    uint32_t map_inv_a, map_inv_b;      // ceil(2^32 / d): d = rl cl, then rl (m fastest) or cl | linear order: d = grid_m grid_n, then grid_m or grid_n
                                        // (exact while dividend x divisor < 2^32; the launcher sets map_slow = 1 for a grid beyond that)
    int map_lrc, map_lxc;               // log2 (xcd_r xcd_c), log2 xcd_c
    int map_rl, map_cl, map_zl;         // row tiles / column tiles / k slices per XCD block
    // k_gemm_pb: XCD-aware 2-D blocks (set by the launcher; xcd_r == 0: the linear order).  The 8 XCDs form a
    // (8 / (xcd_r xcd_c)) x xcd_r x xcd_c grid over (k slices, row tiles, column tiles); each takes one block of it, so the A
    // rows and B rows it pulls through its own L2 are a fraction of the matrix instead of all of one operand
    int xcd_r, xcd_c;
    // k_gemm_pb (kurbm_x3.hip) reads the code as: bits 0-1 ia, bits 2-3 npb = number of B pieces
    // (0 .. npb-1) multiplied with that A tile, bit 4 set; and may walk the segments FASTEST
    // (t = k-tile * nseg + segment), so that every split-K slice gets the same mix of light and heavy tiles
    int seg_fastest;
    // statistics GEMM with 0/1 data: segment 0 (v_pos^T h_pos, both operands 0/1) reads fp8 planes and runs on
    // v_mfma_scale_f32_16x16x128_f8f6f4 -- a k-tile of the same 128 BYTES per row is 128 deep, at twice the bf16 rate and
    // half the bytes per k.  The walk is then in UNITS of 128 k: one fp8 tile, then two 64-deep bf16 tiles of every other
    // segment (2 nseg - 1 tiles per unit; inv_nseg inverts that count)
    int f8pos;
    int lda, ldb;         // elements, multiples of 8
    int M, N, K;          // K per segment, multiple of 64 (the operands are zero-padded to 128)
    int nseg, nkt, kt_total, kt_per_split, nsplit;
    int grid_m, grid_n;
    int m_fastest;
    int a_bytes;          // the A operand is such a byte plane (lda bytes between its rows; one segment): k_gemm_pb<..., AB>
    int bshare;           // (launcher) 2: the PAIRED walk -- a real-valued A operand on three-piece B rows, segments (A piece p) x (B pieces
                          // 0 .. 2 - p) walked segment-fastest, as TWO tiles per k position on 128 x 128 tiles (k_gemm_pb, "BSP"); 0: not
    int map_slow;         // (launcher) 1: the block mapping divides (a grid too large for the multiply-high constants)
    int walk3;            // (launcher) f8pos with ONE other segment of three pieces: the tiles go fp8, 3-piece, 3-piece, ... in whole
                          // units per k slice, and both loops of the statistics kernel step through that pattern instead of decoding a tile list
    // ---- the rest
    const uint16_t* B0;
    const uint16_t* B1;
    const uint16_t* baseA;
    uint32_t offA0, offA1;
    int side;             // k_gemm_pb: some test plane (prob_f32 / out_u) is requested
    int pb_max;           // k_gemm_pb: most B pieces any segment multiplies (3: x3; 1: the rounded-bf16 path)
    int cfg;              // 0: 128 x 128 tile; 2: 256 x 64 (half steps)
    // half-step epilogue
    const float* bias;
    int act, noise;
    RngArgs rng;
    uint16_t* out;        // bf16 [M][ldo]   value plane (sample, or prob when noise = NONE); nullable
    int ldo;
    int out_pieces;       // k_gemm_pb: 1 = the plane is 0/1 (one exact piece); 3 = real-valued, hi / mid / lo, out_plane apart
    int out_bytes;        // the row-major plane of a 0/1 sample leaves as BYTES (0x40 = one), ldo bytes between its rows
    size_t out_plane;
    int ldo_cols;         // columns the row planes cover: ldo when `out` is set (pads are zeroed), else N
    int out_rows_pad;     // bf16 row-major plane: rows [M, out_rows_pad) are written as zeros (0: none) -- the statistics GEMM that reads the
                          // plane as its A operand with k = the batch rows walks the batch padded to 128 (k_gemm_pb, "ATR")
    uint16_t* outT;       // bf16 [N][ldoT]  the same plane transposed; nullable
    int ldoT;
    int outT_pieces;      // 1: outT = round-to-nearest bf16; 3: exact hi / mid / lo pieces, outT_plane apart
    int outT_f8;          // the transposed plane of a 0/1 sample leaves as fp8 bytes (e4m3: 1.0 = 0x38) instead of bf16, at the
                          // bf16 plane's row stride (2 ldoT bytes): operand of the fp8 positive statistics (f8pos)
    int outT_neg;         // the transposed plane holds MINUS the value plane (h_neg: the negative phase of the statistics
                          // is then a plain product -- no sign handling between memory and the MFMAs)
    size_t outT_plane;
    float* out_f32;       // fp32 copy of the value plane (persistent chain, tests); nullable
    float* prob_f32;      // fp32 probabilities next to a sampled plane (tests); nullable
    float* out_u;         // fp32 uniforms (tests); nullable
    int ldo32;
    float* colpart;
    int ld_colpart;
    float colsign;        // k_gemm_pb: colpart[bm][col] = colsign * (column sum of the value plane over the tile's rows)
    // slab epilogue
    float* slab;
    size_t slab_stride;
    int ld_slab;
    int slab_t;           // EPI_SLAB on a byte-plane A operand (k_gemm_pb "ABP"): the tile leaves TRANSPOSED -- M counts the slab's COLUMNS,
                          // N its rows (the positive statistics of real-valued data as h_pos^T x the pieces of v_pos^T)
    int a_tr;             // caller (paired walk of the statistics GEMM): A0 is a ROW-MAJOR plane [k][M], lda its leading dimension -- the
                          // tiles are staged as [k][m] and read by transposed LDS reads (k_gemm_pb, "ATR")
    int pair_ok;          // caller: a real-valued A operand on 128 x 128 tiles may walk two tiles per k position (ctx knob KURBM_X3_PAIR; bshare = 2)
    int map_force;        // caller: 1 = map blocks by division whatever the grid (ctx knob KURBM_MAP_SLOW: tests of that path)
    int xcd2d;            // caller: 1 = let the launcher pick such blocks (ctx knob KURBM_X3_XCD2D), 0 = the linear order
    // softplus epilogue (k_gemm_pb, free energy): rowpart[bn][row] = sum over the tile's columns of softplus(x + bias)
    float* rowpart;
    int ld_rowpart;
    int rp;               // EPI_HALFSTEP (Bernoulli draws, x3): 1 = also write those row partials (k_gemm_pb<..., RP>)
    // diagnostic build only (KURBM_STAMPS): 8 x u64 per workgroup (k_gemm_pb)
    unsigned long long* stamps;
};

// Test hook (kurbm_x3_dump_plane): decode one of the planes a bf16 / x3 step keeps in its workspace back to fp32 [rows][units].
// fmt: 0 bf16, 1 bytes (0x40 = one, k-permuted in groups of 64: kperm64), 2 fp8 e4m3 (0x38 = one; 0/1 planes only).
// transposed: the plane is [units][ld] (k = batch rows) instead of [rows][ld].  pieces > 1: the sum of `pieces` bf16 planes
// `plane` elements apart (hi + mid + lo, exact in fp32).  sign: -1 for planes stored negated (h_neg).
struct DumpArgs {
    const void* src;
    float* out;
    int rows, units, ld, ld_out, fmt, transposed, pieces;
    size_t plane;
    float sign;
};
hipError_t launch_dump_plane(const DumpArgs& a, hipStream_t st);

unsigned long long* get_stamp_buffer();
void tile_shape(int cfg, int* bm, int* bn);
hipError_t launch_gemm_pb(int epi, const GemmArgsB& g, hipStream_t st);
// pieces = 1: round to nearest bf16; 3: exact split x = hi + mid + lo, piece j at out + j * out_plane
// colpart (nullable): [ceil(rows / 64)][ld_colpart] column sums of each 64-row band of `in`
// 0/1 data (pieces = 1 only).  outT_f8: the transposed plane as fp8 bytes (1.0 = 0x38) at the bf16 plane's row stride;
// out_bytes: the row-major plane as bytes (0x40 = one), ldo BYTES between its rows
hipError_t launch_f32_to_bf16(const float* in, int rows, int cols, int ld_in, uint16_t* out, int ldo, int out_rows,
                              uint16_t* outT, int ldoT, int outT_rows, int pieces, size_t out_plane, size_t outT_plane,
                              float* colpart, int ld_colpart, hipStream_t st, int outT_f8 = 0, int out_bytes = 0);
// *flag |= 1 if some element is not exactly a bf16 value, |= 2 if some element is neither 0.0 nor 1.0
hipError_t launch_bf16_exact_check(const float* in, int rows, int cols, int ld_in, int* flag, hipStream_t st);
hipError_t launch_gemm(int layout, int cfg, int epi, const GemmArgs& g, hipStream_t st);
hipError_t launch_philox_uniform(float* out, int rows, int cols, int ld, const RngArgs& rng, hipStream_t st);
hipError_t launch_reduce_apply(const ReduceArgs& a, hipStream_t st);
hipError_t launch_reduce_apply_split(const ReduceArgs& a, hipStream_t st, const PeerSrc* peer = nullptr);
hipError_t launch_apply_delta(const ApplyArgs& a, hipStream_t st);
hipError_t launch_free_energy_finish(const FinishArgs& a, hipStream_t st);

}  // namespace kurbm

// kurbm_small.hip: the whole CD-1 step of a small RBM in one launch
namespace kurbm {
struct SmallArgs {
    float* W; float* b_h; float* b_v;
    const float* v;
    float* h_pos; float* v_neg; float* h_neg;     // workspace planes [rows][ldh] / [rows][ldn]
    float* h_posT; float* v_negT; float* h_negT;  // ... and transposed: [n_hid][ldt] / [n_vis][ldt], ldt = the batch rounded up to 16
    unsigned* bar;                                 // the grid barrier's words in the context's status block (kurbm_small.hip: grid_barrier)
    unsigned* status;
    unsigned long long timeout_ticks;              // of the 100 MHz constant clock
    RngArgs rng_h, rng_v;                          // the counters of the h_pos and v_neg sampling sites
    int n_vis, n_hid, ldw, rows, ldv, ldh, ldn, ldt;
    int which, gauss;
    float lr;
    float* score; float* F;    // k_score_small: the mean |F(v) - F(v')| (device float) and, nullable, F(v) then F(v') [2 rows]
    int local;                 // 1: the XCD-local schedule of phases 1-3 (kurbm_small.hip)
    unsigned xcc_map;          // ... and the XCD of group g (workgroups g, g + 8, ...) in nibble g, as the context's probe found it
};
hipError_t launch_cd1_small(const SmallArgs& a, int nblk, hipStream_t st);
hipError_t launch_xcc_probe(unsigned* out, int nblk, hipStream_t st);
hipError_t launch_score_small(const SmallArgs& a, int nblk, hipStream_t st);
}
// kurbm_peer.hip <-> kurbm_api.hip
struct kurbm_ctx;
struct kurbm_peer;
namespace kurbm {
unsigned* ctx_status_word(kurbm_ctx* ctx);
float* peer_delta(kurbm_peer* x);                                            // this rank's `delta` region: the step writes its packed sums here
int peer_geometry_ok(const kurbm_peer* x, int device, int n_vis, int n_hid); // 0 or KURBM_ERR_* (message set)
int peer_exchange_shot1(kurbm_peer* x, unsigned* status, hipStream_t st, PeerSrc* src);   // next epoch: sum the own band; fills `src` for shot 2
}  // namespace kurbm
