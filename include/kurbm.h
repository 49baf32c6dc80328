/*
 * kurbm.h -- C ABI of the MI355X (gfx950) RBM / DBN contrastive-divergence engine.
 *
 * This is the drop-in boundary for the hot path of tonandr/keras_unsupervised's
 * ku/ebm package.  The reference has no FFI of its own: its narrowest seam is the set of
 * seven K.function closures an RBM object owns (reference ku/ebm/rbm.py:48, :54, :76,
 * :127, :129, :132, :137) plus its three variables (rbm.py:30-40).  Each entry point
 * below cites the closure(s) it replaces.  The Python classes in
 * keras_unsupervised_amd/ebm bind these with ctypes (see INTEGRATION.md for the stub a
 * reference maintainer would add).
 *
 * Conventions (all entry points)
 *   - every pointer named v/h/W/b_*, out_*, delta, workspace is a DEVICE pointer into
 *     caller-owned memory; the library never allocates or frees device memory and never
 *     synchronises with the host;
 *   - matrices are row-major fp32 [rows, ld]; ld is in floats, ld % 4 == 0, ld >= columns;
 *     base addresses are 16-byte aligned.  Columns [cols, ld) are padding: inputs may hold
 *     anything there, outputs are left untouched there;
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream);
 *   - return value: 0 = KURBM_OK, < 0 = error; kurbm_last_error() returns a thread-local
 *     message for the last failing call on this thread;
 *   - a kurbm_ctx belongs to one device and must not be used from two threads at once.
 *
 * Random numbers: Philox4x32-10, key = (seed_lo, seed_hi),
 *   counter = (col, (row0 + row) >> 2, stream_id, step), word (row0 + row) & 3,
 *   u = bitcast_f32((word & 0x7FFFFF) | 0x3F800000) - 1.0f.   row0 % 4 == 0 is required.
 *   Normals (Gaussian visible units): Box-Muller over the planes stream_id and
 *   stream_id | 0x80000000.  The CPU statement of the same contract is oracle/philox.py.
 */
#ifndef KURBM_H
#define KURBM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KURBM_ABI_VERSION 5

typedef struct kurbm_ctx kurbm_ctx;
typedef struct kurbm_comm kurbm_comm; /* one RCCL communicator + the library's comm stream and events, per GPU */
typedef void* kurbm_stream_t; /* hipStream_t */

enum {
    KURBM_OK = 0,
    KURBM_ERR_ARG = -1,         /* bad pointer / shape / alignment */
    KURBM_ERR_HIP = -2,         /* a HIP runtime call failed */
    KURBM_ERR_UNSUPPORTED = -3, /* valid request this build does not implement */
    KURBM_ERR_WORKSPACE = -4,   /* workspace too small */
    KURBM_ERR_COMM = -5         /* RCCL missing or an RCCL call failed */
};

/* Activation applied to the pre-activation x = in.W(+T) + bias. */
enum {
    KURBM_ACT_SIGMOID = 0, /* Bernoulli units: rbm.py:47, :53, :124 */
    KURBM_ACT_RELU = 1,    /* Gaussian-mode hidden threshold: rbm.py:59, :86 */
    KURBM_ACT_LINEAR = 2   /* Gaussian visible mean: rbm.py:64-65, :143 */
};

/* What is drawn from the activated value p. */
enum {
    KURBM_NOISE_NONE = 0,      /* no draw: only out_prob is written (h_neg, rbm.py:124) */
    KURBM_NOISE_BERNOULLI = 1, /* out_sample = (u < p) ? 1 : 0   (rbm.py:46-47, :121-123) */
    KURBM_NOISE_GAUSSIAN = 2   /* out_sample = p + z, z ~ N(0,1) (rbm.py:64-66, :143-144) */
};

/* RBM.mode of the reference (rbm.py:14-16). */
enum { KURBM_MODE_VISIBLE_BERNOULLI = 0, KURBM_MODE_VISIBLE_GAUSSIAN = 1 };

/* The three variables of an RBM (rbm.py:30-40), resident on the device. */
typedef struct {
    int32_t n_vis;  /* rows of W */
    int32_t n_hid;  /* columns of W */
    int32_t ldw;    /* leading dimension of W in floats */
    int32_t _pad;
    float* W;       /* [n_vis, ldw]  rbm_weight      rbm.py:30-33 */
    float* b_h;     /* [n_hid]       rbm_hidden_bias rbm.py:34-37 */
    float* b_v;     /* [n_vis]       visible_bias    rbm.py:38-40 */
} kurbm_params;

/* One sampling site of the RNG contract. */
typedef struct {
    uint64_t seed;
    uint64_t row0;      /* global index of the first row; multiple of 4 */
    uint32_t stream_id;
    uint32_t step;
} kurbm_rng;

/* Options of one contrastive-divergence parameter update. */
typedef struct {
    int32_t k;          /* Gibbs iterations (reference: 1)                          */
    int32_t mode;       /* KURBM_MODE_*                                             */
    float lr;           /* hps['lr'], applied to batch SUMS (rbm.py:128,130,133)    */
    int32_t apply;      /* 1: W,b_h,b_v += lr*delta in place; 0: only emit delta    */
    float* delta_out;   /* nullable; packed [n_vis*n_hid | n_hid | n_vis] fp32 sums */
    float* v_chain;     /* nullable; persistent chain [rows, ldv], read then overwritten
                           with v_neg (extension; absent from the reference)         */
    uint64_t seed;
    uint64_t row0;      /* global row offset of this shard (data parallel); % 4 == 0 */
    uint32_t step;      /* parameter-update counter                                  */
    uint32_t chain;     /* chain id: stream ids are chain*64 + {2t, 2t-1}            */
    const void* v_planes;      /* nullable; x3 only: the bf16 planes kurbm_x3_convert_rows made of exactly the rows of
                                  v_batch -- the step then skips its own conversion (the data matrix is static for
                                  a whole fit(): rbm.py:211 slices the same V every epoch)               */
    uint64_t v_planes_stride;  /* kurbm_cd_epoch_x3: bytes between the planes of consecutive batches   */
} kurbm_cd_opts;

/* ---- context ------------------------------------------------------------------- */
int kurbm_abi_version(void);
const char* kurbm_last_error(void);
int kurbm_ctx_create(int device, kurbm_ctx** out);
void kurbm_ctx_destroy(kurbm_ctx* ctx);
/* Tuning / test hook: the library reads its experiment knobs (the KURBM_* variables listed in DESIGN.md) from the
 * environment once, in kurbm_ctx_create; this sets one of them (by its environment name, e.g. "KURBM_X3_TALL") on a
 * live context.  -1 = automatic where a knob has an automatic setting.  No production caller needs it. */
int kurbm_ctx_set_option(kurbm_ctx* ctx, const char* name, int value);
/* Sticky status bits of the context's kernels, read back from the device (THIS call synchronises with the device; the
 * others never do) and cleared.  Every device-side wait in this library is bounded; a wait that runs into its bound sets a bit
 * here and SKIPS its work instead of hanging the GPU.  Bit 1: the peer exchange (kurbm_peer_*) gave up waiting for a rank's flag.
 * Bit 2: a barrier of kurbm_cd_step_small timed out -- its grid was not resident (a CU mask, a device shared with a kernel
 * that never ends).  0 in every run this build has seen; the host classes read it at the end of every fit() and raise. */
int kurbm_ctx_status(kurbm_ctx* ctx, int* bits);

/* ---- RNG test hook: out[r, c] = u(row0 + r, c) under the contract above -------- */
int kurbm_philox_uniform(kurbm_ctx* ctx, float* out, int rows, int cols, int ld,
                         const kurbm_rng* rng, kurbm_stream_t stream);

/*
 * visible -> hidden half step.   Replaces transform_func (rbm.py:46-48 / :58-60), the body
 * of RBM.call (rbm.py:80-86) and, with noise = NONE, the h_neg expression (rbm.py:124).
 *   x = v.W + b_h ; p = act(x) ; out_prob = p (nullable) ; out_sample per `noise` (nullable)
 * ONE kernel launch: MFMA GEMM + fused bias / activation / Philox draw epilogue.
 */
int kurbm_half_step_vh(kurbm_ctx* ctx, const kurbm_params* p, const float* v, int rows, int ldv,
                       int act, int noise, const kurbm_rng* rng,
                       float* out_sample, float* out_prob, int ldh, kurbm_stream_t stream);

/*
 * hidden -> visible half step.   Replaces inv_transform_func (rbm.py:52-54 / :64-67) and
 * the v_neg expression of the CD graph (rbm.py:121-123 / :143-144).  W is read as stored
 * (never transposed in memory):   x = h.W^T + b_v.
 */
int kurbm_half_step_hv(kurbm_ctx* ctx, const kurbm_params* p, const float* h, int rows, int ldh,
                       int act, int noise, const kurbm_rng* rng,
                       float* out_sample, float* out_prob, int ldv, kurbm_stream_t stream);

/* Test hooks: the same launches, additionally writing the uniforms that were compared
 * against the probabilities (out_u, nullable, same shape/ld as the outputs). */
int kurbm_half_step_vh_dbg(kurbm_ctx* ctx, const kurbm_params* p, const float* v, int rows, int ldv,
                           int act, int noise, const kurbm_rng* rng, float* out_sample,
                           float* out_prob, float* out_u, int ldh, kurbm_stream_t stream);
int kurbm_half_step_hv_dbg(kurbm_ctx* ctx, const kurbm_params* p, const float* h, int rows, int ldh,
                           int act, int noise, const kurbm_rng* rng, float* out_sample,
                           float* out_prob, float* out_u, int ldv, kurbm_stream_t stream);

/* Bytes of scratch kurbm_cd_step / kurbm_free_energy need for batches of <= rows. */
size_t kurbm_workspace_bytes(kurbm_ctx* ctx, int rows, int n_vis, int n_hid, int k);

/*
 * One CD-k parameter update from one Gibbs chain.   Replaces rbm_weight_update_func,
 * hidden_bias_update_func and visible_bias_update_func (rbm.py:125-134) evaluated on ONE
 * chain (update_mode "fused"); `chain` selects an independent chain so the host can also
 * replay the reference's three-chain sequence (rbm.py:214-216).
 * Launch sequence: half_vh(sample) ; k x [half_hv(sample) ; half_vh] ; outer-product
 * split-K GEMM over [v_pos ; -v_neg]^T.[h_pos ; h_neg] ; reduce (+ apply).
 * `which` is a bit mask of what `apply` touches: 1 = W, 2 = b_h, 4 = b_v (7 = all).
 */
int kurbm_cd_step(kurbm_ctx* ctx, const kurbm_params* p, const float* v_batch, int rows, int ldv,
                  const kurbm_cd_opts* opts, int which, void* workspace, size_t workspace_bytes,
                  kurbm_stream_t stream);

/*
 * One pass of the training loop over a resident data matrix: for each contiguous batch, in order,
 * remainder last (rbm.py:110-111, :163, :211, :218), one kurbm_cd_step with opts->step advancing by
 * one per batch.  Replaces the Python-level batch loop of RBM.fit for update_mode "fused" when no
 * per-step score is requested; identical launches, no host round trip per step.  A persistent
 * chain (opts->v_chain) must hold batch_size rows.  Returns the number of steps taken (>= 0) or an
 * error code (< 0).
 */
int kurbm_cd_epoch(kurbm_ctx* ctx, const kurbm_params* p, const float* V, int n_rows, int ldv, int batch_size,
                   const kurbm_cd_opts* opts, void* workspace, size_t workspace_bytes, kurbm_stream_t stream);

/* params += lr * delta for a packed delta (after the data-parallel all-reduce). */
int kurbm_apply_delta(kurbm_ctx* ctx, const kurbm_params* p, const float* delta, float lr,
                      int which, kurbm_stream_t stream);

/*
 * Free energy F(v) = -(v.b_v + sum_j softplus((v.W + b_h)_j)).   Replaces
 * free_energy_func (rbm.py:73-76), with the overflow-free softplus.
 */
int kurbm_free_energy(kurbm_ctx* ctx, const kurbm_params* p, const float* v, int rows, int ldv,
                      float* F, void* workspace, size_t workspace_bytes, kurbm_stream_t stream);

/*
 * Sufficient-statistics GEMM alone (bench / test hook for the dominant kernel):
 *   delta[0 : n_vis*n_hid] = v_pos^T.h_pos - v_neg^T.h_neg   (dense, ld = n_hid)
 */
int kurbm_outer_delta(kurbm_ctx* ctx, const float* v_pos, const float* h_pos, const float* v_neg,
                      const float* h_neg, int rows, int n_vis, int n_hid, int ldv, int ldh,
                      float* delta_w, void* workspace, size_t workspace_bytes,
                      kurbm_stream_t stream);

/*
 * The split-K slab GEMM of kurbm_outer_delta WITHOUT the slab reduction (measurement hook: this
 * is the dominant kernel of a CD step, timed alone for the roofline line of bench.py).  The
 * partial sums stay in `workspace`.
 */
int kurbm_outer_partial(kurbm_ctx* ctx, const float* v_pos, const float* h_pos, const float* v_neg,
                        const float* h_neg, int rows, int n_vis, int n_hid, int ldv, int ldh,
                        void* workspace, size_t workspace_bytes, kurbm_stream_t stream);

/* ---- bf16 storage / fp32 accumulate (extension; BASELINE.json config 5) --------------------
 *
 * The fp32 parameters of kurbm_params stay authoritative.  A MIRROR buffer (caller-owned device
 * memory, kurbm_bf16_mirror_bytes) holds their bf16 images in both orientations; it must be
 * refreshed (kurbm_bf16_mirror_refresh) whenever the caller writes the fp32 weights, and is kept in
 * sync by kurbm_cd_step_bf16 itself.  Inputs and outputs of these entry points are fp32 exactly as in
 * the fp32 API; only the matrix products run on bf16 operands (weights, data and probabilities
 * rounded to nearest even; 0/1 samples are exact), accumulated in fp32.
 */
size_t kurbm_bf16_mirror_bytes(kurbm_ctx* ctx, int n_vis, int n_hid);
int kurbm_bf16_mirror_refresh(kurbm_ctx* ctx, const kurbm_params* p, void* mirror, size_t mirror_bytes,
                              kurbm_stream_t stream);
size_t kurbm_bf16_workspace_bytes(kurbm_ctx* ctx, int rows, int n_vis, int n_hid, int k);

/* Same contract as kurbm_cd_step, products in bf16. */
int kurbm_cd_step_bf16(kurbm_ctx* ctx, const kurbm_params* p, void* mirror, size_t mirror_bytes,
                       const float* v_batch, int rows, int ldv, const kurbm_cd_opts* opts, int which,
                       void* workspace, size_t workspace_bytes, kurbm_stream_t stream);

/* One half step with bf16 products (test hook): dir 0 = v->h, 1 = h->v; fp32 in, fp32 planes out. */
int kurbm_half_step_bf16(kurbm_ctx* ctx, const kurbm_params* p, void* mirror, size_t mirror_bytes, int dir,
                         const float* in, int rows, int ld_in, int act, int noise, const kurbm_rng* rng,
                         float* out_sample, float* out_prob, float* out_u, int ld_out,
                         void* workspace, size_t workspace_bytes, kurbm_stream_t stream);

/* ---- x3: fp32 values as exact bf16 triples on the bf16 matrix cores (extension) ---------------
 *
 * gfx950 multiplies bf16 operands 16x faster than fp32 ones (v_mfma_f32_16x16x32_bf16 vs
 * v_mfma_f32_16x16x4_f32), and every fp32 value is EXACTLY the sum of three bf16 values
 * (x = hi + mid + lo, round-to-nearest at each stage).  The x3 entry points keep fp32 storage and
 * fp32 accumulation, but carry each real-valued GEMM operand as its three pieces and run one
 * k-segment of the product per pair of pieces; 0/1 samples (h_pos, v_neg, the hidden and visible
 * states of a Bernoulli chain) are a single exact piece.  Products of pieces are exact in fp32, so the
 * result differs from an fp32 GEMM only in the order of the fp32 additions.  Same reference
 * operations as the fp32 entry points (rbm.py:46-47, :121-134), both visible modes: with
 * MODE_VISIBLE_GAUSSIAN (rbm.py:55-67, :139-159) the negative visibles are real-valued and travel as three
 * pieces too (row-major for the next half step, transposed for the statistics).
 *
 * v_pieces: 1 promises that every element of v_batch (and of opts->v_chain) is exactly a bf16
 * value, which 0/1 data is (check with kurbm_bf16_exact); 3 splits the batch as well.  When both
 * operands of a product are split, the three pairs of pieces below 2^-24 of the product are left out.
 * 1 | KURBM_V_BINARY additionally promises that every element of v_batch (and of opts->v_chain) is 0.0 or 1.0 (bit 1 of
 * kurbm_bf16_exact's flag clear).  The library then keeps the data planes, like those of its own 0/1 samples, as bytes
 * instead of bf16 (half the traffic of the A operand of a half step), and runs the positive statistics v_pos^T h_pos, a
 * 0/1 x 0/1 product, on the fp8 matrix cores (twice the bf16 rate).  Both are exact: the operands are 0 or 1, the
 * accumulation is fp32.  A window of resident planes (kurbm_x3_convert_rows) must be made and used with the same
 * v_pieces value.
 * The mirror holds the pieces of W in both orientations and follows the rules of the bf16 mirror.
 */
#define KURBM_V_BINARY 0x10
size_t kurbm_x3_mirror_bytes(kurbm_ctx* ctx, int n_vis, int n_hid);
int kurbm_x3_mirror_refresh(kurbm_ctx* ctx, const kurbm_params* p, void* mirror, size_t mirror_bytes,
                            kurbm_stream_t stream);
size_t kurbm_x3_workspace_bytes(kurbm_ctx* ctx, int rows, int n_vis, int n_hid, int k, int v_pieces);

/* Same contract as kurbm_cd_step. */
int kurbm_cd_step_x3(kurbm_ctx* ctx, const kurbm_params* p, void* mirror, size_t mirror_bytes,
                     const float* v_batch, int v_pieces, int rows, int ldv, const kurbm_cd_opts* opts,
                     int which, void* workspace, size_t workspace_bytes, kurbm_stream_t stream);

/*
 * Measurement hook: ONE launch of the kurbm_cd_step_x3 sequence, on the planes a previous complete
 * step left in `workspace` (same arguments as that step).  stage: 0 v_pos -> bf16 pieces, 1 h_pos half
 * step, 2 v_neg half step(s), 3 h_neg half step, 4 statistics GEMM, 5 slab reduction + parameter
 * update + mirror refresh, 6 mirror refresh alone.  bench.py times the kernels of a step with it.
 */
int kurbm_cd_step_x3_stage(kurbm_ctx* ctx, const kurbm_params* p, void* mirror, size_t mirror_bytes,
                           const float* v_batch, int v_pieces, int rows, int ldv, const kurbm_cd_opts* opts,
                           int which, int stage, void* workspace, size_t workspace_bytes,
                           kurbm_stream_t stream);

/* Test hook: the planes the last complete kurbm_cd_step_x3 (same rows, v_pieces, opts->mode; CD-1, no persistent chain) left
 * in `workspace`, decoded to fp32 [rows][units] -- bytes (0x40 = one, k-permuted), fp8 (0x38 = one), bf16, or the sum of
 * three bf16 pieces; transposed planes are transposed back, the negated h_neg plane is un-negated.  The tests compare them
 * with the states the half-step hooks return for the same counters, i.e. with the oracle. */
enum { KURBM_PLANE_H_POS = 0, KURBM_PLANE_H_POS_T = 1, KURBM_PLANE_V_NEG = 2, KURBM_PLANE_V_NEG_T = 3, KURBM_PLANE_H_NEG_T = 4 };
int kurbm_x3_dump_plane(kurbm_ctx* ctx, int which, int rows, int n_vis, int n_hid, int v_pieces, int mode,
                        const void* workspace, size_t workspace_bytes, float* out, int ld_out, kurbm_stream_t stream);

/* The per-step score of fit(verbose = 1), rbm.py:225-233, on the x3 path in ONE call with nothing returned to the host:
 * fe = F(v_batch); v' = a one-step reconstruction of v_batch (fresh chain: opts->chain, sites 0 and 1, opts->step, opts->seed,
 * opts->row0, opts->mode; opts->v_planes as for a step); fe' = F(v'); *score = mean |fe - fe'| (device float, 16-byte
 * aligned); F (nullable, device [2 * rows]) receives fe then fe'.  Workspace as for kurbm_cd_step_x3 on the same rows; the
 * caller reads *score back when it wants to print it (the reference prints every step: rbm.py:234). */
int kurbm_score_x3(kurbm_ctx* ctx, const kurbm_params* p, void* mirror, size_t mirror_bytes, const float* v_batch,
                   int v_pieces, int rows, int ldv, const kurbm_cd_opts* opts, float* score, float* F, void* workspace,
                   size_t workspace_bytes, kurbm_stream_t stream);

/* kurbm_free_energy on the x3 path (rbm.py:73-76): workspace >= the bf16 pieces of v + one float per row and
 * 128-column tile (kurbm_x3_workspace_bytes of the same rows is enough). */
int kurbm_free_energy_x3(kurbm_ctx* ctx, const kurbm_params* p, void* mirror, size_t mirror_bytes, const float* v,
                         int v_pieces, int rows, int ldv, float* F, void* workspace, size_t workspace_bytes,
                         kurbm_stream_t stream);

/* kurbm_cd_epoch on the x3 path: every batch of an epoch in one call (fused updates; returns the number of steps).
 * All work is ordered on `stream`.  With opts->v_planes the planes of batch t are read at v_planes + t * v_planes_stride. */
int kurbm_cd_epoch_x3(kurbm_ctx* ctx, const kurbm_params* p, void* mirror, size_t mirror_bytes, const float* V,
                      int v_pieces, int n_rows, int ldv, int batch_size, const kurbm_cd_opts* opts,
                      void* workspace, size_t workspace_bytes, kurbm_stream_t stream);

/*
 * The data-parallel x3 step in pieces, so that the all-reduce of the first rows of dW can run while the rest is
 * still being computed:  kurbm_cd_chain_x3 = the chain alone (v_pos -> bf16, the three half steps; leaves the
 * states in `workspace`);  kurbm_x3_stats_rows = the statistics of visible rows [m_lo, m_hi) of that chain into
 * opts->delta_out (packed [dW | db_h | db_v]; rows m_lo .. m_hi-1 of dW, and with m_hi == n_vis also db_h and
 * db_v).  m_lo a multiple of 128; opts->apply = 0.  Calling it for [0, m) and [m, n_vis) gives the delta of
 * kurbm_cd_step_x3(apply = 0) up to the order of the fp32 additions (the split-K slicing differs).
 */
int kurbm_cd_chain_x3(kurbm_ctx* ctx, const kurbm_params* p, void* mirror, size_t mirror_bytes, const float* v_batch,
                      int v_pieces, int rows, int ldv, const kurbm_cd_opts* opts, void* workspace,
                      size_t workspace_bytes, kurbm_stream_t stream);
int kurbm_x3_stats_rows(kurbm_ctx* ctx, const kurbm_params* p, void* mirror, size_t mirror_bytes,
                        const float* v_batch, int v_pieces, int rows, int ldv, const kurbm_cd_opts* opts, int m_lo,
                        int m_hi, void* workspace, size_t workspace_bytes, kurbm_stream_t stream);

/* kurbm_apply_delta for the x3 path (data-parallel step): W, b_h, b_v += lr * delta AND the mirror's weight
 * pieces rewritten, in one launch. */
int kurbm_x3_apply_delta(kurbm_ctx* ctx, const kurbm_params* p, void* mirror, size_t mirror_bytes,
                         const float* delta, float lr, int which, kurbm_stream_t stream);

/*
 * Resident data planes.  RBM.fit slices the same matrix V every epoch (rbm.py:113, :211), so its bf16 images -- the pieces
 * of the rows row-major and transposed, and their column sums per 64-row band -- are made ONCE per window of rows and a
 * step given opts->v_planes reads them instead of converting the window again.  `planes` is caller-owned device memory of
 * kurbm_x3_planes_bytes(rows) bytes per window; its layout follows `rows` (a remainder batch has its own) and v_pieces:
 * with 1 | KURBM_V_BINARY the row-major plane holds bytes and the transposed one fp8 (see above), so the steps that read a
 * window must pass the v_pieces it was made with, under the same KURBM_X3_BYTES / KURBM_X3_F8POS settings of the context.
 */
size_t kurbm_x3_planes_bytes(kurbm_ctx* ctx, int rows, int n_vis, int v_pieces);
int kurbm_x3_convert_rows(kurbm_ctx* ctx, const float* v, int rows, int ldv, int n_vis, int v_pieces, void* planes,
                          size_t planes_bytes, kurbm_stream_t stream);

/* One half step on the x3 path (transform / test hook): dir 0 = v->h, 1 = h->v. */
int kurbm_half_step_x3(kurbm_ctx* ctx, const kurbm_params* p, void* mirror, size_t mirror_bytes, int dir,
                       const float* in, int in_pieces, int rows, int ld_in, int act, int noise,
                       const kurbm_rng* rng, float* out_sample, float* out_prob, float* out_u, int ld_out,
                       void* workspace, size_t workspace_bytes, kurbm_stream_t stream);

/* ---- data parallel: the exchange step ------------------------------------
 *
 * The reference has no multi-device path.  Each GPU runs the chain on its rows of the batch (rbm.py:119-124: rows
 * never interact) and the updates are SUMS over the batch (rbm.py:125-134), so ONE sum all-reduce of the packed
 * [dW | db_h | db_v] buffer reproduces the single-GPU update up to the order of the fp32 additions.  The collective
 * is RCCL's ncclAllReduce over xGMI, called from inside this library (bound at run time: librccl.so.1, or the file
 * KURBM_RCCL_LIB names); the host language only carries the 128-byte unique id from rank 0 to the other ranks.
 * One process per GPU: kurbm_comm_unique_id on rank 0, kurbm_comm_init_rank everywhere (collective: returns when all
 * ranks have joined).  One process driving several GPUs: kurbm_comm_init_all (ncclCommInitAll), `out` an array of ndev.
 */
#define KURBM_UNIQUE_ID_BYTES 128
int kurbm_comm_unique_id(void* id, size_t id_bytes);                                    /* ncclGetUniqueId  */
int kurbm_comm_init_rank(int device, int nranks, int rank, const void* id, size_t id_bytes,
                         kurbm_comm** out);                                             /* ncclCommInitRank */
int kurbm_comm_init_all(int ndev, const int* devs, kurbm_comm** out);                   /* ncclCommInitAll  */
int kurbm_comm_count(const kurbm_comm* comm);   /* ranks as RCCL reports them (ncclCommCount); < 0 = error */
int kurbm_comm_rank(const kurbm_comm* comm);    /* ncclCommUserRank; < 0 = error */
void kurbm_comm_destroy(kurbm_comm* comm);
/* In-place sum over all ranks of buf[0 .. n) (device fp32), ordered on `stream`. */
int kurbm_allreduce_sum_f32(kurbm_comm* comm, float* buf, size_t n, kurbm_stream_t stream);

/*
 * The data-parallel x3 step in ONE call: this rank's chain on its `rows` rows (opts->row0 = their global index), the
 * statistics GEMM in n_chunks row ranges of dW, each range's packed sums all-reduced on the library's comm stream
 * while the next range is still being computed on `stream` (db_h, db_v travel with the last range), then -- with
 * opts->apply = 1 -- W, b_h, b_v += lr * (summed delta) and the weight-piece mirror rewritten, in one launch, once the
 * last all-reduce has landed.  opts->delta_out (packed, required) holds the SUMMED statistics afterwards.  rows = 0
 * is legal (a rank that owns no rows of a remainder batch contributes zeros; v_batch is not read).  n_chunks <= 0:
 * ONE range, all-reduced on `stream` itself, whatever the size of the exchange (on MI355X / ROCm 7 the hand-off between two
 * streams costs more than a 3.2 MB all-reduce can hide, and the several-range schedule has not run at a world size above 1
 * yet) unless the context knob KURBM_DP_CHUNKS names a count -- DESIGN.md section 5.  Several ranges are an OPT-IN
 * (n_chunks > 1 or KURBM_DP_CHUNKS > 1): each one is then APPLIED (its rows of W, its part of the mirror) on the comm stream
 * as soon as its all-reduce has landed, under the statistics GEMM of the next; a HIP failure in that loop still joins the
 * comm stream back to `stream` before it is reported.  Every rank must pass the same n_chunks.  All work is ordered on
 * `stream` from the caller's point of view.
 * Every argument is checked before the first collective is enqueued, so a refused call has not joined an all-reduce
 * (the checks depend on arguments all ranks share; `rows` may differ).
 */
int kurbm_cd_step_x3_dp(kurbm_ctx* ctx, kurbm_comm* comm, const kurbm_params* p, void* mirror, size_t mirror_bytes,
                        const float* v_batch, int v_pieces, int rows, int ldv, const kurbm_cd_opts* opts, int n_chunks,
                        void* workspace, size_t workspace_bytes, kurbm_stream_t stream);
/* The same step on the rounded-bf16 path (BASELINE.json config 5: 4096 x 4096, 67 MB of packed sums per step). */
int kurbm_cd_step_bf16_dp(kurbm_ctx* ctx, kurbm_comm* comm, const kurbm_params* p, void* mirror, size_t mirror_bytes,
                          const float* v_batch, int rows, int ldv, const kurbm_cd_opts* opts, int n_chunks, void* workspace,
                          size_t workspace_bytes, kurbm_stream_t stream);

/* ---- small RBMs: one launch per step -----------------------------------------------------------------------------
 * The whole fused CD-1 update (rbm.py:120-134) of a SMALL problem -- the reference example's own 784 -> 128 at batch 128,
 * BASELINE.json configs[0] -- in ONE grid-resident launch (phases separated by device-scope barriers; csrc/kurbm_small.hip):
 * at these sizes the five launches of kurbm_cd_step are five launch latencies for a few microseconds of arithmetic.  Same
 * arguments, workspace (kurbm_workspace_bytes) and Philox counters as kurbm_cd_step; CD-1 from the data only (opts->k = 1,
 * no v_chain), applied in place (opts->apply = 1, no delta_out; `which` honoured).  Results agree with kurbm_cd_step to fp32
 * rounding (another summation order), not bit for bit.  The grid must be resident (at most one workgroup per CU, an otherwise
 * idle device); a barrier that times out sets bit 2 of kurbm_ctx_status and the update is skipped.  Where the context's probe
 * found the dispatcher dealing consecutive workgroups to consecutive XCDs (MI355X in SPX mode), phases 1-3 of a 16-row band of
 * the batch run inside ONE XCD -- the workgroups that find themselves on it -- and hand their planes over through its L2
 * (KURBM_SMALL_LOCAL, default 1; the launch is then always the whole device).  kurbm_cd_epoch_small: every batch of an epoch
 * in one call (returns the number of steps). */
int kurbm_cd_step_small(kurbm_ctx* ctx, const kurbm_params* p, const float* v_batch, int rows, int ldv,
                        const kurbm_cd_opts* opts, int which, void* workspace, size_t workspace_bytes, kurbm_stream_t stream);
int kurbm_cd_epoch_small(kurbm_ctx* ctx, const kurbm_params* p, const float* V, int n_rows, int ldv, int batch_size,
                         const kurbm_cd_opts* opts, void* workspace, size_t workspace_bytes, kurbm_stream_t stream);
/* ... and the per-step score of fit(verbose = 1), rbm.py:225-233, the same way: *score = mean |F(v) - F(v')| over the batch,
 * v' a fresh one-step reconstruction (opts->chain, sites 0 and 1, opts->seed / row0 / step / mode), in ONE launch with nothing
 * returned to the host; F (nullable, device [2 * rows]) receives F(v) then F(v').  At most 512 rows.  Workspace as for
 * kurbm_cd_step (kurbm_workspace_bytes).  Replaces the reference's four graph executions of its score (two free energies,
 * two sampling half steps). */
int kurbm_score_small(kurbm_ctx* ctx, const kurbm_params* p, const float* v_batch, int rows, int ldv, const kurbm_cd_opts* opts,
                      float* score, float* F, void* workspace, size_t workspace_bytes, kurbm_stream_t stream);
/* `score` is 8-byte aligned and receives TWO floats in one system-scope store: the score, then 1.0f -- so it may be pinned host
 * memory that the caller polls for the second word instead of synchronising with the device.  kurbm_cd_epoch_small_scored is
 * the batch loop of fit(verbose = 1) as one call: the update of every batch followed by its score (opts->step counts up for
 * both; the score draws from chain `score_chain`), scores[2 i], scores[2 i + 1] = the score of step i and its 1.0f.  Returns the
 * number of steps. */
int kurbm_cd_epoch_small_scored(kurbm_ctx* ctx, const kurbm_params* p, const float* V, int n_rows, int ldv, int batch_size,
                                const kurbm_cd_opts* opts, int score_chain, float* scores, void* workspace, size_t workspace_bytes,
                                kurbm_stream_t stream);

/* ---- data parallel, plan B: the exchange through peer pointers, no collective library ----------------------------
 *
 * A two-shot all-reduce over hipIpc-mapped buffers, its second shot fused into the launch that applies the update
 * (keras_unsupervised_amd/csrc/kurbm_peer.hip has the protocol).  Every rank: kurbm_peer_create (allocates and owns the
 * rank's exchange buffer for an n_vis x n_hid RBM), kurbm_peer_handle (its hipIpc handle, kurbm_peer_handle_bytes() bytes --
 * the host language carries the handles between the ranks, as it carries RCCL's unique id), kurbm_peer_connect with ALL
 * ranks' handles in rank order (maps the peers' buffers).  Then kurbm_cd_step_x3_peer in lock step on all ranks: the same
 * step as kurbm_cd_step_x3_dp with opts->apply = 1 (opts->delta_out is ignored: the sums live in the exchange buffers), sums
 * added in rank order, so every replica ends with identical bits.  n_hid % 4 == 0; at most 8 ranks.  A rank that fails an
 * argument check returns before it has published anything.  Every device-side wait is bounded (environment
 * KURBM_PEER_TIMEOUT_MS, default 10 000): a peer that never arrives sets bit 1 of kurbm_ctx_status and the update is skipped.
 * kurbm_peer_allreduce_sum_f32: the plain in-place sum of n <= n_vis n_hid + n_hid + n_vis floats (16-byte aligned), same protocol.
 * RCCL (kurbm_comm_*) stays the default exchange; this one also runs between PROCESSES THAT SHARE ONE GPU, which RCCL refuses.
 */
typedef struct kurbm_peer kurbm_peer;
size_t kurbm_peer_handle_bytes(void);
int kurbm_peer_create(int device, int nranks, int rank, int n_vis, int n_hid, kurbm_peer** out);
int kurbm_peer_handle(kurbm_peer* peer, void* handle, size_t handle_bytes);
int kurbm_peer_connect(kurbm_peer* peer, const void* handles, size_t bytes);
int kurbm_peer_ranks(const kurbm_peer* peer);
int kurbm_peer_allreduce_sum_f32(kurbm_ctx* ctx, kurbm_peer* peer, float* buf, size_t n, kurbm_stream_t stream);
int kurbm_cd_step_x3_peer(kurbm_ctx* ctx, kurbm_peer* peer, const kurbm_params* p, void* mirror, size_t mirror_bytes,
                          const float* v_batch, int v_pieces, int rows, int ldv, const kurbm_cd_opts* opts, void* workspace,
                          size_t workspace_bytes, kurbm_stream_t stream);
void kurbm_peer_destroy(kurbm_peer* peer);

/* *flag (device int, zeroed by the caller) |= 1 if some element of x [rows][ld] is not exactly representable in bf16,
 * |= 2 if some element is neither 0.0 nor 1.0.  0: v_pieces = 1 | KURBM_V_BINARY; 2: v_pieces = 1; else 3. */
int kurbm_bf16_exact(kurbm_ctx* ctx, const float* x, int rows, int cols, int ld, int* flag,
                     kurbm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* KURBM_H */
