// kurbm_api.hip -- the extern "C" boundary declared in include/kurbm.h.
//
// Host-side planning (tile configuration, split-K factor, workspace carving) and the launch
// sequences; all arithmetic is in kurbm_kernels.hip.  No allocation, no host synchronisation.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/kurbm.h"
#include "kurbm_kernels.h"
#include "kurbm_comm.h"

using namespace kurbm;

// Experiment knobs.  Every one is read from the environment ONCE, at kurbm_ctx_create, into the context (no getenv on
// the launch path); kurbm_ctx_set_option changes one on a live context (tests and tuning sweeps).  KN_AUTO = "let the
// planner decide".
enum { KN_LDPAD, KN_X3_F8POS, KN_X3_BYTES, KN_X3_STATS_TALL, KN_BF16_SPLIT, KN_X3_FULL, KN_X3_TALL, KN_X3_MFAST, KN_X3_STATS_MFAST,
       KN_UNFUSED_MIRROR, KN_X3_XCD2D, KN_REDUCE_TR, KN_DP_CHUNKS, KN_X3_STATS_BYTES, KN_MAP_SLOW, KN_X3_PAIR, KN_X3_SPLIT_STATS, KN_X3_ATR, KN_SMALL_LOCAL,
       KN_COUNT };
constexpr int KN_AUTO = -1;
static const struct { const char* env; int dflt; } KNOBS[KN_COUNT] = {
    {"KURBM_LDPAD", 0},            // extra elements per bf16 plane row (L2 channel camping probe: no effect)
    {"KURBM_X3_F8POS", 1},         // 0: the positive statistics of 0/1 data stay on bf16 planes
    {"KURBM_X3_BYTES", 1},         // 0: the row-major planes of 0/1 samples / data stay bf16 (1: bytes, half the A tiles)
    {"KURBM_X3_STATS_TALL", 1},    // 0: 128 x 128 tiles for the x3 statistics GEMM (1: 256 x 64 where n_vis > 128)
    {"KURBM_BF16_SPLIT", KN_AUTO}, // split-K slices of the bf16 / x3 statistics GEMM
    {"KURBM_X3_FULL", 0},          // 1: all nine piece pairs of a real x real product
    {"KURBM_X3_TALL", KN_AUTO},    // 0 / 1: never / always 256 x 64 half-step tiles
    {"KURBM_X3_MFAST", 1},         // block order of the x3 half steps
    {"KURBM_X3_STATS_MFAST", 0},   // block order of the x3 statistics GEMM
    {"KURBM_UNFUSED_MIRROR", 0},   // 1: slab reduce and weight-piece mirror as two launches
    {"KURBM_X3_XCD2D", 1},         // 0: linear block order of k_gemm_pb instead of one 2-D block of tiles per XCD
    {"KURBM_REDUCE_TR", 0},        // tile height (16 / 32 / 64) of the slab-reduce + mirror launch; 0: by the grid it makes
    {"KURBM_DP_CHUNKS", 0},        // row ranges of dW in the data-parallel step when the caller passes n_chunks <= 0; 0: by message size
    {"KURBM_X3_STATS_BYTES", 1},   // 0: v_neg^T reaches the statistics GEMM as a bf16 plane (1: as bytes where the positive half is fp8)
    {"KURBM_MAP_SLOW", 0},         // 1: k_gemm_pb maps its blocks by integer division (the path of grids too large for the multiply-high
                                   //    constants: tests)
    {"KURBM_X3_PAIR", 1},          // 0: a real-valued A operand walks its three segments one after the other (256 x 64 tiles, the generic walk);
                                   //    1: two tiles per k position on 128 x 128 tiles (k_gemm_pb, "BSP")
    {"KURBM_X3_SPLIT_STATS", 1},   // 0: the statistics of real-valued data in ONE launch (round 3: three one-piece positive tiles per k position);
                                   //    1: two launches -- the positive half as the transposed problem h_pos^T (bytes) x the pieces of v_pos^T, the
                                   //    negative half on byte planes (Bernoulli visibles) or on the paired walk (Gaussian visibles)
    {"KURBM_X3_ATR", 1},           // 0: Gaussian visibles leave the h -> v half step as pieces in BOTH orientations; 1: row-major only, and the
                                   //    negative statistics read them through transposed LDS reads (with KURBM_X3_SPLIT_STATS)
    {"KURBM_SMALL_LOCAL", 1},      // 1: the one-launch small step hands h_pos / v_neg between the workgroups of ONE XCD through its L2 (kurbm_small.hip;
                                   //    where the context's probe found the dispatcher dealing consecutive workgroups to consecutive XCDs), 0: every phase over the whole grid
};

constexpr size_t STATUS_BYTES = 4096;
struct kurbm_ctx {
    int device;
    int ncu;
    // tuning overrides (environment, read once at ctx creation): -1 = automatic
    int force_cfg[3];   // per layout: KURBM_CFG_VH / KURBM_CFG_HV / KURBM_CFG_OUTER
    int force_split;    // KURBM_SPLIT
    int tile_major;     // KURBM_TILE_MAJOR (default 1): k-slices of a statistics tile share an XCD
    int knob[KN_COUNT];
    int xcc_round_robin;      // 1: a whole-device grid puts one workgroup of every octet (b / 8) on each of eight XCDs (probed at creation)
    unsigned xcc_map;         // the XCDs of workgroups 0 .. 7 of the probe launch, a nibble each (diagnostic)
    unsigned* status;   // device word, sticky: kurbm_ctx_status, in front of the only device memory the library owns (STATUS_BYTES: behind
                        // the status word the grid barrier of kurbm_cd_step_small -- words 64 .. 223: eight per-XCD arrival counters, the
                        // grid's counter, the generation word, one 64-byte line each)
};

static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

static thread_local std::string g_err;

static int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

namespace kurbm {
unsigned* ctx_status_word(kurbm_ctx* ctx) { return ctx ? ctx->status : nullptr; }
int fail_msg(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
}  // namespace kurbm

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return fail(KURBM_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

static bool bad_matrix(const void* p, int ld, int cols) { return !p || !aligned16(p) || ld % 4 != 0 || ld < cols; }

static int check_params(const kurbm_params* p) {
    if (!p) return fail(KURBM_ERR_ARG, "params is null");
    if (p->n_vis <= 0 || p->n_hid <= 0) return fail(KURBM_ERR_ARG, "n_vis/n_hid must be positive");
    if (bad_matrix(p->W, p->ldw, p->n_hid)) return fail(KURBM_ERR_ARG, "W: null, misaligned, or ldw %% 4 != 0 / ldw < n_hid");
    if (!p->b_h || !p->b_v) return fail(KURBM_ERR_ARG, "bias pointer is null");
    return KURBM_OK;
}

static RngArgs make_rng(uint64_t seed, uint64_t row0, uint32_t stream_id, uint32_t step) {
    RngArgs r;
    r.seed_lo = (uint32_t)(seed & 0xFFFFFFFFu);
    r.seed_hi = (uint32_t)(seed >> 32);
    r.stream_id = stream_id;
    r.step = step;
    r.row0 = row0;
    return r;
}

// ---- planning -------------------------------------------------------------------------
// Tile choice for an M x N output.  Model, fitted to MI355X measurements of this kernel family
// (profiles/): time ~ padded MFMA work x CU imbalance / efficiency, where a CU that holds
// two or more workgroups (their barriers and epilogues interleave) runs ~0.70 MFMA-busy and a CU
// with a single workgroup ~0.62.  Ties go to the larger tile (less L2 -> LDS traffic).
static int pick_cfg(int ncu, int M, int N, int* gm_out, int* gn_out, int force = -1) {
    int best = 0;
    double best_cost = -1.0;
    long long best_blocks = 0;
    for (int c = 0; c < CFG_COUNT; ++c) {
        if (force >= 0 && force < CFG_COUNT && c != force) continue;
        int bm, bn;
        tile_shape(c, &bm, &bn);
        const long long blocks = (long long)ceil_div(M, bm) * ceil_div(N, bn);
        const long long per_cu = (blocks + ncu - 1) / ncu;                 // what the busiest CU runs
        const double eff = (blocks >= 2LL * ncu) ? 0.70 : (blocks > ncu ? 0.66 : 0.62);
        const double cost = (double)per_cu * bm * bn / eff;
        if (best_cost < 0 || cost < best_cost * 0.999 || (cost <= best_cost * 1.001 && blocks < best_blocks)) {
            best = c; best_cost = cost; best_blocks = blocks;
        }
    }
    int bm, bn;
    tile_shape(best, &bm, &bn);
    *gm_out = ceil_div(M, bm);
    *gn_out = ceil_div(N, bn);
    return best;
}

struct OuterPlan { int cfg, gm, gn, nkt, kt_total, nsplit, nsplit_bound, kt_per_split, ld_slab; };

// statistics GEMM: output n_vis x n_hid, k = batch rows (two signed segments)
static OuterPlan plan_outer(const kurbm_ctx* ctx, int rows, int n_vis, int n_hid) {
    const int ncu = ctx->ncu;
    OuterPlan pl;
    // tile by least padded area (independent of rows so the slab count is monotone in rows)
    long long best = -1;
    pl.cfg = 0; pl.gm = pl.gn = 1;
    const int fc = ctx->force_cfg[LAYOUT_OUTER];
    for (int c = (fc >= 0 ? fc : 0); c < (fc >= 0 ? fc + 1 : 3); ++c) {
        int bm, bn;
        tile_shape(c, &bm, &bn);
        const int gm = ceil_div(n_vis, bm), gn = ceil_div(n_hid, bn);
        const long long area = (long long)gm * gn * bm * bn;
        if (best < 0 || area < best) { best = area; pl.cfg = c; pl.gm = gm; pl.gn = gn; }
    }
    pl.nkt = rows / 32;          // full k-tiles per segment; a rows % 32 tail goes to the last slice
    pl.kt_total = 2 * pl.nkt;
    const int tiles = pl.gm * pl.gn;
    // two workgroups per CU: their barriers / slab stores interleave (measured 8-10 % over one per CU)
    int s = ctx->force_split > 0 ? ctx->force_split : (2 * ncu) / tiles;
    if (s < 1) s = 1;
    if (s > pl.kt_total) s = pl.kt_total;
    if (s < 1) s = 1;
    pl.nsplit_bound = s;  // monotone in rows: what the workspace reserves
    pl.kt_per_split = pl.kt_total > 0 ? ceil_div(pl.kt_total, s) : 0;
    pl.nsplit = pl.kt_total > 0 ? ceil_div(pl.kt_total, pl.kt_per_split) : 1;
    pl.ld_slab = round_up(n_hid, 4);
    return pl;
}

struct Workspace {
    float *h_pos, *h_neg, *h_tmp, *v_neg, *part_h, *part_v, *slab;
    float *small_t;   // kurbm_cd_step_small's transposed planes (rows <= SMALL_ROWS_MAX only)
    int ldh, ldv, ld_part_h, ld_part_v, max_row_tiles;
    size_t slab_stride, bytes;
};

static size_t align_up(size_t x) { return (x + 255) / 256 * 256; }
constexpr int SMALL_ROWS_MAX = 512;   // batches up to here carry the transposed planes of the one-launch step in their workspace

static Workspace carve(const kurbm_ctx* ctx, void* base, int rows, int n_vis, int n_hid, int k) {
    Workspace w;
    w.ldh = round_up(n_hid, 4);
    w.ldv = round_up(n_vis, 4);
    w.max_row_tiles = ceil_div(rows, 64);
    w.ld_part_h = w.ldh;
    w.ld_part_v = w.ldv;
    const OuterPlan pl = plan_outer(ctx, rows, n_vis, n_hid);
    w.slab_stride = (size_t)n_vis * pl.ld_slab;
    size_t off = 0;
    char* b = static_cast<char*>(base);
    auto take = [&](size_t nfloat) { float* p = reinterpret_cast<float*>(b + off); off = align_up(off + nfloat * 4); return p; };
    w.h_pos = take((size_t)rows * w.ldh);
    w.h_neg = take((size_t)rows * w.ldh);
    w.h_tmp = take((size_t)rows * w.ldh);   // intermediate h_t (k > 1) / persistent-chain start
    w.v_neg = take((size_t)rows * w.ldv);
    w.part_h = take((size_t)w.max_row_tiles * w.ld_part_h);
    w.part_v = take((size_t)w.max_row_tiles * w.ld_part_v);
    w.slab = take(w.slab_stride * pl.nsplit_bound);
    // (the one-launch score keeps its row partials there: 2 ceil(n_hid / 16) + ceil(n_vis / 16) + 1 rows -- the + 4 covers one-unit layers)
    w.small_t = rows <= SMALL_ROWS_MAX ? take((size_t)(2 * n_hid + n_vis + 4) * round_up(rows, 16)) : nullptr;
    (void)k;
    // free energy: row partials [col tiles][round_up(rows,4)] alias the front of the workspace
    const size_t fe = align_up((size_t)ceil_div(n_hid, 64) * round_up(rows, 4) * 4);
    w.bytes = off > fe ? off : fe;
    return w;
}

// ---- the half step ---------------------------------------------------------------------
static int half_step(kurbm_ctx* ctx, int layout, const kurbm_params* p, const float* in, int rows, int ld_in,
                     int act, int noise, const RngArgs* rng, float* out_sample, float* out_prob, float* out_u,
                     int ldo, const float* ref, int ldref, float* colpart, int ld_colpart, int* grid_m_out,
                     hipStream_t st) {
    const int K = (layout == LAYOUT_VH) ? p->n_vis : p->n_hid;
    const int N = (layout == LAYOUT_VH) ? p->n_hid : p->n_vis;
    if (rows <= 0) return fail(KURBM_ERR_ARG, "rows must be positive");
    if (bad_matrix(in, ld_in, K)) return fail(KURBM_ERR_ARG, "input: null, misaligned, ld %% 4 != 0 or ld < columns");
    if (act < ACT_SIGMOID || act > ACT_LINEAR) return fail(KURBM_ERR_ARG, "unknown activation %d", act);
    if (noise < NOISE_NONE || noise > NOISE_GAUSSIAN) return fail(KURBM_ERR_ARG, "unknown noise %d", noise);
    if (!out_sample && !out_prob) return fail(KURBM_ERR_ARG, "both outputs are null");
    if (noise != NOISE_NONE && !rng) return fail(KURBM_ERR_ARG, "rng is null");
    if (noise != NOISE_NONE && (rng->row0 & 3)) return fail(KURBM_ERR_ARG, "rng.row0 must be a multiple of 4");
    if (noise == NOISE_NONE && out_sample && !out_prob) { out_prob = out_sample; out_sample = nullptr; }
    if ((out_sample && bad_matrix(out_sample, ldo, N)) || (out_prob && bad_matrix(out_prob, ldo, N)))
        return fail(KURBM_ERR_ARG, "output: misaligned, ld %% 4 != 0 or ld < columns");

    GemmArgs g;
    memset(&g, 0, sizeof g);
    g.A0 = in; g.lda = ld_in;
    g.B0 = p->W; g.ldb = p->ldw;
    g.M = rows; g.N = N; g.K = K;
    g.nseg = 1;
    g.nkt = K / 32;
    g.kt_total = g.nkt; g.kt_per_split = g.nkt; g.nsplit = 1;
    const int cfg = pick_cfg(ctx->ncu, rows, N, &g.grid_m, &g.grid_n, ctx->force_cfg[layout]);
    g.bias = (layout == LAYOUT_VH) ? p->b_h : p->b_v;
    g.out_sample = out_sample; g.out_prob = out_prob; g.out_u = out_u; g.ldo = ldo;
    g.ref = ref; g.ldref = ldref; g.colpart = colpart; g.ld_colpart = ld_colpart;
    g.act = act; g.noise = noise;
    if (rng) g.rng = *rng;
    g.m_fastest = (g.grid_m < g.grid_n) ? 1 : 0;
    if (grid_m_out) *grid_m_out = g.grid_m;
    HIP_TRY(launch_gemm(layout, cfg, EPI_HALFSTEP, g, st));
    return KURBM_OK;
}

static int outer_slabs(kurbm_ctx* ctx, const float* v_pos, const float* h_pos, const float* v_neg, const float* h_neg,
                       int rows, int n_vis, int n_hid, int ldv, int ldh, float* slab, size_t slab_stride,
                       const OuterPlan& pl, hipStream_t st) {
    GemmArgs g;
    memset(&g, 0, sizeof g);
    g.A0 = v_pos; g.A1 = v_neg; g.lda = ldv;
    g.B0 = h_pos; g.B1 = h_neg; g.ldb = ldh;
    g.M = n_vis; g.N = n_hid; g.K = rows;
    g.nseg = 2;
    g.nkt = pl.nkt; g.kt_total = pl.kt_total; g.kt_per_split = pl.kt_per_split; g.nsplit = pl.nsplit;
    g.grid_m = pl.gm; g.grid_n = pl.gn;
    g.slab = slab; g.slab_stride = slab_stride; g.ld_slab = pl.ld_slab;
    g.tile_major = ctx->tile_major;
    HIP_TRY(launch_gemm(LAYOUT_OUTER, pl.cfg, EPI_SLAB, g, st));
    return KURBM_OK;
}

// ---- C ABI -----------------------------------------------------------------------------
extern "C" {

int kurbm_abi_version(void) { return KURBM_ABI_VERSION; }

const char* kurbm_last_error(void) { return g_err.c_str(); }

int kurbm_ctx_create(int device, kurbm_ctx** out) {
    if (!out) return fail(KURBM_ERR_ARG, "out is null");
    *out = nullptr;
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail(KURBM_ERR_ARG, "device %d out of range (%d visible)", device, n);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(KURBM_ERR_UNSUPPORTED, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
    kurbm_ctx* c = new kurbm_ctx;
    c->device = device;
    c->ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    c->force_cfg[LAYOUT_VH] = env_int("KURBM_CFG_VH", -1);
    c->force_cfg[LAYOUT_HV] = env_int("KURBM_CFG_HV", -1);
    c->force_cfg[LAYOUT_OUTER] = env_int("KURBM_CFG_OUTER", -1);
    c->force_split = env_int("KURBM_SPLIT", -1);
    c->tile_major = env_int("KURBM_TILE_MAJOR", 1);
    for (int i = 0; i < KN_COUNT; ++i) c->knob[i] = env_int(KNOBS[i].env, KNOBS[i].dflt);
    c->status = nullptr;
    {
        int prev = 0;
        (void)hipGetDevice(&prev);
        hipError_t e = hipSetDevice(device);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->status), STATUS_BYTES);
        if (e == hipSuccess) e = hipMemset(c->status, 0, STATUS_BYTES);
        // where do the workgroups of a grid run?  (the XCD-local schedule of the one-launch small step stands on the answer)
        c->xcc_round_robin = 0; c->xcc_map = 0;
        if (e == hipSuccess && c->ncu >= 8 && c->ncu <= 1024) {
            unsigned* probe = nullptr;
            unsigned host[1024];
            if (hipMalloc(reinterpret_cast<void**>(&probe), sizeof(unsigned) * c->ncu) == hipSuccess) {
                if (launch_xcc_probe(probe, c->ncu, nullptr) == hipSuccess &&
                    hipMemcpy(host, probe, sizeof(unsigned) * c->ncu, hipMemcpyDeviceToHost) == hipSuccess) {
                    // the model the schedule stands on: eight XCDs, and on each of them exactly one workgroup of every octet
                    // (workgroup b: octet b / 8) -- consecutive workgroups go to consecutive XCDs, from wherever the dispatcher stood
                    bool rr = (c->ncu % 8) == 0;
                    unsigned seen[8][32];
                    memset(seen, 0, sizeof seen);
                    for (int b = 0; rr && b < c->ncu; ++b) {
                        const unsigned x = host[b];
                        if (x > 7u || (seen[x][(b >> 3) >> 5] >> ((b >> 3) & 31)) & 1u) rr = false;
                        else seen[x][(b >> 3) >> 5] |= 1u << ((b >> 3) & 31);
                    }
                    if (rr) {
                        c->xcc_round_robin = 1;
                        for (int g = 0; g < 8; ++g) c->xcc_map |= (host[g] & 15u) << (4 * g);
                    }
                }
                (void)hipFree(probe);
            }
            (void)hipGetLastError();
        }
        (void)hipSetDevice(prev);
        if (e != hipSuccess) {
            if (c->status) (void)hipFree(c->status);
            delete c;
            return fail(KURBM_ERR_HIP, "status word of the context: %s", hipGetErrorString(e));
        }
    }
    *out = c;
    return KURBM_OK;
}

int kurbm_ctx_status(kurbm_ctx* ctx, int* bits) {
    if (!ctx || !bits) return fail(KURBM_ERR_ARG, "null argument");
    unsigned v = 0;
    HIP_TRY(hipMemcpy(&v, ctx->status, sizeof v, hipMemcpyDeviceToHost));      // (synchronises with the device)
    if (v) HIP_TRY(hipMemset(ctx->status, 0, STATUS_BYTES));      // (also re-arms the grid barrier of kurbm_cd_step_small)
    *bits = (int)v;
    return KURBM_OK;
}

int kurbm_ctx_set_option(kurbm_ctx* ctx, const char* name, int value) {
    if (!ctx || !name) return fail(KURBM_ERR_ARG, "null argument");
    for (int i = 0; i < KN_COUNT; ++i)
        if (strcmp(name, KNOBS[i].env) == 0) { ctx->knob[i] = value; return KURBM_OK; }
    if (strcmp(name, "KURBM_CFG_VH") == 0) { ctx->force_cfg[LAYOUT_VH] = value; return KURBM_OK; }
    if (strcmp(name, "KURBM_CFG_HV") == 0) { ctx->force_cfg[LAYOUT_HV] = value; return KURBM_OK; }
    if (strcmp(name, "KURBM_CFG_OUTER") == 0) { ctx->force_cfg[LAYOUT_OUTER] = value; return KURBM_OK; }
    if (strcmp(name, "KURBM_SPLIT") == 0) { ctx->force_split = value; return KURBM_OK; }
    if (strcmp(name, "KURBM_TILE_MAJOR") == 0) { ctx->tile_major = value; return KURBM_OK; }
    return fail(KURBM_ERR_ARG, "unknown option %s", name);
}

void kurbm_ctx_destroy(kurbm_ctx* ctx) {
    if (!ctx) return;
    if (ctx->status) (void)hipFree(ctx->status);
    delete ctx;
}

int kurbm_philox_uniform(kurbm_ctx* ctx, float* out, int rows, int cols, int ld, const kurbm_rng* rng,
                         kurbm_stream_t stream) {
    if (!ctx || !out || !rng) return fail(KURBM_ERR_ARG, "null argument");
    if (rows <= 0 || cols <= 0 || ld < cols) return fail(KURBM_ERR_ARG, "bad shape");
    if (rng->row0 & 3) return fail(KURBM_ERR_ARG, "rng.row0 must be a multiple of 4");
    const RngArgs r = make_rng(rng->seed, rng->row0, rng->stream_id, rng->step);
    HIP_TRY(launch_philox_uniform(out, rows, cols, ld, r, static_cast<hipStream_t>(stream)));
    return KURBM_OK;
}

int kurbm_half_step_vh(kurbm_ctx* ctx, const kurbm_params* p, const float* v, int rows, int ldv, int act, int noise,
                       const kurbm_rng* rng, float* out_sample, float* out_prob, int ldh, kurbm_stream_t stream) {
    return kurbm_half_step_vh_dbg(ctx, p, v, rows, ldv, act, noise, rng, out_sample, out_prob, nullptr, ldh, stream);
}

int kurbm_half_step_hv(kurbm_ctx* ctx, const kurbm_params* p, const float* h, int rows, int ldh, int act, int noise,
                       const kurbm_rng* rng, float* out_sample, float* out_prob, int ldv, kurbm_stream_t stream) {
    return kurbm_half_step_hv_dbg(ctx, p, h, rows, ldh, act, noise, rng, out_sample, out_prob, nullptr, ldv, stream);
}

int kurbm_half_step_vh_dbg(kurbm_ctx* ctx, const kurbm_params* p, const float* v, int rows, int ldv, int act,
                           int noise, const kurbm_rng* rng, float* out_sample, float* out_prob, float* out_u, int ldh,
                           kurbm_stream_t stream) {
    if (!ctx) return fail(KURBM_ERR_ARG, "ctx is null");
    if (int e = check_params(p)) return e;
    RngArgs r;
    if (rng) r = make_rng(rng->seed, rng->row0, rng->stream_id, rng->step);
    return half_step(ctx, LAYOUT_VH, p, v, rows, ldv, act, noise, rng ? &r : nullptr, out_sample, out_prob, out_u, ldh,
                     nullptr, 0, nullptr, 0, nullptr, static_cast<hipStream_t>(stream));
}

int kurbm_half_step_hv_dbg(kurbm_ctx* ctx, const kurbm_params* p, const float* h, int rows, int ldh, int act,
                           int noise, const kurbm_rng* rng, float* out_sample, float* out_prob, float* out_u, int ldv,
                           kurbm_stream_t stream) {
    if (!ctx) return fail(KURBM_ERR_ARG, "ctx is null");
    if (int e = check_params(p)) return e;
    RngArgs r;
    if (rng) r = make_rng(rng->seed, rng->row0, rng->stream_id, rng->step);
    return half_step(ctx, LAYOUT_HV, p, h, rows, ldh, act, noise, rng ? &r : nullptr, out_sample, out_prob, out_u, ldv,
                     nullptr, 0, nullptr, 0, nullptr, static_cast<hipStream_t>(stream));
}

size_t kurbm_workspace_bytes(kurbm_ctx* ctx, int rows, int n_vis, int n_hid, int k) {
    if (!ctx || rows <= 0 || n_vis <= 0 || n_hid <= 0) return 0;
    return carve(ctx, nullptr, rows, n_vis, n_hid, k).bytes;
}

int kurbm_cd_step(kurbm_ctx* ctx, const kurbm_params* p, const float* v_batch, int rows, int ldv,
                  const kurbm_cd_opts* o, int which, void* workspace, size_t workspace_bytes, kurbm_stream_t stream) {
    if (!ctx || !o) return fail(KURBM_ERR_ARG, "null argument");
    if (int e = check_params(p)) return e;
    if (rows <= 0) return fail(KURBM_ERR_ARG, "rows must be positive");
    if (bad_matrix(v_batch, ldv, p->n_vis)) return fail(KURBM_ERR_ARG, "v_batch: null, misaligned, ld %% 4 != 0 or ld < n_vis");
    if (o->k < 1 || o->k > 15) return fail(KURBM_ERR_ARG, "k must be in [1, 15]");
    if (o->mode != KURBM_MODE_VISIBLE_BERNOULLI && o->mode != KURBM_MODE_VISIBLE_GAUSSIAN)
        return fail(KURBM_ERR_ARG, "unknown mode %d", o->mode);
    if (o->row0 & 3) return fail(KURBM_ERR_ARG, "row0 must be a multiple of 4");
    if (!workspace || !aligned16(workspace)) return fail(KURBM_ERR_ARG, "workspace is null or misaligned");
    if (o->v_chain && !aligned16(o->v_chain)) return fail(KURBM_ERR_ARG, "v_chain is misaligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const Workspace w = carve(ctx, workspace, rows, p->n_vis, p->n_hid, o->k);
    if (w.bytes > workspace_bytes)
        return fail(KURBM_ERR_WORKSPACE, "workspace too small: need %zu bytes, got %zu", w.bytes, workspace_bytes);

    const bool gauss = (o->mode == KURBM_MODE_VISIBLE_GAUSSIAN);
    const int act_h = gauss ? ACT_RELU : ACT_SIGMOID;             // rbm.py:59 vs :47
    const int act_v = gauss ? ACT_LINEAR : ACT_SIGMOID;           // rbm.py:64-65 vs :53
    const int noise_v = gauss ? NOISE_GAUSSIAN : NOISE_BERNOULLI;
    const uint32_t base = o->chain * 64u;
    const bool need_w = (which & 1) || o->delta_out;
    int e;

    // h_pos ~ p(h | v_pos)                                        rbm.py:120 (self.transform)
    RngArgs r = make_rng(o->seed, o->row0, base + 0u, o->step);
    if ((e = half_step(ctx, LAYOUT_VH, p, v_batch, rows, ldv, act_h, NOISE_BERNOULLI, &r, w.h_pos, nullptr, nullptr,
                       w.ldh, nullptr, 0, nullptr, 0, nullptr, st)))
        return e;
    const float* h_cur = w.h_pos;
    if (o->v_chain) {  // persistent chain: the negative phase starts from the stored fantasy
        r = make_rng(o->seed, o->row0, base + 32u, o->step);
        if ((e = half_step(ctx, LAYOUT_VH, p, o->v_chain, rows, ldv, act_h, NOISE_BERNOULLI, &r, w.h_tmp, nullptr,
                           nullptr, w.ldh, nullptr, 0, nullptr, 0, nullptr, st)))
            return e;
        h_cur = w.h_tmp;
    }
    // the final v sample is written straight into the persistent chain when there is one
    float* v_last = o->v_chain ? o->v_chain : w.v_neg;
    int gm_v = 0, gm_h = 0;
    for (int t = 1; t <= o->k; ++t) {
        const bool last = (t == o->k);
        // v_t ~ p(v | h_{t-1})                                    rbm.py:121-123 / :143-144
        r = make_rng(o->seed, o->row0, base + 2u * t - 1u, o->step);
        float* v_out = last ? v_last : w.v_neg;
        if ((e = half_step(ctx, LAYOUT_HV, p, h_cur, rows, w.ldh, act_v, noise_v, &r, v_out, nullptr, nullptr, ldv,
                           last ? v_batch : nullptr, ldv, w.part_v, w.ld_part_v, last ? &gm_v : nullptr, st)))
            return e;
        if (!last) {
            r = make_rng(o->seed, o->row0, base + 2u * t, o->step);
            if ((e = half_step(ctx, LAYOUT_VH, p, v_out, rows, ldv, act_h, NOISE_BERNOULLI, &r, w.h_tmp, nullptr, nullptr,
                               w.ldh, nullptr, 0, nullptr, 0, nullptr, st)))
                return e;
            h_cur = w.h_tmp;
        }
    }
    // h_neg = sigmoid(v_neg.W + b_h): probabilities, sigmoid in both modes   rbm.py:124 / :145
    if ((e = half_step(ctx, LAYOUT_VH, p, v_last, rows, ldv, ACT_SIGMOID, NOISE_NONE, nullptr, nullptr, w.h_neg, nullptr,
                       w.ldh, w.h_pos, w.ldh, w.part_h, w.ld_part_h, &gm_h, st)))
        return e;

    // dW = v_pos^T.h_pos - v_neg^T.h_neg  (rbm.py:125-126) as split-K slabs
    const OuterPlan pl = plan_outer(ctx, rows, p->n_vis, p->n_hid);
    if (need_w)
        if ((e = outer_slabs(ctx, v_batch, w.h_pos, v_last, w.h_neg, rows, p->n_vis, p->n_hid, ldv, w.ldh, w.slab,
                             w.slab_stride, pl, st)))
            return e;

    // reduce slabs / bias partials, apply lr * sums (rbm.py:127-134) and/or emit the packed delta
    ReduceArgs a;
    memset(&a, 0, sizeof a);
    a.slab = w.slab; a.slab_stride = w.slab_stride; a.nslab = pl.nsplit; a.ld_slab = pl.ld_slab;
    a.n_vis = p->n_vis; a.n_hid = p->n_hid; a.ldw = p->ldw;
    a.nblk_w = need_w ? (int)(((long long)p->n_vis * (pl.ld_slab / 4) + 255) / 256) : 0;
    if (need_w && ctx->tile_major) {
        tile_shape(pl.cfg, &a.tile_bm, &a.tile_bn);
        a.grid_m = pl.gm; a.grid_n = pl.gn; a.parts = 4; a.m_fastest = 0;
        a.nblk_w = pl.gm * pl.gn * a.parts;
    }
    a.lr = o->lr;
    const bool ap = o->apply != 0;
    a.W = (ap && (which & 1)) ? p->W : nullptr;
    a.delta_w = o->delta_out;
    a.part_h = w.part_h; a.nrow_tiles_h = gm_h; a.ld_part_h = w.ld_part_h;
    a.b_h = (ap && (which & 2)) ? p->b_h : nullptr;
    a.delta_bh = o->delta_out ? o->delta_out + (size_t)p->n_vis * p->n_hid : nullptr;
    a.part_v = w.part_v; a.nrow_tiles_v = gm_v; a.ld_part_v = w.ld_part_v;
    a.b_v = (ap && (which & 4)) ? p->b_v : nullptr;
    a.delta_bv = o->delta_out ? o->delta_out + (size_t)p->n_vis * p->n_hid + p->n_hid : nullptr;
    HIP_TRY(launch_reduce_apply(a, st));
    return KURBM_OK;
}

// One CD-1 update of a small RBM in ONE launch (kurbm_small.hip).  Workspace as kurbm_cd_step (kurbm_workspace_bytes).
int kurbm_cd_step_small(kurbm_ctx* ctx, const kurbm_params* p, const float* v_batch, int rows, int ldv, const kurbm_cd_opts* o,
                        int which, void* workspace, size_t workspace_bytes, kurbm_stream_t stream) {
    if (!ctx || !o) return fail(KURBM_ERR_ARG, "null argument");
    if (int e = check_params(p)) return e;
    if (rows <= 0) return fail(KURBM_ERR_ARG, "rows must be positive");
    if (bad_matrix(v_batch, ldv, p->n_vis)) return fail(KURBM_ERR_ARG, "v_batch: null, misaligned, ld %% 4 != 0 or ld < n_vis");
    if (o->k != 1 || o->v_chain || !o->apply || o->delta_out)
        return fail(KURBM_ERR_UNSUPPORTED, "kurbm_cd_step_small runs CD-1 from the data, applied in place (k = 1, no v_chain, apply = 1, no delta_out)");
    if (o->mode != KURBM_MODE_VISIBLE_BERNOULLI && o->mode != KURBM_MODE_VISIBLE_GAUSSIAN) return fail(KURBM_ERR_ARG, "unknown mode %d", o->mode);
    if (o->row0 & 3) return fail(KURBM_ERR_ARG, "row0 must be a multiple of 4");
    if (!workspace || !aligned16(workspace)) return fail(KURBM_ERR_ARG, "workspace is null or misaligned");
    const Workspace w = carve(ctx, workspace, rows, p->n_vis, p->n_hid, 1);
    if (w.bytes > workspace_bytes) return fail(KURBM_ERR_WORKSPACE, "workspace too small: need %zu bytes, got %zu", w.bytes, workspace_bytes);
    SmallArgs a;
    memset(&a, 0, sizeof a);
    a.W = p->W; a.b_h = p->b_h; a.b_v = p->b_v; a.n_vis = p->n_vis; a.n_hid = p->n_hid; a.ldw = p->ldw;
    a.v = v_batch; a.rows = rows; a.ldv = ldv;
    if (rows > SMALL_ROWS_MAX) return fail(KURBM_ERR_UNSUPPORTED, "kurbm_cd_step_small: at most %d rows per batch", SMALL_ROWS_MAX);
    a.h_pos = w.h_pos; a.h_neg = w.h_neg; a.v_neg = w.v_neg; a.ldh = w.ldh; a.ldn = w.ldv;
    a.ldt = round_up(rows, 16);
    a.h_posT = w.small_t; a.h_negT = a.h_posT + (size_t)p->n_hid * a.ldt; a.v_negT = a.h_negT + (size_t)p->n_hid * a.ldt;
    a.bar = ctx->status + 64; a.status = ctx->status;
    a.timeout_ticks = 200000000ull;          // 2 s of the 100 MHz clock: only a grid that is not resident ever gets there
    const uint32_t base = o->chain * 64u;
    a.rng_h = make_rng(o->seed, o->row0, base + 0u, o->step);
    a.rng_v = make_rng(o->seed, o->row0, base + 1u, o->step);
    a.which = which; a.gauss = (o->mode == KURBM_MODE_VISIBLE_GAUSSIAN) ? 1 : 0; a.lr = o->lr;
    // workgroups: enough for the widest phase (one 16 x 16 tile of a half step per workgroup, one tile of W per wave), never more
    // than the CUs -- the grid must be resident, one workgroup per CU
    const int tm = ceil_div(rows, 16), tv = ceil_div(p->n_vis, 16), th = ceil_div(p->n_hid, 16);
    int nblk = tm * th > tm * tv ? tm * th : tm * tv;
    const int stat = ceil_div(tv * th + tv + th, 8);   // (eight waves per workgroup: kurbm_small.hip SMALL_WAVES)
    if (stat > nblk) nblk = stat;
    if (nblk > ctx->ncu) nblk = ctx->ncu;
    if (nblk > 1024) nblk = 1024;
    // the XCD-local schedule: every XCD's workgroups own the row tiles tm = its number (mod 8), so the whole grid is launched
    a.local = (ctx->knob[KN_SMALL_LOCAL] != 0 && ctx->xcc_round_robin && ctx->ncu <= 256) ? 1 : 0;   // (a group's barrier: 32 words)
    a.xcc_map = ctx->xcc_map;
    if (a.local) nblk = ctx->ncu;
    HIP_TRY(launch_cd1_small(a, nblk, static_cast<hipStream_t>(stream)));
    return KURBM_OK;
}

// The per-step score of fit(verbose = 1) for a small RBM in ONE launch (kurbm_small.hip: k_score_small).
int kurbm_score_small(kurbm_ctx* ctx, const kurbm_params* p, const float* v_batch, int rows, int ldv, const kurbm_cd_opts* o,
                      float* score, float* F, void* workspace, size_t workspace_bytes, kurbm_stream_t stream) {
    if (!ctx || !o || !score) return fail(KURBM_ERR_ARG, "null argument");
    if (reinterpret_cast<uintptr_t>(score) & 7) return fail(KURBM_ERR_ARG, "score must be 8-byte aligned (two floats are written)");
    if (int e = check_params(p)) return e;
    if (rows <= 0 || rows > SMALL_ROWS_MAX) return fail(KURBM_ERR_ARG, "rows must be in [1, %d]", SMALL_ROWS_MAX);
    if (bad_matrix(v_batch, ldv, p->n_vis)) return fail(KURBM_ERR_ARG, "v_batch: null, misaligned, ld %% 4 != 0 or ld < n_vis");
    if (o->mode != KURBM_MODE_VISIBLE_BERNOULLI && o->mode != KURBM_MODE_VISIBLE_GAUSSIAN) return fail(KURBM_ERR_ARG, "unknown mode %d", o->mode);
    if (o->row0 & 3) return fail(KURBM_ERR_ARG, "row0 must be a multiple of 4");
    if (!workspace || !aligned16(workspace)) return fail(KURBM_ERR_ARG, "workspace is null or misaligned");
    const Workspace w = carve(ctx, workspace, rows, p->n_vis, p->n_hid, 1);
    if (w.bytes > workspace_bytes) return fail(KURBM_ERR_WORKSPACE, "workspace too small: need %zu bytes, got %zu", w.bytes, workspace_bytes);
    SmallArgs a;
    memset(&a, 0, sizeof a);
    a.W = p->W; a.b_h = p->b_h; a.b_v = p->b_v; a.n_vis = p->n_vis; a.n_hid = p->n_hid; a.ldw = p->ldw;
    a.v = v_batch; a.rows = rows; a.ldv = ldv;
    a.h_pos = w.h_pos; a.h_neg = w.h_neg; a.v_neg = w.v_neg; a.ldh = w.ldh; a.ldn = w.ldv;
    a.ldt = round_up(rows, 16);
    a.h_posT = w.small_t;                    // (the row partials live in the transposed planes' space)
    a.bar = ctx->status + 64; a.status = ctx->status;
    a.timeout_ticks = 200000000ull;
    const uint32_t base = o->chain * 64u;
    a.rng_h = make_rng(o->seed, o->row0, base + 0u, o->step);
    a.rng_v = make_rng(o->seed, o->row0, base + 1u, o->step);
    a.gauss = (o->mode == KURBM_MODE_VISIBLE_GAUSSIAN) ? 1 : 0;
    a.score = score; a.F = F;
    const int tm = ceil_div(rows, 16), tv = ceil_div(p->n_vis, 16), th = ceil_div(p->n_hid, 16);
    int nblk = tm * th > tm * tv ? tm * th : tm * tv;
    if (nblk > ctx->ncu) nblk = ctx->ncu;
    a.local = (ctx->knob[KN_SMALL_LOCAL] != 0 && ctx->xcc_round_robin && ctx->ncu <= 256) ? 1 : 0;
    a.xcc_map = ctx->xcc_map;
    if (a.local) nblk = ctx->ncu;
    HIP_TRY(launch_score_small(a, nblk, static_cast<hipStream_t>(stream)));
    return KURBM_OK;
}

// An epoch of fit(verbose = 1) for a small RBM in one call: every batch's update followed by its score (rbm.py:211-234), two
// launches per step, nothing read back -- scores[2 * step] receives the score and scores[2 * step + 1] a 1.0f when it has landed
// (device memory, or pinned host memory that the caller polls while the device runs on).  Returns the number of steps.
int kurbm_cd_epoch_small_scored(kurbm_ctx* ctx, const kurbm_params* p, const float* V, int n_rows, int ldv, int batch_size,
                                const kurbm_cd_opts* opts, int score_chain, float* scores, void* workspace, size_t workspace_bytes,
                                kurbm_stream_t stream) {
    if (!ctx || !opts || !scores) return fail(KURBM_ERR_ARG, "null argument");
    if (n_rows < 0 || batch_size <= 0) return fail(KURBM_ERR_ARG, "bad row count / batch size");
    kurbm_cd_opts o = *opts, so = *opts;
    so.chain = score_chain;
    int steps = 0;
    for (int lo = 0; lo < n_rows; lo += batch_size, ++steps) {
        const int rows = (n_rows - lo < batch_size) ? n_rows - lo : batch_size;
        if (int e = kurbm_cd_step_small(ctx, p, V + (size_t)lo * ldv, rows, ldv, &o, 7, workspace, workspace_bytes, stream)) return e;
        if (int e = kurbm_score_small(ctx, p, V + (size_t)lo * ldv, rows, ldv, &so, scores + 2 * (size_t)steps, nullptr, workspace,
                                      workspace_bytes, stream)) return e;
        ++o.step; ++so.step;
    }
    return steps;
}

int kurbm_cd_epoch_small(kurbm_ctx* ctx, const kurbm_params* p, const float* V, int n_rows, int ldv, int batch_size,
                         const kurbm_cd_opts* opts, void* workspace, size_t workspace_bytes, kurbm_stream_t stream) {
    if (!ctx || !opts) return fail(KURBM_ERR_ARG, "null argument");
    if (n_rows < 0 || batch_size <= 0) return fail(KURBM_ERR_ARG, "bad row count / batch size");
    kurbm_cd_opts o = *opts;
    int steps = 0;
    for (int lo = 0; lo < n_rows; lo += batch_size, ++steps) {
        const int rows = (n_rows - lo < batch_size) ? n_rows - lo : batch_size;
        if (int e = kurbm_cd_step_small(ctx, p, V + (size_t)lo * ldv, rows, ldv, &o, 7, workspace, workspace_bytes, stream)) return e;
        ++o.step;
    }
    return steps;
}

int kurbm_cd_epoch(kurbm_ctx* ctx, const kurbm_params* p, const float* V, int n_rows, int ldv, int batch_size,
                   const kurbm_cd_opts* opts, void* workspace, size_t workspace_bytes, kurbm_stream_t stream) {
    if (!ctx || !opts) return fail(KURBM_ERR_ARG, "null argument");
    if (n_rows < 0 || batch_size <= 0) return fail(KURBM_ERR_ARG, "bad row count / batch size");
    if (opts->delta_out || !opts->apply) return fail(KURBM_ERR_ARG, "kurbm_cd_epoch applies in place: apply = 1, delta_out = null");
    kurbm_cd_opts o = *opts;
    int steps = 0;
    for (int lo = 0; lo < n_rows; lo += batch_size, ++steps) {
        const int rows = (n_rows - lo < batch_size) ? n_rows - lo : batch_size;
        if (int e = kurbm_cd_step(ctx, p, V + (size_t)lo * ldv, rows, ldv, &o, 7, workspace, workspace_bytes, stream)) return e;
        ++o.step;
    }
    return steps;
}

int kurbm_apply_delta(kurbm_ctx* ctx, const kurbm_params* p, const float* delta, float lr, int which,
                      kurbm_stream_t stream) {
    if (!ctx || !delta) return fail(KURBM_ERR_ARG, "null argument");
    if (int e = check_params(p)) return e;
    ApplyArgs a;
    a.delta = delta;
    a.W = (which & 1) ? p->W : nullptr;
    a.b_h = (which & 2) ? p->b_h : nullptr;
    a.b_v = (which & 4) ? p->b_v : nullptr;
    a.n_vis = p->n_vis; a.n_hid = p->n_hid; a.ldw = p->ldw; a.lr = lr;
    HIP_TRY(launch_apply_delta(a, static_cast<hipStream_t>(stream)));
    return KURBM_OK;
}

int kurbm_free_energy(kurbm_ctx* ctx, const kurbm_params* p, const float* v, int rows, int ldv, float* F,
                      void* workspace, size_t workspace_bytes, kurbm_stream_t stream) {
    if (!ctx || !F) return fail(KURBM_ERR_ARG, "null argument");
    if (int e = check_params(p)) return e;
    if (rows <= 0) return fail(KURBM_ERR_ARG, "rows must be positive");
    if (bad_matrix(v, ldv, p->n_vis)) return fail(KURBM_ERR_ARG, "v: null, misaligned, ld %% 4 != 0 or ld < n_vis");
    if (!workspace || !aligned16(workspace)) return fail(KURBM_ERR_ARG, "workspace is null or misaligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    GemmArgs g;
    memset(&g, 0, sizeof g);
    g.A0 = v; g.lda = ldv; g.B0 = p->W; g.ldb = p->ldw;
    g.M = rows; g.N = p->n_hid; g.K = p->n_vis;
    g.nseg = 1;
    g.nkt = g.K / 32; g.kt_total = g.nkt; g.kt_per_split = g.nkt; g.nsplit = 1;
    const int cfg = pick_cfg(ctx->ncu, rows, g.N, &g.grid_m, &g.grid_n, ctx->force_cfg[LAYOUT_VH]);
    g.bias = p->b_h;
    g.rowpart = static_cast<float*>(workspace);
    g.ld_rowpart = round_up(rows, 4);
    const size_t need = (size_t)g.grid_n * g.ld_rowpart * 4;
    if (need > workspace_bytes) return fail(KURBM_ERR_WORKSPACE, "workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
    HIP_TRY(launch_gemm(LAYOUT_VH, cfg, EPI_SOFTPLUS, g, st));
    FinishArgs f;
    f.v = v; f.b_v = p->b_v; f.rowpart = g.rowpart; f.F = F;
    f.rows = rows; f.n_vis = p->n_vis; f.ldv = ldv; f.ncol_tiles = g.grid_n; f.ld_rowpart = g.ld_rowpart;
    HIP_TRY(launch_free_energy_finish(f, st));
    return KURBM_OK;
}

int kurbm_outer_delta(kurbm_ctx* ctx, const float* v_pos, const float* h_pos, const float* v_neg, const float* h_neg,
                      int rows, int n_vis, int n_hid, int ldv, int ldh, float* delta_w, void* workspace,
                      size_t workspace_bytes, kurbm_stream_t stream) {
    if (!ctx || !delta_w) return fail(KURBM_ERR_ARG, "null argument");
    if (rows <= 0 || n_vis <= 0 || n_hid <= 0) return fail(KURBM_ERR_ARG, "bad shape");
    if (bad_matrix(v_pos, ldv, n_vis) || bad_matrix(v_neg, ldv, n_vis) || bad_matrix(h_pos, ldh, n_hid) ||
        bad_matrix(h_neg, ldh, n_hid))
        return fail(KURBM_ERR_ARG, "operand: null, misaligned, ld %% 4 != 0 or ld < columns");
    if (!workspace || !aligned16(workspace)) return fail(KURBM_ERR_ARG, "workspace is null or misaligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const Workspace w = carve(ctx, workspace, rows, n_vis, n_hid, 1);
    if (w.bytes > workspace_bytes)
        return fail(KURBM_ERR_WORKSPACE, "workspace too small: need %zu bytes, got %zu", w.bytes, workspace_bytes);
    const OuterPlan pl = plan_outer(ctx, rows, n_vis, n_hid);
    if (int e = outer_slabs(ctx, v_pos, h_pos, v_neg, h_neg, rows, n_vis, n_hid, ldv, ldh, w.slab, w.slab_stride, pl, st))
        return e;
    ReduceArgs a;
    memset(&a, 0, sizeof a);
    a.slab = w.slab; a.slab_stride = w.slab_stride; a.nslab = pl.nsplit; a.ld_slab = pl.ld_slab;
    a.n_vis = n_vis; a.n_hid = n_hid; a.ldw = 0;
    a.nblk_w = (int)(((long long)n_vis * (pl.ld_slab / 4) + 255) / 256);
    if (ctx->tile_major) {
        tile_shape(pl.cfg, &a.tile_bm, &a.tile_bn);
        a.grid_m = pl.gm; a.grid_n = pl.gn; a.parts = 4; a.m_fastest = 0;
        a.nblk_w = pl.gm * pl.gn * a.parts;
    }
    a.delta_w = delta_w;
    a.n_hid = n_hid;
    // no bias work: n_hid/n_vis bias blocks see null partial pointers and do nothing
    HIP_TRY(launch_reduce_apply(a, st));
    return KURBM_OK;
}

int kurbm_outer_partial(kurbm_ctx* ctx, const float* v_pos, const float* h_pos, const float* v_neg, const float* h_neg,
                        int rows, int n_vis, int n_hid, int ldv, int ldh, void* workspace, size_t workspace_bytes,
                        kurbm_stream_t stream) {
    if (!ctx) return fail(KURBM_ERR_ARG, "null argument");
    if (rows <= 0 || n_vis <= 0 || n_hid <= 0) return fail(KURBM_ERR_ARG, "bad shape");
    if (bad_matrix(v_pos, ldv, n_vis) || bad_matrix(v_neg, ldv, n_vis) || bad_matrix(h_pos, ldh, n_hid) ||
        bad_matrix(h_neg, ldh, n_hid))
        return fail(KURBM_ERR_ARG, "operand: null, misaligned, ld %% 4 != 0 or ld < columns");
    if (!workspace || !aligned16(workspace)) return fail(KURBM_ERR_ARG, "workspace is null or misaligned");
    const Workspace w = carve(ctx, workspace, rows, n_vis, n_hid, 1);
    if (w.bytes > workspace_bytes)
        return fail(KURBM_ERR_WORKSPACE, "workspace too small: need %zu bytes, got %zu", w.bytes, workspace_bytes);
    const OuterPlan pl = plan_outer(ctx, rows, n_vis, n_hid);
    return outer_slabs(ctx, v_pos, h_pos, v_neg, h_neg, rows, n_vis, n_hid, ldv, ldh, w.slab, w.slab_stride, pl,
                       static_cast<hipStream_t>(stream));
}

#ifdef KURBM_SMALL_STAMPS
/* diagnostic build only: the eight phase stamps of the last kurbm_cd_step_small (100 MHz ticks) */
int kurbm_debug_small_stamps(kurbm_ctx* ctx, unsigned long long* out8) {
    HIP_TRY(hipMemcpy(out8, ctx->status + 40, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return KURBM_OK;
}
/* ... and the fine stamps of the h -> v phase (workgroup 0, its first pass) */
int kurbm_debug_small_fine_stamps(kurbm_ctx* ctx, unsigned long long* out8) {
    HIP_TRY(hipMemcpy(out8, ctx->status + 768, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return KURBM_OK;
}
#endif
#ifdef KURBM_STAMPS
/* diagnostic library only: not part of include/kurbm.h */
void kurbm_debug_set_stamp_buffer(void* p) { set_stamp_buffer(static_cast<unsigned long long*>(p)); }
void kurbm_debug_set_off(int mask) { set_debug_off(mask); }
#endif

}  // extern "C"

// ======================================================================================
// bf16-operand variants (kurbm_bf16.hip): mirrors, workspace, launch sequences
//   pieces = 1  "bf16": operands rounded to bf16 (BASELINE.json config 5)
//   pieces = 3  "x3":   every fp32 operand carried EXACTLY as three bf16 pieces (hi + mid + lo), one
//                       k-segment of the NT GEMM per pair of pieces; 0/1 samples are a single piece
// ======================================================================================
// Leading dimension of a bf16 plane = its k extent (+ KURBM_LDPAD elements: an experiment knob.  Row strides
// that are multiples of 2 KiB were suspected of camping on one L2 channel; padding them changed nothing.)
static inline int ld_pad(const kurbm_ctx* ctx, int k) { return k + ctx->knob[KN_LDPAD]; }

struct Mirror { uint16_t *Wb, *Wtb; int Kh, Kv, ldW, ldWt, pieces; size_t planeW, planeWt, bytes; };

static Mirror carve_mirror(const kurbm_ctx* ctx, void* base, int n_vis, int n_hid, int pieces) {
    Mirror m;
    m.pieces = pieces;
    m.Kh = round_up(n_hid, 128);   // k extent of W  [n_vis][Kh]   (B operand of h->v)
    m.Kv = round_up(n_vis, 128);   // k extent of Wt [n_hid][Kv]   (B operand of v->h)
    m.ldW = ld_pad(ctx, m.Kh);
    m.ldWt = ld_pad(ctx, m.Kv);
    m.planeW = (size_t)n_vis * m.ldW;
    m.planeWt = (size_t)n_hid * m.ldWt;
    char* b = static_cast<char*>(base);
    size_t off = 0;
    m.Wb = reinterpret_cast<uint16_t*>(b + off);  off = align_up(off + pieces * m.planeW * 2);
    m.Wtb = reinterpret_cast<uint16_t*>(b + off); off = align_up(off + pieces * m.planeWt * 2);
    m.bytes = off;
    return m;
}

struct WorkspaceB {
    uint16_t *vb, *vbT, *hb, *hbT, *v2b, *v2bT, *h2b, *hnT, *cb;
    float *part_h, *part_v, *slab, *tmp32;
    float* rowpart;     // kurbm_score_x3's row partials
    int Kv, Kh, Kb, Lv, Lh, Lb, ldh32, ldv32, max_row_tiles;   // K*: k extents; L*: leading dimensions of the bf16 planes
    size_t planeV, planeVT, planeHT;   // distance between the pieces of v_pos (both images) and of h_neg^T
    size_t slab_stride, bytes;
};

struct OuterPlanB { int gm, gn, nkt, kt_total, nsplit, nsplit_bound, kt_per_split, ld_slab, cfg; };

// the statistics GEMM on k_gemm_pb: k-tile 64, one workgroup per CU, nseg segments walked fastest, so a slice is a whole
// number of k positions
static OuterPlanB plan_outer_bf16(const kurbm_ctx* ctx, int rows, int n_vis, int n_hid, int nseg, int s_max = 1 << 30,
                                  bool f8pos = false, bool x3 = false) {
    OuterPlanB pl;
    // x3: 256 x 64 tiles -- the three-piece B tile is the heavy operand (3 x 8 KB against 3 x 16), so a k-tile is 56 KB instead
    // of 64 for the same MFMAs, and 784 x 1024 fills 256 workgroups instead of 224 (statistics GEMM 28.5 -> 27.4 us)
    pl.cfg = (x3 && ctx->knob[KN_X3_STATS_TALL] != 0 && n_vis > 128) ? 2 : 0;
    pl.gm = ceil_div(n_vis, pl.cfg == 2 ? 256 : 128);
    pl.gn = ceil_div(n_hid, pl.cfg == 2 ? 64 : 128);
    // f8pos: the walk is in units of 128 k -- one fp8 tile of segment 0, two 64-deep tiles of every other segment
    const int per = f8pos ? 2 * nseg - 1 : nseg;
    pl.nkt = round_up(rows, 128) / (f8pos ? 128 : 64);
    pl.kt_total = per * pl.nkt;
    int s = ctx->knob[KN_BF16_SPLIT] != KN_AUTO ? ctx->knob[KN_BF16_SPLIT] : ctx->ncu / (pl.gm * pl.gn);
    if (s > s_max) s = s_max;
    if (s < 1) s = 1;
    if (s > pl.nkt) s = pl.nkt;
    pl.nsplit_bound = s;
    pl.kt_per_split = ceil_div(pl.kt_total, s);
    pl.kt_per_split = round_up(pl.kt_per_split, per);
    pl.nsplit = ceil_div(pl.kt_total, pl.kt_per_split);
    pl.ld_slab = round_up(n_hid, 4);
    return pl;
}

// segment codes of k_gemm_pb: A piece ia against B pieces 0 .. npb-1
static int pb_codes(const kurbm_ctx* ctx, int a_pieces, int b_pieces, unsigned set, unsigned long long* codes, int nseg) {
    const bool full = ctx->knob[KN_X3_FULL] != 0;
    for (int ia = 0; ia < a_pieces; ++ia) {
        int npb = b_pieces;
        if (a_pieces > 1 && b_pieces > 1 && !full) npb = b_pieces - ia;     // pairs with ia + ib <= 2
        if (npb < 1) npb = 1;
        *codes |= (unsigned long long)((unsigned)ia | ((unsigned)npb << 2) | (set << 4)) << (5 * nseg);
        ++nseg;
    }
    return nseg;
}

// The statistics of REAL-VALUED data (x3, v_pieces = 3) as two launches (KURBM_X3_SPLIT_STATS).
//   positive:  the transposed problem -- A = h_pos^T [n_hid][batch] as a k-permuted byte plane, B = the three pieces of v_pos^T
//              [n_vis][batch]; 256 x 64 tiles of (hidden x visible) that leave transposed into ordinary slabs;
//   negative:  Bernoulli visibles: A = v_neg^T as bytes, B = the pieces of -h_neg^T, 256 x 64 tiles (the plain byte-plane walk);
//              Gaussian visibles: A = the three pieces of v_neg^T, B = those of -h_neg^T, 128 x 128 tiles on the paired walk.
// Each launch cuts k into slices of whole 128-deep blocks so that its grid fills the chip; the slabs of both are summed by the
// one reduce launch (negative slabs first).
struct SplitStats { OuterPlanB pos, neg; };
static SplitStats plan_split_stats(const kurbm_ctx* ctx, int rows, int n_vis, int n_hid, bool gauss, int s_max = 1 << 30) {
    SplitStats sp;
    const int nkt = round_up(rows, 128) / 64;
    auto fill = [&](OuterPlanB& pl, int gm, int gn, int cfg, int per) {
        pl.gm = gm; pl.gn = gn; pl.cfg = cfg;
        pl.nkt = nkt; pl.kt_total = per * nkt;
        int s = ctx->knob[KN_BF16_SPLIT] != KN_AUTO ? ctx->knob[KN_BF16_SPLIT] : ctx->ncu / (gm * gn);
        if (s > s_max) s = s_max;      // (a row range of the statistics keeps inside the slab memory carved for the whole matrix)
        if (s < 1) s = 1;
        if (s > nkt / 2) s = nkt / 2;
        if (s < 1) s = 1;
        pl.nsplit_bound = s;
        pl.kt_per_split = round_up(ceil_div(pl.kt_total, s), 2 * per);      // whole 128-deep blocks (two k positions)
        pl.nsplit = ceil_div(pl.kt_total, pl.kt_per_split);
        pl.ld_slab = round_up(n_hid, 4);
    };
    fill(sp.pos, ceil_div(n_hid, 256), ceil_div(n_vis, 64), 2, 1);
    if (gauss) fill(sp.neg, ceil_div(n_vis, 128), ceil_div(n_hid, 128), 0, 3);
    else fill(sp.neg, ceil_div(n_vis, 256), ceil_div(n_hid, 64), 2, 1);
    return sp;
}

static WorkspaceB carve_bf16(const kurbm_ctx* ctx, void* base, int rows, int n_vis, int n_hid, int pieces, int v_pieces) {
    WorkspaceB w;
    w.Kv = round_up(n_vis, 128);
    w.Kh = round_up(n_hid, 128);
    w.Kb = round_up(rows, 128);
    w.Lv = ld_pad(ctx, w.Kv); w.Lh = ld_pad(ctx, w.Kh); w.Lb = ld_pad(ctx, w.Kb);
    w.ldh32 = round_up(n_hid, 4);
    w.ldv32 = round_up(n_vis, 4);
    w.max_row_tiles = ceil_div(rows, 128);
    w.planeV = (size_t)w.Kb * w.Lv;
    w.planeVT = (size_t)n_vis * w.Lb;
    w.planeHT = (size_t)n_hid * w.Lb;
    const OuterPlanB pl = plan_outer_bf16(ctx, rows, n_vis, n_hid, pieces == 3 ? v_pieces + 1 : 2, 1 << 30, false, pieces == 3);
    w.slab_stride = (size_t)n_vis * pl.ld_slab;
    char* b = static_cast<char*>(base);
    size_t off = 0;
    auto take16 = [&](size_t n) { uint16_t* p = reinterpret_cast<uint16_t*>(b + off); off = align_up(off + n * 2); return p; };
    auto take32 = [&](size_t n) { float* p = reinterpret_cast<float*>(b + off); off = align_up(off + n * 4); return p; };
    w.vb = take16(v_pieces * w.planeV);    w.vbT = take16(v_pieces * w.planeVT);     // v_pos, both orientations
    w.hb = take16((size_t)w.Kb * w.Lh);    w.hbT = take16((size_t)n_hid * w.Lb);     // h_pos
    w.v2b = take16((pieces == 3 ? 3 : 1) * w.planeV);                                // v_t / v_neg (x3: room for the three
    w.v2bT = take16((pieces == 3 ? 3 : 1) * w.planeVT);                              //  pieces of Gaussian visibles)
    w.h2b = take16((size_t)w.Kb * w.Lh);                                             // h_t (k > 1, chain start)
    w.hnT = take16(pieces * w.planeHT);                                              // h_neg probabilities, transposed
    w.cb = take16(v_pieces * w.planeV);                                              // persistent chain as bf16
    // bias partials.  bf16: one row per row tile.  x3: hidden = rows of +sum(h_pos) then rows of -sum(h_neg);
    // visible = one row per 64-row band of v_pos (conversion kernel) then rows of -sum(v_neg)
    // (k_gemm_pb writes one row per 64-row half of a 128-row tile: two rows per tile)
    w.part_h = take32((size_t)4 * w.max_row_tiles * w.ldh32);
    w.part_v = take32((size_t)(ceil_div(rows, 64) + 2 * w.max_row_tiles) * w.ldv32);
    int nslab_res = pl.nsplit_bound;
    if (pieces == 3 && v_pieces == 3) {   // (the mode is not known here: room for either negative launch)
        const SplitStats b = plan_split_stats(ctx, rows, n_vis, n_hid, false), gs = plan_split_stats(ctx, rows, n_vis, n_hid, true);
        const int need = b.pos.nsplit_bound + (b.neg.nsplit_bound > gs.neg.nsplit_bound ? b.neg.nsplit_bound : gs.neg.nsplit_bound);
        if (need > nslab_res) nslab_res = need;
    }
    w.slab = take32(w.slab_stride * nslab_res);
    w.tmp32 = take32((size_t)rows * (w.ldh32 > w.ldv32 ? w.ldh32 : w.ldv32));       // fp32 plane for the test hook
    // the score of fit(verbose = 1): softplus row partials of F(v) and F(v') per 64-column tile, |F - F'| per row
    w.rowpart = take32((size_t)(2 * ceil_div(n_hid, 64) + 1) * round_up(rows, 4));
    w.bytes = off;
    return w;
}

// The bf16 planes of a window of `rows` data rows, as the x3 step reads them: pieces of v row-major [Kb][Lv] (A operand of
// the first half step), transposed [n_vis][Lb] (A operand of the positive statistics), and the column sums of each 64-row
// band (the positive half of db_v).  Same extents as the v_pos planes of a workspace carved for `rows` rows, so a step can
// read them IN PLACE OF its own conversion (kurbm_cd_opts::v_planes): the data matrix is static for a whole fit().
struct VPlanes { uint16_t *vb, *vbT; float* part_v; size_t bytes; };
static VPlanes carve_vplanes(const kurbm_ctx* ctx, const void* base, int rows, int n_vis, int v_pieces) {
    VPlanes v;
    const int Kv = round_up(n_vis, 128), Kb = round_up(rows, 128);
    const size_t planeV = (size_t)Kb * ld_pad(ctx, Kv), planeVT = (size_t)n_vis * ld_pad(ctx, Kb);
    char* b = static_cast<char*>(const_cast<void*>(base));
    size_t off = 0;
    v.vb = reinterpret_cast<uint16_t*>(b + off);  off = align_up(off + v_pieces * planeV * 2);
    v.vbT = reinterpret_cast<uint16_t*>(b + off); off = align_up(off + v_pieces * planeVT * 2);
    v.part_v = reinterpret_cast<float*>(b + off); off = align_up(off + (size_t)ceil_div(rows, 64) * round_up(n_vis, 4) * 4);
    v.bytes = off;
    return v;
}

static inline uint32_t inv_of(int nkt) { return nkt > 1 ? (uint32_t)(0x100000000ull / (unsigned)nkt) + 1u : 0u; }

// optional outputs and side products of a bf16 half step
struct HalfOutB {
    uint16_t* out = nullptr; int ldo = 0;                 // bf16 value plane, row-major
    int out_pieces = 1; size_t out_plane = 0;             // (x3, real-valued plane: its three pieces)
    int out_rows_pad = 0;                                 // rows [rows, out_rows_pad) of that plane are written as zeros
    uint16_t* outT = nullptr; int ldoT = 0;               // ... transposed
    int outT_pieces = 1; size_t outT_plane = 0;
    bool outT_neg = false;                                // the transposed plane is stored negated
    bool outT_f8 = false;                                 // the transposed plane of a 0/1 sample as fp8 bytes
    bool outT_b8 = false;                                 // ... as k-permuted bytes (0x40 = one), at the bf16 plane's row stride
    bool out_bytes = false;                               // the row-major plane of a 0/1 sample as bytes (0x40 = one)
    float* out_f32 = nullptr; float* prob_f32 = nullptr; float* out_u = nullptr; int ldo32 = 0;
    float* colpart = nullptr; int ld_colpart = 0;
    float colsign = 1.f;                                  // x3: colpart = colsign * column sums of the value plane
    int* grid_m_out = nullptr;
    float* rowpart = nullptr; int ld_rowpart = 0;         // x3, Bernoulli draws: softplus row sums of the tile's columns (the score)
    int* grid_n_out = nullptr;                            // ... and how many column tiles wrote them
};

// one half step: A [rows][lda] bf16 in a_pieces pieces (k padded), weights from the mirror
static int half_step_b(kurbm_ctx* ctx, int layout, const kurbm_params* p, const Mirror& m, const uint16_t* A, int lda,
                       int a_pieces, size_t a_plane, int rows, int act, int noise, const RngArgs* rng, const HalfOutB& o,
                       hipStream_t st, bool a_bytes = false) {
    GemmArgsB g;
    memset(&g, 0, sizeof g);
    const bool vh = (layout == LAYOUT_VH);
    g.A0 = A; g.lda = lda; g.a_plane0 = a_plane;
    g.B0 = vh ? m.Wtb : m.Wb; g.ldb = vh ? m.ldWt : m.ldW; g.b_plane0 = vh ? m.planeWt : m.planeW;
    // k extent: whole 64-deep k-tiles over the units (the planes and the mirror are zero-padded to 128 beyond them, so the
    // last tile reads zeros past n; 784 visibles are 13 tiles, not the 14 of the padded extent)
    g.M = rows; g.N = vh ? p->n_hid : p->n_vis; g.K = round_up(vh ? p->n_vis : p->n_hid, 64);
    {
        // x3: one A tile against the three pieces of the weight tile (kurbm_x3.hip); rounded bf16: against its one piece
        g.nseg = pb_codes(ctx, a_pieces, m.pieces, 0u, &g.seg_codes, 0);
        g.pb_max = m.pieces;
        // a real-valued A operand: its three segments per k position, so that they can share one staging of the B pieces
        // (k_gemm_pb, "BSH"; launch_gemm_pb checks the pattern)
        // (k_gemm_pb, "BSP"; launch_gemm_pb checks the pattern) on 128 x 128 tiles as TWO tiles per k position: the paired walk
        const bool pair = g.nseg == 3 && m.pieces == 3 && ctx->knob[KN_X3_PAIR] != 0;
        if (pair) { g.seg_fastest = 1; g.inv_nseg = inv_of(3); }
        g.pair_ok = pair ? 1 : 0;
        g.a_bytes = a_bytes ? 1 : 0;   // (a byte plane of 0/1 values: one piece, lda bytes between its rows)
        g.cfg = 0;
        // 2: 256 x 64 tiles (fewer bytes per k-tile).  Whole 256-row tiles only: the bias partial rows are laid out
        // per 64 rows, two per 128-row tile, and an even number of those is what both tilings agree on
        const int tall = ctx->knob[KN_X3_TALL];
        // (x3 only: with ONE piece per weight the B tile is the light operand and 128 x 128 tiles take fewer bytes per MFMA --
        //  A 16 KB + B 16 KB against 32 + 8 for the same 128 MFMAs per wave pair; 4096 x 4096 PCD-10, 1024 rows: 0.977 against
        //  1.033 ms per step)
        if (!g.cfg && !pair && ceil_div(rows, 128) % 2 == 0 &&
            (tall == 1 || (tall < 0 && m.pieces == 3 && (rows / 256) * ceil_div(g.N, 64) * 4 >= 3 * ctx->ncu))) g.cfg = 2;
        // (64-column tiles: a row-major plane is the next GEMM's A operand, k-padded to 128 -- cover the padded row)
        g.grid_m = ceil_div(rows, g.cfg == 2 ? 256 : 128);
        g.grid_n = g.cfg ? ceil_div(o.out ? round_up(g.N, 128) : g.N, 64) : ceil_div(g.N, 128);
        g.nkt = g.K / 64;
        g.inv_nkt = inv_of(g.nkt);
        g.kt_total = g.nseg * g.nkt; g.kt_per_split = g.kt_total; g.nsplit = 1;
        g.bias = vh ? p->b_h : p->b_v;
        g.act = act; g.noise = noise;
        if (rng) g.rng = *rng;
        g.out = o.out; g.ldo = o.ldo; g.ldo_cols = o.out ? o.ldo : g.N;
        g.out_pieces = o.out_pieces; g.out_plane = o.out_plane; g.out_bytes = o.out_bytes ? 1 : 0; g.out_rows_pad = o.out_rows_pad;
        g.outT = o.outT; g.ldoT = o.ldoT; g.outT_pieces = o.outT_pieces; g.outT_plane = o.outT_plane;
        g.outT_neg = o.outT_neg ? 1 : 0; g.outT_f8 = o.outT_b8 ? 2 : o.outT_f8 ? 1 : 0;
        g.out_f32 = o.out_f32; g.prob_f32 = o.prob_f32; g.out_u = o.out_u; g.ldo32 = o.ldo32;
        g.colpart = o.colpart; g.ld_colpart = o.ld_colpart; g.colsign = o.colsign;
        if (o.rowpart) {
            g.rowpart = o.rowpart; g.ld_rowpart = o.ld_rowpart; g.rp = 1;
            if (o.grid_n_out) *o.grid_n_out = ceil_div(g.N, g.cfg == 2 ? 64 : 128);   // (tiles in the column padding write zeros)
        }
        // row tiles fastest: an XCD's run of workgroups then shares ONE column tile, whose weight pieces
        // (the operand loaded straight into registers, one tile ahead) stay in that XCD's L2
        g.m_fastest = ctx->knob[KN_X3_MFAST];
        if (o.grid_m_out) *o.grid_m_out = ceil_div(rows, 128);   // (in 128-row units whatever the tile)
        g.xcd2d = ctx->knob[KN_X3_XCD2D]; g.map_force = ctx->knob[KN_MAP_SLOW];
        HIP_TRY(launch_gemm_pb(EPI_HALFSTEP, g, st));
        return KURBM_OK;
    }
}

static int mirror_refresh_any(kurbm_ctx* ctx, int pieces, const kurbm_params* p, void* mirror, size_t mirror_bytes,
                              kurbm_stream_t stream) {
    if (!ctx) return fail(KURBM_ERR_ARG, "ctx is null");
    if (int e = check_params(p)) return e;
    if (!mirror || !aligned16(mirror)) return fail(KURBM_ERR_ARG, "mirror is null or misaligned");
    const Mirror m = carve_mirror(ctx, mirror, p->n_vis, p->n_hid, pieces);
    if (m.bytes > mirror_bytes) return fail(KURBM_ERR_WORKSPACE, "mirror too small: need %zu bytes, got %zu", m.bytes, mirror_bytes);
    HIP_TRY(launch_f32_to_bf16(p->W, p->n_vis, p->n_hid, p->ldw, m.Wb, m.ldW, p->n_vis, m.Wtb, m.ldWt, p->n_hid, pieces,
                               m.planeW, m.planeWt, nullptr, 0, static_cast<hipStream_t>(stream)));
    return KURBM_OK;
}

static int half_step_any(kurbm_ctx* ctx, int pieces, int in_pieces, const kurbm_params* p, void* mirror, size_t mirror_bytes,
                         int dir, const float* in, int rows, int ld_in, int act, int noise, const kurbm_rng* rng,
                         float* out_sample, float* out_prob, float* out_u, int ld_out, void* workspace,
                         size_t workspace_bytes, kurbm_stream_t stream) {
    if (!ctx) return fail(KURBM_ERR_ARG, "ctx is null");
    if (int e = check_params(p)) return e;
    if (dir != 0 && dir != 1) return fail(KURBM_ERR_ARG, "dir must be 0 (v->h) or 1 (h->v)");
    if (in_pieces != 1 && in_pieces != 3) return fail(KURBM_ERR_ARG, "in_pieces must be 1 or 3");
    const int K = dir == 0 ? p->n_vis : p->n_hid, N = dir == 0 ? p->n_hid : p->n_vis;
    if (rows <= 0 || bad_matrix(in, ld_in, K)) return fail(KURBM_ERR_ARG, "input: null, misaligned, ld %% 4 != 0 or ld < columns");
    if (noise != NOISE_NONE && (!rng || (rng->row0 & 3))) return fail(KURBM_ERR_ARG, "rng missing or row0 not a multiple of 4");
    if (!out_sample && !out_prob) return fail(KURBM_ERR_ARG, "both outputs are null");
    if (ld_out % 4 != 0 || ld_out < N) return fail(KURBM_ERR_ARG, "ld_out %% 4 != 0 or ld_out < columns");
    if (!mirror || !workspace || !aligned16(mirror) || !aligned16(workspace)) return fail(KURBM_ERR_ARG, "mirror/workspace null or misaligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const Mirror m = carve_mirror(ctx, mirror, p->n_vis, p->n_hid, pieces);
    if (m.bytes > mirror_bytes) return fail(KURBM_ERR_WORKSPACE, "mirror too small");
    // the input plane is staged in the larger of the two row-major buffers' shapes: use a private carve
    const int Kp = ld_pad(ctx, round_up(K, 128)), Kb = round_up(rows, 128);   // Kp: leading dimension of the staged input
    const size_t plane = (size_t)Kb * Kp;
    const size_t need = align_up(in_pieces * plane * 2);
    if (need > workspace_bytes) return fail(KURBM_ERR_WORKSPACE, "workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
    uint16_t* Ab = static_cast<uint16_t*>(workspace);
    HIP_TRY(launch_f32_to_bf16(in, rows, K, ld_in, Ab, Kp, Kb, nullptr, 0, 0, in_pieces, plane, 0, nullptr, 0, st));
    RngArgs r;
    if (rng) r = make_rng(rng->seed, rng->row0, rng->stream_id, rng->step);
    HalfOutB o;
    o.out_f32 = (noise == NOISE_NONE) ? (out_prob ? out_prob : out_sample) : out_sample;
    o.prob_f32 = (noise == NOISE_NONE) ? nullptr : out_prob;
    o.out_u = out_u; o.ldo32 = ld_out;
    return half_step_b(ctx, dir == 0 ? LAYOUT_VH : LAYOUT_HV, p, m, Ab, Kp, in_pieces, plane, rows, act, noise,
                       rng ? &r : nullptr, o, st);
}

// Every argument check of a bf16 / x3 CD step, before anything is enqueued.  The data-parallel step runs it BEFORE its first
// collective too: a rank that fails here returns without having joined an all-reduce, and so must every other rank -- the
// checks depend only on arguments that are the same on all ranks (shapes, options, buffer sizes), except `rows`, which a rank
// may have none of (zero_rows_ok).
static int check_cd_args(kurbm_ctx* ctx, int pieces, int v_pieces, const kurbm_params* p, const void* mirror, size_t mirror_bytes,
                         const float* v_batch, int rows, int ldv, const kurbm_cd_opts* o, const void* workspace,
                         size_t workspace_bytes, bool zero_rows_ok) {
    if (!ctx || !o) return fail(KURBM_ERR_ARG, "null argument");
    if (int e = check_params(p)) return e;
    if (rows < 0 || (rows == 0 && !zero_rows_ok)) return fail(KURBM_ERR_ARG, "rows must be positive");
    if (rows > 0 && bad_matrix(v_batch, ldv, p->n_vis)) return fail(KURBM_ERR_ARG, "v_batch: null, misaligned, ld %% 4 != 0 or ld < n_vis");
    if (o->k < 1 || o->k > 15) return fail(KURBM_ERR_ARG, "k must be in [1, 15]");
    if (o->mode != KURBM_MODE_VISIBLE_BERNOULLI && o->mode != KURBM_MODE_VISIBLE_GAUSSIAN)
        return fail(KURBM_ERR_ARG, "unknown mode %d", o->mode);
    if (v_pieces == (1 | KURBM_V_BINARY)) v_pieces = 1;
    if (v_pieces != 1 && v_pieces != 3) return fail(KURBM_ERR_ARG, "v_pieces must be 1, 1 | KURBM_V_BINARY or 3");
    // (a rank without rows of a remainder batch passes the clipped start of its empty shard: nothing is drawn for it)
    if (rows > 0 && (o->row0 & 3)) return fail(KURBM_ERR_ARG, "row0 must be a multiple of 4");
    if (!mirror || !workspace || !aligned16(mirror) || !aligned16(workspace)) return fail(KURBM_ERR_ARG, "mirror/workspace null or misaligned");
    if (o->v_chain && !aligned16(o->v_chain)) return fail(KURBM_ERR_ARG, "v_chain is misaligned");
    if (o->v_planes && !aligned16(o->v_planes)) return fail(KURBM_ERR_ARG, "v_planes is misaligned");
    const size_t mb = carve_mirror(ctx, nullptr, p->n_vis, p->n_hid, pieces).bytes;
    if (mb > mirror_bytes) return fail(KURBM_ERR_WORKSPACE, "mirror too small: need %zu bytes, got %zu", mb, mirror_bytes);
    const size_t wb = carve_bf16(ctx, nullptr, rows > 0 ? rows : 1, p->n_vis, p->n_hid, pieces, v_pieces).bytes;
    if (wb > workspace_bytes) return fail(KURBM_ERR_WORKSPACE, "workspace too small: need %zu bytes, got %zu", wb, workspace_bytes);
    return KURBM_OK;
}

// (v_pieces without the KURBM_V_BINARY bit) the statistics of this step run as two launches: x3 on real-valued data, byte planes on
static bool split_stats_on(const kurbm_ctx* ctx, int pieces, int v_pieces) {
    return pieces == 3 && v_pieces == 3 && ctx->knob[KN_X3_SPLIT_STATS] != 0 && ctx->knob[KN_X3_BYTES] != 0;
}

// ... and the negative statistics of Gaussian visibles read v_neg's ROW-MAJOR pieces (no transposed copy is written): needs the
// paired walk on 128 x 128 tiles, whose LDS rows are then the plane's 128-column blocks (n_vis padded to 128 = the plane's ld)
static bool atr_on(const kurbm_ctx* ctx, int pieces, int v_pieces, bool gauss) {
    return gauss && split_stats_on(ctx, pieces, v_pieces) && ctx->knob[KN_X3_ATR] != 0 && ctx->knob[KN_X3_PAIR] != 0 && ctx->knob[KN_LDPAD] == 0;
}

static int cd_step_any(kurbm_ctx* ctx, int pieces, int v_pieces, const kurbm_params* p, void* mirror, size_t mirror_bytes,
                       const float* v_batch, int rows, int ldv, const kurbm_cd_opts* o, int which, void* workspace,
                       size_t workspace_bytes, kurbm_stream_t stream, int only = -1, int m_lo = 0, int m_hi = -1) {
    // `only` 0..6 (measurement hook kurbm_cd_step_x3_stage): launch just that stage of the sequence, on the planes a
    // previous complete step left in the workspace.  8: the chain alone (stages 0-3, kurbm_cd_chain_x3).  7: the
    // statistics of visible rows [m_lo, m_hi) alone (stages 4-5, kurbm_x3_stats_rows) -- the data-parallel step
    // all-reduces the first rows of dW while the rest is still being computed.
#define KURBM_STAGE(n) (only < 0 || only == (n) || (only == 8 && (n) <= 3) || (only == 7 && ((n) == 4 || (n) == 5)))
    if (m_hi < 0) m_hi = p ? p->n_vis : 0;
    if (int e = check_cd_args(ctx, pieces, v_pieces, p, mirror, mirror_bytes, v_batch, rows, ldv, o, workspace, workspace_bytes, false)) return e;
    const bool v_binary = (v_pieces == (1 | KURBM_V_BINARY));
    if (v_binary) v_pieces = 1;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const Mirror m = carve_mirror(ctx, mirror, p->n_vis, p->n_hid, pieces);
    WorkspaceB w = carve_bf16(ctx, workspace, rows, p->n_vis, p->n_hid, pieces, v_pieces);

    const bool gauss = (o->mode == KURBM_MODE_VISIBLE_GAUSSIAN);
    const int act_h = gauss ? ACT_RELU : ACT_SIGMOID;
    const int act_v = gauss ? ACT_LINEAR : ACT_SIGMOID;
    const int noise_v = gauss ? NOISE_GAUSSIAN : NOISE_BERNOULLI;
    const uint32_t base = o->chain * 64u;
    const bool need_w = (which & 1) || o->delta_out;
    // x3 with Gaussian visibles: v_t / v_neg are real-valued, so they travel as three pieces like real-valued data
    const int vn_pieces = (pieces == 3 && gauss) ? 3 : 1;
    // 0/1 data on the x3 path: v_pos^T and h_pos^T leave as fp8 planes and their product runs on the fp8 matrix cores
    const bool f8pos = v_binary && pieces == 3 && ctx->knob[KN_X3_F8POS] != 0;
    // ... and every row-major plane of 0/1 values is a BYTE plane (half the A tile of the half step that reads it): the
    // hidden samples always, v_pos / the persistent chain for 0/1 data, the negative visibles in Bernoulli mode
    // (the rounded-bf16 path too: its 0/1 states are exact as bytes, so nothing changes but the bytes the half steps stage)
    const bool byt = ctx->knob[KN_X3_BYTES] != 0;
    const bool hbytes = byt, vbytes = byt && v_binary && pieces == 3, nbytes = byt && !gauss;
    // ... and v_neg^T, the A operand of the statistics GEMM's negative half, where the positive half runs on fp8 planes (the
    // walk is then whole units of fp8 / 3-piece / 3-piece tiles: k_gemm_pb<..., EPI_SLAB, ..., AB>): 32 KB per 128 k, not 64
    const bool tbytes = f8pos && nbytes && ctx->knob[KN_X3_STATS_BYTES] != 0 && ctx->knob[KN_X3_STATS_TALL] != 0 && p->n_vis > 128;
    // real-valued data: the statistics as two launches (plan_split_stats), h_pos^T -- and v_neg^T of Bernoulli visibles -- as byte planes
    const bool split_stats = split_stats_on(ctx, pieces, v_pieces);
    const bool tbytes_n = tbytes || (split_stats && nbytes);
    // ... and Gaussian visibles need no transposed copy of v_neg at all: the paired walk of the negative statistics reads the row-major
    // pieces through transposed LDS reads (k_gemm_pb, "ATR")
    const bool vneg_tr = atr_on(ctx, pieces, v_pieces, gauss);
    int e;

    // v_pos -> bf16 pieces, row-major (A of the v->h step) and transposed (statistics)
    // (x3: plus the column sums of v_pos per 64-row band, the positive half of the visible-bias statistics)
    const int gp_v = ceil_div(rows, 64);
    // resident planes (kurbm_x3_convert_rows ran once for these rows): no conversion; the positive column sums stay where
    // they are and the slab reducer adds them in front of the negative ones (same order of additions either way)
    const float* part_v_pos = nullptr;
    if (o->v_planes) {
        if (!aligned16(o->v_planes)) return fail(KURBM_ERR_ARG, "v_planes is misaligned");
        const VPlanes vp = carve_vplanes(ctx, o->v_planes, rows, p->n_vis, v_pieces);
        w.vb = vp.vb; w.vbT = vp.vbT; part_v_pos = vp.part_v;
    } else if (KURBM_STAGE(0))
        HIP_TRY(launch_f32_to_bf16(v_batch, rows, p->n_vis, ldv, w.vb, w.Lv, w.Kb, w.vbT, w.Lb, p->n_vis, v_pieces, w.planeV,
                                   w.planeVT, w.part_v, w.ldv32, st, f8pos ? 1 : 0, vbytes ? 1 : 0));
    // h_pos ~ p(h | v_pos)                                          rbm.py:120
    RngArgs r = make_rng(o->seed, o->row0, base + 0u, o->step);
    if (KURBM_STAGE(1)) {
        HalfOutB ho;
        ho.out = w.hb; ho.ldo = w.Lh; ho.out_bytes = hbytes; ho.outT = w.hbT; ho.ldoT = w.Lb; ho.outT_f8 = f8pos; ho.outT_b8 = split_stats;
        ho.colpart = w.part_h; ho.ld_colpart = w.ldh32; ho.colsign = 1.f;
        if ((e = half_step_b(ctx, LAYOUT_VH, p, m, w.vb, w.Lv, v_pieces, w.planeV, rows, act_h, NOISE_BERNOULLI, &r, ho, st, vbytes))) return e;
    }
    const uint16_t* h_cur = w.hb;
    if (o->v_chain && KURBM_STAGE(1)) {   // persistent chain: negative phase starts from the stored fantasy particles
        HIP_TRY(launch_f32_to_bf16(o->v_chain, rows, p->n_vis, ldv, w.cb, w.Lv, w.Kb, nullptr, 0, 0, v_pieces, w.planeV, 0, nullptr, 0, st,
                                   0, vbytes ? 1 : 0));
        r = make_rng(o->seed, o->row0, base + 32u, o->step);
        HalfOutB ho;
        ho.out = w.h2b; ho.ldo = w.Lh; ho.out_bytes = hbytes;
        if ((e = half_step_b(ctx, LAYOUT_VH, p, m, w.cb, w.Lv, v_pieces, w.planeV, rows, act_h, NOISE_BERNOULLI, &r, ho, st, vbytes))) return e;
        h_cur = w.h2b;
    }
    int gm_v = ceil_div(rows, 128), gm_h = gm_v;   // row tiles of the half steps (bias partial rows)
    for (int t = 1; t <= o->k; ++t) {
        const bool last = (t == o->k);
        r = make_rng(o->seed, o->row0, base + 2u * t - 1u, o->step);      // v_t ~ p(v | h_{t-1})   rbm.py:121-123
        if (KURBM_STAGE(2)) {
            HalfOutB ho;
            ho.out = w.v2b; ho.ldo = w.Lv; ho.out_pieces = vn_pieces; ho.out_plane = w.planeV; ho.out_bytes = nbytes;
            if (last) {
                if (vneg_tr) ho.out_rows_pad = w.Kb;      // (the statistics walk the batch padded to 128 rows of this plane)
                if (!vneg_tr) { ho.outT = w.v2bT; ho.ldoT = w.Lb; ho.outT_pieces = vn_pieces; ho.outT_plane = w.planeVT; ho.outT_b8 = tbytes_n; }
                ho.out_f32 = o->v_chain; ho.ldo32 = ldv;
                ho.grid_m_out = &gm_v;
            }
            // -sum(v_neg) below the bands of +sum(v_pos); nothing for the inner steps of CD-k
            ho.colpart = last ? w.part_v + (size_t)gp_v * w.ldv32 : nullptr; ho.ld_colpart = w.ldv32;
            ho.colsign = -1.f;
            if ((e = half_step_b(ctx, LAYOUT_HV, p, m, h_cur, w.Lh, 1, 0, rows, act_v, noise_v, &r, ho, st, hbytes))) return e;
        }
        if (!last && KURBM_STAGE(2)) {
            r = make_rng(o->seed, o->row0, base + 2u * t, o->step);
            HalfOutB ho;
            ho.out = w.h2b; ho.ldo = w.Lh; ho.out_bytes = hbytes;
            if ((e = half_step_b(ctx, LAYOUT_VH, p, m, w.v2b, w.Lv, vn_pieces, w.planeV, rows, act_h, NOISE_BERNOULLI, &r, ho, st, nbytes))) return e;
            h_cur = w.h2b;
        }
    }
    // dW = v_pos^T.h_pos - v_neg^T.h_neg: NT GEMM over the transposed images, k = batch;
    // segments: (piece of v_pos) x h_pos, then v_neg x (piece of -h_neg)
    const int nseg_st = pieces == 3 ? v_pieces + vn_pieces : 2;
    const OuterPlanB plf = plan_outer_bf16(ctx, rows, p->n_vis, p->n_hid, nseg_st, 1 << 30, f8pos, pieces == 3);
    const int Mr = m_hi - m_lo;                         // visible rows of this call (all of them unless only == 7)
    const bool sub = (Mr != p->n_vis);
    // a row range keeps inside the slab memory carved for the whole matrix
    const OuterPlanB pl = sub ? plan_outer_bf16(ctx, rows, Mr, p->n_hid, nseg_st,
                                                (int)((w.slab_stride * plf.nsplit_bound) / ((size_t)Mr * plf.ld_slab)), f8pos, pieces == 3)
                              : plf;
    const size_t slab_stride = sub ? (size_t)Mr * pl.ld_slab : w.slab_stride;
    const bool ap = o->apply != 0;
    // (a row range [m_lo, m_hi) of the visible units: each half may use half of the slabs carved for the whole matrix)
    const SplitStats sps0 = plan_split_stats(ctx, rows, p->n_vis, p->n_hid, gauss);
    const SplitStats sps = sub ? plan_split_stats(ctx, rows, Mr, p->n_hid, gauss,
                                                  (int)((w.slab_stride * (size_t)(sps0.neg.nsplit_bound + sps0.pos.nsplit_bound)) / ((size_t)Mr * sps0.pos.ld_slab) / 2))
                               : sps0;
    // h_neg = sigmoid(v_neg.W + b_h), probabilities (rbm.py:124): only its transposed image is needed, and only by the
    // statistics GEMM, where it enters with a minus sign: it is stored as -h_neg
    if (KURBM_STAGE(3)) {
        HalfOutB ho;
        ho.outT = w.hnT; ho.ldoT = w.Lb; ho.outT_pieces = pieces; ho.outT_plane = w.planeHT; ho.outT_neg = true;
        ho.colpart = w.part_h + (size_t)2 * ceil_div(rows, 128) * w.ldh32; ho.ld_colpart = w.ldh32; ho.colsign = -1.f;
        ho.grid_m_out = &gm_h;
        if ((e = half_step_b(ctx, LAYOUT_VH, p, m, w.v2b, w.Lv, vn_pieces, w.planeV, rows, ACT_SIGMOID, NOISE_NONE, nullptr, ho, st, nbytes))) return e;
    }

    int nslab_used = split_stats ? sps.neg.nsplit + sps.pos.nsplit : pl.nsplit;
    ReduceArgs a;
    memset(&a, 0, sizeof a);
    a.slab = w.slab; a.slab_stride = slab_stride; a.ld_slab = pl.ld_slab;
    a.nslab = nslab_used;
    a.n_vis = Mr; a.n_hid = p->n_hid; a.ldw = p->ldw;
    a.nblk_w = need_w ? (int)(((long long)Mr * (pl.ld_slab / 4) + 255) / 256) : 0;
    a.lr = o->lr;
    a.W = (ap && (which & 1)) ? p->W : nullptr;
    a.delta_w = o->delta_out ? o->delta_out + (size_t)m_lo * p->n_hid : nullptr;
    a.part_h = w.part_h; a.nrow_tiles_h = 4 * gm_h; a.ld_part_h = w.ldh32;
    a.b_h = (ap && (which & 2)) ? p->b_h : nullptr;
    a.delta_bh = o->delta_out ? o->delta_out + (size_t)p->n_vis * p->n_hid : nullptr;
    a.part_v = w.part_v; a.nrow_tiles_v = gp_v + 2 * gm_v; a.ld_part_v = w.ldv32;
    if (part_v_pos) {   // rows [0, gp_v) from the resident planes, then the workspace's negative rows
        a.part_v = part_v_pos; a.nrow_tiles_v = gp_v;
        a.part_v2 = w.part_v + (size_t)gp_v * w.ldv32; a.nrow_tiles_v2 = 2 * gm_v;
    }
    a.b_v = (ap && (which & 4)) ? p->b_v : nullptr;
    a.delta_bv = o->delta_out ? o->delta_out + (size_t)p->n_vis * p->n_hid + p->n_hid : nullptr;
    if (sub && m_hi != p->n_vis) { a.part_h = nullptr; a.part_v = nullptr; a.part_v2 = nullptr; }   // the bias sums leave with the LAST rows
    if (sub) a.n_vis_bias = p->n_vis;
    const bool mirror_in_reduce = a.W && need_w && !ctx->knob[KN_UNFUSED_MIRROR];
    if (mirror_in_reduce) {
        // the fp32 master moves: the slab reduction writes the new weights AND their bf16 pieces in one pass
        a.Wb = m.Wb; a.ldWb = m.ldW; a.planeWb = m.planeW;
        a.Wtb = m.Wtb; a.ldWtb = m.ldWt; a.planeWtb = m.planeWt; a.pieces = pieces;
        a.tile_rows = ctx->knob[KN_REDUCE_TR];
    }
    if (need_w && split_stats && KURBM_STAGE(4)) {
        // ---- negative half: slabs [0, neg.nsplit)
        {
            const OuterPlanB& q = sps.neg;
            GemmArgsB g;
            memset(&g, 0, sizeof g);
            g.A0 = w.v2bT + (size_t)m_lo * w.Lb; g.a_plane0 = w.planeVT; g.B0 = w.hnT; g.b_plane0 = w.planeHT;
            g.lda = w.Lb; g.ldb = w.Lb;
            g.M = Mr; g.N = p->n_hid; g.K = w.Kb;
            g.grid_m = q.gm; g.grid_n = q.gn; g.cfg = q.cfg;
            g.slab = w.slab; g.slab_stride = slab_stride; g.ld_slab = q.ld_slab;
            g.pb_max = 3;
            if (gauss) {   // (v_neg piece a) x (-h_neg pieces 0 .. 2 - a), two tiles per k position (k_gemm_pb, "BSP")
                g.nseg = pb_codes(ctx, 3, 3, 0u, &g.seg_codes, 0);
                g.seg_fastest = 1; g.inv_nseg = inv_of(g.nseg); g.pair_ok = 1;
                if (vneg_tr) {   // A = the ROW-MAJOR pieces [batch][n_vis]: columns m_lo .. of them
                    g.A0 = w.v2b + m_lo; g.a_plane0 = w.planeV; g.lda = w.Lv; g.a_tr = 1;
                }
            } else {       // v_neg^T as bytes against the three pieces of -h_neg^T (k_gemm_pb, "ABP")
                g.nseg = pb_codes(ctx, 1, 3, 0u, &g.seg_codes, 0);
                g.a_bytes = 1; g.lda = 2 * w.Lb;
            }
            g.m_fastest = ctx->knob[KN_X3_STATS_MFAST];
            g.nkt = q.nkt; g.inv_nkt = inv_of(g.nkt);
            g.kt_total = q.kt_total; g.kt_per_split = q.kt_per_split; g.nsplit = q.nsplit;
            g.xcd2d = ctx->knob[KN_X3_XCD2D]; g.map_force = ctx->knob[KN_MAP_SLOW];
            HIP_TRY(launch_gemm_pb(EPI_SLAB, g, st));
        }
        // ---- positive half, transposed: h_pos^T (bytes) x the pieces of v_pos^T -> slabs [neg.nsplit, +pos.nsplit)
        {
            const OuterPlanB& q = sps.pos;
            GemmArgsB g;
            memset(&g, 0, sizeof g);
            g.A0 = w.hbT; g.a_bytes = 1; g.lda = 2 * w.Lb;
            g.B0 = w.vbT + (size_t)m_lo * w.Lb; g.b_plane0 = w.planeVT; g.ldb = w.Lb;
            g.M = p->n_hid; g.N = Mr; g.K = w.Kb;
            g.grid_m = q.gm; g.grid_n = q.gn; g.cfg = q.cfg;
            g.slab = w.slab + (size_t)sps.neg.nsplit * slab_stride; g.slab_stride = slab_stride; g.ld_slab = q.ld_slab; g.slab_t = 1;
            g.pb_max = 3;
            g.nseg = pb_codes(ctx, 1, 3, 0u, &g.seg_codes, 0);
            g.m_fastest = ctx->knob[KN_X3_STATS_MFAST];
            g.nkt = q.nkt; g.inv_nkt = inv_of(g.nkt);
            g.kt_total = q.kt_total; g.kt_per_split = q.kt_per_split; g.nsplit = q.nsplit;
            g.xcd2d = ctx->knob[KN_X3_XCD2D]; g.map_force = ctx->knob[KN_MAP_SLOW];
            HIP_TRY(launch_gemm_pb(EPI_SLAB, g, st));
        }
    } else if (need_w && KURBM_STAGE(4)) {
        GemmArgsB g;
        memset(&g, 0, sizeof g);
        g.A0 = w.vbT + (size_t)m_lo * w.Lb; g.a_plane0 = w.planeVT; g.B0 = w.hbT;
        g.A1 = w.v2bT + (size_t)m_lo * w.Lb; g.a_plane1 = w.planeVT; g.B1 = w.hnT; g.b_plane1 = w.planeHT;
        g.lda = w.Lb; g.ldb = w.Lb;
        if (tbytes) { g.a_bytes = 1; g.lda = 2 * w.Lb; }   // (both A planes hold one byte per element at the bf16 planes' row stride)
        g.M = Mr; g.N = p->n_hid; g.K = w.Kb;
        g.grid_m = pl.gm; g.grid_n = pl.gn;
        g.slab = w.slab; g.slab_stride = slab_stride; g.ld_slab = pl.ld_slab;
        // (piece of v_pos) x h_pos, then v_neg x (all pieces of h_neg), walked segment-fastest
        g.nseg = pb_codes(ctx, v_pieces, 1, 0u, &g.seg_codes, 0);
        g.nseg = pb_codes(ctx, vn_pieces, pieces, 1u, &g.seg_codes, g.nseg);
        g.pb_max = pieces;
        g.seg_fastest = 1; g.inv_nseg = inv_of(g.nseg);
        if (f8pos) { g.f8pos = 1; g.inv_nseg = inv_of(2 * g.nseg - 1); }   // (tiles per 128-deep unit)
        g.m_fastest = ctx->knob[KN_X3_STATS_MFAST];
        g.cfg = pl.cfg;
        g.nkt = pl.nkt; g.inv_nkt = inv_of(g.nkt);
        g.kt_total = pl.kt_total; g.kt_per_split = pl.kt_per_split; g.nsplit = pl.nsplit;
        g.xcd2d = ctx->knob[KN_X3_XCD2D]; g.map_force = ctx->knob[KN_MAP_SLOW];
        HIP_TRY(launch_gemm_pb(EPI_SLAB, g, st));
    }
    if (mirror_in_reduce) {
        if (KURBM_STAGE(5)) HIP_TRY(launch_reduce_apply_split(a, st));
        if (only == 6)
            HIP_TRY(launch_f32_to_bf16(p->W, p->n_vis, p->n_hid, p->ldw, m.Wb, m.ldW, p->n_vis, m.Wtb, m.ldWt, p->n_hid, pieces,
                                       m.planeW, m.planeWt, nullptr, 0, st));
    } else {
        if (KURBM_STAGE(5)) HIP_TRY(launch_reduce_apply(a, st));
        if (a.W && (KURBM_STAGE(5) || only == 6))   // the fp32 master moved: re-derive both mirrors
            HIP_TRY(launch_f32_to_bf16(p->W, p->n_vis, p->n_hid, p->ldw, m.Wb, m.ldW, p->n_vis, m.Wtb, m.ldWt, p->n_hid, pieces,
                                       m.planeW, m.planeWt, nullptr, 0, st));
    }
#undef KURBM_STAGE
    return KURBM_OK;
}

extern "C" {

size_t kurbm_bf16_mirror_bytes(kurbm_ctx* ctx, int n_vis, int n_hid) {
    if (!ctx || n_vis <= 0 || n_hid <= 0) return 0;
    return carve_mirror(ctx, nullptr, n_vis, n_hid, 1).bytes;
}
size_t kurbm_x3_mirror_bytes(kurbm_ctx* ctx, int n_vis, int n_hid) {
    if (!ctx || n_vis <= 0 || n_hid <= 0) return 0;
    return carve_mirror(ctx, nullptr, n_vis, n_hid, 3).bytes;
}

int kurbm_bf16_mirror_refresh(kurbm_ctx* ctx, const kurbm_params* p, void* mirror, size_t mirror_bytes, kurbm_stream_t stream) {
    return mirror_refresh_any(ctx, 1, p, mirror, mirror_bytes, stream);
}
int kurbm_x3_mirror_refresh(kurbm_ctx* ctx, const kurbm_params* p, void* mirror, size_t mirror_bytes, kurbm_stream_t stream) {
    return mirror_refresh_any(ctx, 3, p, mirror, mirror_bytes, stream);
}

size_t kurbm_bf16_workspace_bytes(kurbm_ctx* ctx, int rows, int n_vis, int n_hid, int k) {
    (void)k;
    if (!ctx || rows <= 0 || n_vis <= 0 || n_hid <= 0) return 0;
    return carve_bf16(ctx, nullptr, rows, n_vis, n_hid, 1, 1).bytes;
}
size_t kurbm_x3_workspace_bytes(kurbm_ctx* ctx, int rows, int n_vis, int n_hid, int k, int v_pieces) {
    (void)k;
    if (v_pieces == (1 | KURBM_V_BINARY)) v_pieces = 1;
    if (!ctx || rows <= 0 || n_vis <= 0 || n_hid <= 0 || (v_pieces != 1 && v_pieces != 3)) return 0;
    return carve_bf16(ctx, nullptr, rows, n_vis, n_hid, 3, v_pieces).bytes;
}

int kurbm_half_step_bf16(kurbm_ctx* ctx, const kurbm_params* p, void* mirror, size_t mirror_bytes, int dir,
                         const float* in, int rows, int ld_in, int act, int noise, const kurbm_rng* rng,
                         float* out_sample, float* out_prob, float* out_u, int ld_out, void* workspace,
                         size_t workspace_bytes, kurbm_stream_t stream) {
    return half_step_any(ctx, 1, 1, p, mirror, mirror_bytes, dir, in, rows, ld_in, act, noise, rng, out_sample, out_prob,
                         out_u, ld_out, workspace, workspace_bytes, stream);
}
int kurbm_half_step_x3(kurbm_ctx* ctx, const kurbm_params* p, void* mirror, size_t mirror_bytes, int dir,
                       const float* in, int in_pieces, int rows, int ld_in, int act, int noise, const kurbm_rng* rng,
                       float* out_sample, float* out_prob, float* out_u, int ld_out, void* workspace,
                       size_t workspace_bytes, kurbm_stream_t stream) {
    return half_step_any(ctx, 3, in_pieces, p, mirror, mirror_bytes, dir, in, rows, ld_in, act, noise, rng, out_sample,
                         out_prob, out_u, ld_out, workspace, workspace_bytes, stream);
}

int kurbm_cd_step_bf16(kurbm_ctx* ctx, const kurbm_params* p, void* mirror, size_t mirror_bytes, const float* v_batch,
                       int rows, int ldv, const kurbm_cd_opts* o, int which, void* workspace, size_t workspace_bytes,
                       kurbm_stream_t stream) {
    return cd_step_any(ctx, 1, 1, p, mirror, mirror_bytes, v_batch, rows, ldv, o, which, workspace, workspace_bytes, stream);
}
int kurbm_cd_step_x3(kurbm_ctx* ctx, const kurbm_params* p, void* mirror, size_t mirror_bytes, const float* v_batch,
                     int v_pieces, int rows, int ldv, const kurbm_cd_opts* o, int which, void* workspace,
                     size_t workspace_bytes, kurbm_stream_t stream) {
    return cd_step_any(ctx, 3, v_pieces, p, mirror, mirror_bytes, v_batch, rows, ldv, o, which, workspace, workspace_bytes, stream);
}

int kurbm_cd_step_x3_stage(kurbm_ctx* ctx, const kurbm_params* p, void* mirror, size_t mirror_bytes, const float* v_batch,
                           int v_pieces, int rows, int ldv, const kurbm_cd_opts* o, int which, int stage, void* workspace,
                           size_t workspace_bytes, kurbm_stream_t stream) {
    if (stage < 0 || stage > 6) return fail(KURBM_ERR_ARG, "stage must be in [0, 6]");
    return cd_step_any(ctx, 3, v_pieces, p, mirror, mirror_bytes, v_batch, rows, ldv, o, which, workspace, workspace_bytes, stream,
                       stage);
}

int kurbm_free_energy_x3(kurbm_ctx* ctx, const kurbm_params* p, void* mirror, size_t mirror_bytes, const float* v, int v_pieces,
                         int rows, int ldv, float* F, void* workspace, size_t workspace_bytes, kurbm_stream_t stream) {
    if (!ctx || !F) return fail(KURBM_ERR_ARG, "null argument");
    if (int e = check_params(p)) return e;
    if (rows <= 0) return fail(KURBM_ERR_ARG, "rows must be positive");
    if (v_pieces == (1 | KURBM_V_BINARY)) v_pieces = 1;
    if (v_pieces != 1 && v_pieces != 3) return fail(KURBM_ERR_ARG, "v_pieces must be 1 or 3");
    if (bad_matrix(v, ldv, p->n_vis)) return fail(KURBM_ERR_ARG, "v: null, misaligned, ld %% 4 != 0 or ld < n_vis");
    if (!mirror || !workspace || !aligned16(mirror) || !aligned16(workspace)) return fail(KURBM_ERR_ARG, "mirror/workspace null or misaligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const Mirror m = carve_mirror(ctx, mirror, p->n_vis, p->n_hid, 3);
    if (m.bytes > mirror_bytes) return fail(KURBM_ERR_WORKSPACE, "mirror too small");
    // workspace: the bf16 pieces of v [Kb][Lp], then the row partials [column tiles][rows]
    const int Lp = ld_pad(ctx, m.Kv), Kb = round_up(rows, 128);
    const size_t plane = (size_t)Kb * Lp;
    const size_t a_bytes = align_up(v_pieces * plane * 2);
    GemmArgsB g;
    memset(&g, 0, sizeof g);
    g.grid_m = ceil_div(rows, 128); g.grid_n = ceil_div(p->n_hid, 128);
    g.ld_rowpart = round_up(rows, 4);
    const size_t need = a_bytes + (size_t)g.grid_n * g.ld_rowpart * 4;
    if (need > workspace_bytes) return fail(KURBM_ERR_WORKSPACE, "workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
    uint16_t* Ab = static_cast<uint16_t*>(workspace);
    HIP_TRY(launch_f32_to_bf16(v, rows, p->n_vis, ldv, Ab, Lp, Kb, nullptr, 0, 0, v_pieces, plane, 0, nullptr, 0, st));
    g.A0 = Ab; g.lda = Lp; g.a_plane0 = plane;
    g.B0 = m.Wtb; g.ldb = m.ldWt; g.b_plane0 = m.planeWt;
    g.M = rows; g.N = p->n_hid; g.K = m.Kv;
    g.nseg = pb_codes(ctx, v_pieces, 3, 0u, &g.seg_codes, 0);
    g.nkt = g.K / 64; g.inv_nkt = inv_of(g.nkt);
    g.kt_total = g.nseg * g.nkt; g.kt_per_split = g.kt_total; g.nsplit = 1;
    g.m_fastest = 1;
    g.bias = p->b_h;
    g.rowpart = reinterpret_cast<float*>(static_cast<char*>(workspace) + a_bytes);
    g.xcd2d = ctx->knob[KN_X3_XCD2D]; g.map_force = ctx->knob[KN_MAP_SLOW];
    HIP_TRY(launch_gemm_pb(EPI_SOFTPLUS, g, st));
    FinishArgs f;
    f.v = v; f.b_v = p->b_v; f.rowpart = g.rowpart; f.F = F;
    f.rows = rows; f.n_vis = p->n_vis; f.ldv = ldv; f.ncol_tiles = g.grid_n; f.ld_rowpart = g.ld_rowpart;
    HIP_TRY(launch_free_energy_finish(f, st));
    return KURBM_OK;
}

int kurbm_score_x3(kurbm_ctx* ctx, const kurbm_params* p, void* mirror, size_t mirror_bytes, const float* v_batch, int v_pieces,
                   int rows, int ldv, const kurbm_cd_opts* o, float* score, float* F, void* workspace, size_t workspace_bytes,
                   kurbm_stream_t stream) {
    // rbm.py:225-233 in one call, nothing returned to the host: h ~ p(h | v) [half step, which also leaves the softplus row sums
    // of F(v): same accumulators], v' ~ p(v | h) [half step; its row-major plane feeds the next GEMM, an fp32 copy the v'.b_v
    // term], F(v') [GEMM + softplus row sums], then |F - F'| per row and its mean into *score.  Draws: chain o->chain, sites 0 (h) and 1 (v'), step o->step -- the
    // counters RBM._score has always used.
    if (!score || !aligned16(score)) return fail(KURBM_ERR_ARG, "score is null or misaligned");
    if (int e = check_cd_args(ctx, 3, v_pieces, p, mirror, mirror_bytes, v_batch, rows, ldv, o, workspace, workspace_bytes, false)) return e;
    const bool v_binary = (v_pieces == (1 | KURBM_V_BINARY));
    if (v_binary) v_pieces = 1;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const Mirror m = carve_mirror(ctx, mirror, p->n_vis, p->n_hid, 3);
    WorkspaceB w = carve_bf16(ctx, workspace, rows, p->n_vis, p->n_hid, 3, v_pieces);
    const bool gauss = (o->mode == KURBM_MODE_VISIBLE_GAUSSIAN);
    const bool byt = ctx->knob[KN_X3_BYTES] != 0;
    const bool vbytes = byt && v_binary, hbytes = byt, nbytes = byt && !gauss;
    const bool f8pos = v_binary && ctx->knob[KN_X3_F8POS] != 0;      // (only so that a conversion here writes what a step would)
    const int vn_pieces = gauss ? 3 : 1;
    const uint32_t base = o->chain * 64u;
    // row partials of the two free energies (one row per 64- or 128-column tile), then |F - F'| per row
    const int ncol = ceil_div(p->n_hid, 128), ncol_max = ceil_div(p->n_hid, 64), ld_rp = round_up(rows, 4);
    float* rp0 = w.rowpart; float* rp1 = rp0 + (size_t)ncol_max * ld_rp; float* absdiff = rp1 + (size_t)ncol_max * ld_rp;
    if (!aligned16(p->b_v)) return fail(KURBM_ERR_ARG, "b_v must be 16-byte aligned for the score");
    if (o->v_planes) {
        const VPlanes vp = carve_vplanes(ctx, o->v_planes, rows, p->n_vis, v_pieces);
        w.vb = vp.vb;
    } else {
        HIP_TRY(launch_f32_to_bf16(v_batch, rows, p->n_vis, ldv, w.vb, w.Lv, w.Kb, nullptr, 0, 0, v_pieces, w.planeV, 0, nullptr, 0, st,
                                   0, vbytes ? 1 : 0));
    }
    (void)f8pos;
    int e;
    int ncol0 = ncol;
    RngArgs r = make_rng(o->seed, o->row0, base + 0u, o->step);
    {
        // h ~ p(h | v) AND the softplus row sums of F(v), from the same accumulators: one GEMM for rbm.py:227 and :230
        HalfOutB ho;
        ho.out = w.hb; ho.ldo = w.Lh; ho.out_bytes = hbytes;
        ho.rowpart = rp0; ho.ld_rowpart = ld_rp; ho.grid_n_out = &ncol0;
        if ((e = half_step_b(ctx, LAYOUT_VH, p, m, w.vb, w.Lv, v_pieces, w.planeV, rows, gauss ? ACT_RELU : ACT_SIGMOID, NOISE_BERNOULLI,
                             &r, ho, st, vbytes))) return e;                                            // h          rbm.py:230
    }
    r = make_rng(o->seed, o->row0, base + 1u, o->step);
    {
        HalfOutB ho;
        ho.out = w.v2b; ho.ldo = w.Lv; ho.out_pieces = vn_pieces; ho.out_plane = w.planeV; ho.out_bytes = nbytes;
        // (a 0/1 reconstruction: the score kernel takes v'.b_v from the byte plane; real-valued v' also leaves as fp32)
        if (!nbytes) { ho.out_f32 = w.tmp32; ho.ldo32 = w.ldv32; }
        if ((e = half_step_b(ctx, LAYOUT_HV, p, m, w.hb, w.Lh, 1, 0, rows, gauss ? ACT_LINEAR : ACT_SIGMOID,
                             gauss ? NOISE_GAUSSIAN : NOISE_BERNOULLI, &r, ho, st, hbytes))) return e;   // v'
    }
    int ncol1 = ncol;
    {
        // F(v'): the same GEMM as a half step that draws nothing and stores nothing but its softplus row sums     rbm.py:231
        HalfOutB ho;
        ho.rowpart = rp1; ho.ld_rowpart = ld_rp; ho.grid_n_out = &ncol1;
        if ((e = half_step_b(ctx, LAYOUT_VH, p, m, w.v2b, w.Lv, vn_pieces, w.planeV, rows, ACT_SIGMOID, NOISE_NONE, nullptr, ho, st, nbytes))) return e;
    }
    ScoreArgs a;
    memset(&a, 0, sizeof a);
    if (nbytes) { a.v1b = reinterpret_cast<const unsigned char*>(w.v2b); a.ldv1b = w.Lv; }
    a.v = v_batch; a.v1 = w.tmp32; a.b_v = p->b_v; a.rowpart = rp0; a.rowpart1 = rp1; a.F = F; a.absdiff = absdiff; a.score = score;
    a.rows = rows; a.n_vis = p->n_vis; a.ldv = ldv; a.ldv1 = w.ldv32; a.ncol_tiles = ncol0; a.ncol_tiles1 = ncol1; a.ld_rowpart = ld_rp;
    HIP_TRY(launch_score(a, st));                                                                      // mean |F - F'|  rbm.py:233
    return KURBM_OK;
}

int kurbm_x3_dump_plane(kurbm_ctx* ctx, int which, int rows, int n_vis, int n_hid, int v_pieces, int mode, const void* workspace,
                        size_t workspace_bytes, float* out, int ld_out, kurbm_stream_t stream) {
    // the planes the last complete kurbm_cd_step_x3 on (rows, v_pieces, mode) left in `workspace`, decoded to fp32 [rows][units]
    if (!ctx || !workspace || !out) return fail(KURBM_ERR_ARG, "null argument");
    const bool v_binary = (v_pieces == (1 | KURBM_V_BINARY));
    if (v_binary) v_pieces = 1;
    if (rows <= 0 || n_vis <= 0 || n_hid <= 0 || (v_pieces != 1 && v_pieces != 3)) return fail(KURBM_ERR_ARG, "bad shape / v_pieces");
    const WorkspaceB w = carve_bf16(ctx, const_cast<void*>(workspace), rows, n_vis, n_hid, 3, v_pieces);
    if (w.bytes > workspace_bytes) return fail(KURBM_ERR_WORKSPACE, "workspace too small");
    const bool gauss = (mode == KURBM_MODE_VISIBLE_GAUSSIAN), byt = ctx->knob[KN_X3_BYTES] != 0;
    const bool f8pos = v_binary && ctx->knob[KN_X3_F8POS] != 0;
    DumpArgs a;
    memset(&a, 0, sizeof a);
    a.out = out; a.rows = rows; a.ld_out = ld_out; a.pieces = 1; a.sign = 1.f;
    switch (which) {
        case KURBM_PLANE_H_POS:   a.src = w.hb;   a.units = n_hid; a.ld = w.Lh; a.fmt = byt ? 1 : 0; break;
        case KURBM_PLANE_H_POS_T: a.src = w.hbT;  a.units = n_hid; a.ld = w.Lb; a.fmt = f8pos ? 2 : 0; a.transposed = 1;
                                  if (split_stats_on(ctx, 3, v_pieces)) { a.fmt = 1; a.ld = 2 * w.Lb; }   // (bytes at the bf16 plane's row stride)
                                  break;
        case KURBM_PLANE_V_NEG:   a.src = w.v2b;  a.units = n_vis; a.ld = w.Lv; a.fmt = (byt && !gauss) ? 1 : 0;
                                  a.pieces = gauss ? 3 : 1; a.plane = w.planeV; break;
        case KURBM_PLANE_V_NEG_T: a.src = w.v2bT; a.units = n_vis; a.ld = w.Lb; a.transposed = 1;
                                  a.pieces = gauss ? 3 : 1; a.plane = w.planeVT;
                                  if (atr_on(ctx, 3, v_pieces, gauss)) {   // (no transposed copy exists: the statistics read the row-major pieces)
                                      a.src = w.v2b; a.ld = w.Lv; a.transposed = 0; a.plane = w.planeV;
                                  }
                                  if ((f8pos && byt && !gauss && ctx->knob[KN_X3_STATS_BYTES] != 0 && ctx->knob[KN_X3_STATS_TALL] != 0 && n_vis > 128) ||
                                      (split_stats_on(ctx, 3, v_pieces) && !gauss)) {
                                      a.fmt = 1; a.ld = 2 * w.Lb;   // (bytes at the bf16 plane's row stride: cd_step_any, tbytes)
                                  }
                                  break;
        case KURBM_PLANE_H_NEG_T: a.src = w.hnT;  a.units = n_hid; a.ld = w.Lb; a.transposed = 1; a.pieces = 3; a.plane = w.planeHT;
                                  a.sign = -1.f; break;
        default: return fail(KURBM_ERR_ARG, "unknown plane %d", which);
    }
    if (ld_out < a.units) return fail(KURBM_ERR_ARG, "ld_out < units");
    HIP_TRY(launch_dump_plane(a, static_cast<hipStream_t>(stream)));
    return KURBM_OK;
}

int kurbm_cd_epoch_x3(kurbm_ctx* ctx, const kurbm_params* p, void* mirror, size_t mirror_bytes, const float* V,
                      int v_pieces, int n_rows, int ldv, int batch_size, const kurbm_cd_opts* opts, void* workspace,
                      size_t workspace_bytes, kurbm_stream_t stream) {
    if (!ctx || !opts) return fail(KURBM_ERR_ARG, "null argument");
    if (n_rows < 0 || batch_size <= 0) return fail(KURBM_ERR_ARG, "bad row count / batch size");
    if (opts->delta_out || !opts->apply) return fail(KURBM_ERR_ARG, "kurbm_cd_epoch_x3 applies in place: apply = 1, delta_out = null");
    if (opts->v_planes && opts->v_planes_stride < carve_vplanes(ctx, nullptr, batch_size < n_rows ? batch_size : (n_rows > 0 ? n_rows : 1),
                                                                 p ? p->n_vis : 1, v_pieces == 3 ? 3 : 1).bytes)
        return fail(KURBM_ERR_ARG, "v_planes_stride is smaller than the planes of one batch");
    kurbm_cd_opts o = *opts;
    int steps = 0;
    for (int lo = 0; lo < n_rows; lo += batch_size, ++steps) {
        const int rows = (n_rows - lo < batch_size) ? n_rows - lo : batch_size;
        if (opts->v_planes) o.v_planes = static_cast<const char*>(opts->v_planes) + (size_t)steps * opts->v_planes_stride;
        if (int e = cd_step_any(ctx, 3, v_pieces, p, mirror, mirror_bytes, V + (size_t)lo * ldv, rows, ldv, &o, 7, workspace,
                                workspace_bytes, stream))
            return e;
        ++o.step;
    }
    return steps;
}

int kurbm_cd_chain_x3(kurbm_ctx* ctx, const kurbm_params* p, void* mirror, size_t mirror_bytes, const float* v_batch,
                      int v_pieces, int rows, int ldv, const kurbm_cd_opts* o, void* workspace, size_t workspace_bytes,
                      kurbm_stream_t stream) {
    return cd_step_any(ctx, 3, v_pieces, p, mirror, mirror_bytes, v_batch, rows, ldv, o, 7, workspace, workspace_bytes, stream, 8);
}

int kurbm_x3_stats_rows(kurbm_ctx* ctx, const kurbm_params* p, void* mirror, size_t mirror_bytes, const float* v_batch,
                        int v_pieces, int rows, int ldv, const kurbm_cd_opts* o, int m_lo, int m_hi, void* workspace,
                        size_t workspace_bytes, kurbm_stream_t stream) {
    if (!p || !o) return fail(KURBM_ERR_ARG, "null argument");
    if (o->apply || !o->delta_out) return fail(KURBM_ERR_ARG, "kurbm_x3_stats_rows emits sums: apply = 0, delta_out set");
    if (m_lo < 0 || m_hi > p->n_vis || m_lo >= m_hi || (m_lo & 127))
        return fail(KURBM_ERR_ARG, "row range: 0 <= m_lo < m_hi <= n_vis, m_lo a multiple of 128");
    return cd_step_any(ctx, 3, v_pieces, p, mirror, mirror_bytes, v_batch, rows, ldv, o, 7, workspace, workspace_bytes, stream, 7,
                       m_lo, m_hi);
}

// W (rows [m_lo, m_hi)), and with `bias` b_h and b_v, += lr * (packed delta); the bf16 pieces of the new weights rewritten
// in both orientations -- one launch: the packed delta is one "slab" and one row of bias partials each.
static int apply_delta_rows(kurbm_ctx* ctx, const kurbm_params* p, void* mirror, size_t mirror_bytes, int pieces, const float* delta,
                            float lr, int which, int m_lo, int m_hi, bool bias, hipStream_t st) {
    const Mirror m = carve_mirror(ctx, mirror, p->n_vis, p->n_hid, pieces);
    if (m.bytes > mirror_bytes) return fail(KURBM_ERR_WORKSPACE, "mirror too small: need %zu bytes, got %zu", m.bytes, mirror_bytes);
    ReduceArgs a;
    memset(&a, 0, sizeof a);
    const size_t nw = (size_t)p->n_vis * p->n_hid;
    a.slab = delta + (size_t)m_lo * p->n_hid; a.slab_stride = 0; a.nslab = 1; a.ld_slab = p->n_hid;
    a.n_vis = m_hi - m_lo; a.n_vis_bias = p->n_vis; a.n_hid = p->n_hid; a.ldw = p->ldw; a.lr = lr;
    a.W = p->W + (size_t)m_lo * p->ldw;
    if (bias) {
        a.part_h = delta + nw; a.nrow_tiles_h = 1; a.ld_part_h = p->n_hid; a.b_h = (which & 2) ? p->b_h : nullptr;
        a.part_v = delta + nw + p->n_hid; a.nrow_tiles_v = 1; a.ld_part_v = p->n_vis; a.b_v = (which & 4) ? p->b_v : nullptr;
    }
    a.Wb = m.Wb + (size_t)m_lo * m.ldW; a.ldWb = m.ldW; a.planeWb = m.planeW;
    a.Wtb = m.Wtb + m_lo; a.ldWtb = m.ldWt; a.planeWtb = m.planeWt; a.pieces = pieces;
    // the transposed mirror's k columns of these rows; the last range also rewrites the zero k padding behind n_vis
    a.wtb_k_ext = (m_hi == p->n_vis) ? m.ldWt - m_lo : m_hi - m_lo;
    a.tile_rows = ctx->knob[KN_REDUCE_TR];
    HIP_TRY(launch_reduce_apply_split(a, st));
    return KURBM_OK;
}

// Row ranges of dW in the data-parallel step.  The AUTOMATIC choice (n_chunks <= 0, knob KURBM_DP_CHUNKS unset) is ONE range,
// all-reduced on the caller's stream, whatever the size of the exchange: on MI355X / ROCm 7 a hand-off between two HIP streams
// costs 12-16 us per event wait and a statistics GEMM cut in two ~13 us (tools/dp_times.py) -- ~49 us before a second range's
// all-reduce can start, more than the half of a 3.2 MB exchange it could hide -- and the several-range schedule (range i
// all-reduced and applied on the comm stream under the statistics GEMM of range i + 1) has never run at a world size above 1
// (tests/test_two_gpus.py is skipped on every box this build has seen).  It stays in the tree as an OPT-IN: the caller's
// n_chunks > 1, or KURBM_DP_CHUNKS > 1 (every rank must pass the same value), e.g. 4 ranges of ~16 MiB for the 67 MB of a
// 4096 x 4096 RBM once tests/test_two_gpus.py has passed on a real node.
static int auto_chunks(size_t /*bytes*/) { return 1; }

static int cd_step_dp_any(kurbm_ctx* ctx, kurbm_comm* comm, int pieces, const kurbm_params* p, void* mirror, size_t mirror_bytes,
                          const float* v_batch, int v_pieces, int rows, int ldv, const kurbm_cd_opts* opts, int n_chunks,
                          void* workspace, size_t workspace_bytes, kurbm_stream_t stream) {
    if (!ctx || !comm || !opts) return fail(KURBM_ERR_ARG, "null argument");
    if (int e = check_params(p)) return e;
    if (!opts->delta_out || !aligned16(opts->delta_out)) return fail(KURBM_ERR_ARG, "delta_out (the packed sums) is required, 16-byte aligned");
    if (comm->device != ctx->device) return fail(KURBM_ERR_ARG, "communicator is on device %d, context on %d", comm->device, ctx->device);
    if (rows < 0) return fail(KURBM_ERR_ARG, "rows must be >= 0");
    // everything cd_step_any would refuse is refused HERE, before the first collective is enqueued: a rank that returned
    // an error from inside the sequence would leave the other ranks waiting in ncclAllReduce
    if (int e = check_cd_args(ctx, pieces, v_pieces, p, mirror, mirror_bytes, v_batch, rows, ldv, opts, workspace, workspace_bytes, true)) return e;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t nw = (size_t)p->n_vis * p->n_hid, ntot = nw + p->n_hid + p->n_vis;
    // row ranges of dW: boundaries on multiples of 128 (the statistics tile), the same on every rank
    if (n_chunks <= 0) n_chunks = ctx->knob[KN_DP_CHUNKS] > 0 ? ctx->knob[KN_DP_CHUNKS] : auto_chunks(ntot * sizeof(float));
    if (n_chunks > kurbm_comm::MAX_CHUNKS) n_chunks = kurbm_comm::MAX_CHUNKS;
    int bound[kurbm_comm::MAX_CHUNKS + 1];
    int nc = 0;
    bound[0] = 0;
    for (int c = 1; c < n_chunks; ++c) {
        const int b = (int)((long long)p->n_vis * c / n_chunks) / 128 * 128;
        if (b > bound[nc] && b < p->n_vis) bound[++nc] = b;
    }
    bound[++nc] = p->n_vis;
    kurbm_cd_opts o = *opts;
    o.apply = 0;
    // several ranges: each one is applied (W rows, the mirror's pieces of those rows) on the comm stream as soon as its
    // all-reduce has landed, under the statistics GEMM of the next; that needs 16-byte-aligned rows in the packed buffer
    const bool apply_ranges = opts->apply && nc > 1 && (p->n_hid & 3) == 0;
    if (rows == 0) HIP_TRY(hipMemsetAsync(o.delta_out, 0, ntot * sizeof(float), st));
    else if (nc > 1)
        if (int e = cd_step_any(ctx, pieces, v_pieces, p, mirror, mirror_bytes, v_batch, rows, ldv, &o, 7, workspace, workspace_bytes, stream, 8))
            return e;
    // Once the first collective of a several-range step is enqueued, this rank must finish the sequence as far as the other
    // ranks can see it: a failure inside the loop (a HIP error of a launch or an event call) still records ev_done on the comm
    // stream and makes the caller's stream wait for it before the error is returned -- the comm stream never stays ahead of
    // the caller's, and the collectives already enqueued complete against the peers' matching ones.  (Argument errors cannot
    // occur here: check_cd_args ran before anything was enqueued.)
    int err = KURBM_OK;
    bool on_comm = false;
    auto range = [&](int c) -> int {
        if (rows > 0) {
            const int e = (nc == 1)
                ? cd_step_any(ctx, pieces, v_pieces, p, mirror, mirror_bytes, v_batch, rows, ldv, &o, 7, workspace, workspace_bytes, stream)
                : cd_step_any(ctx, pieces, v_pieces, p, mirror, mirror_bytes, v_batch, rows, ldv, &o, 7, workspace, workspace_bytes, stream, 7,
                              bound[c], bound[c + 1]);
            if (e) return e;
        }
        const size_t lo = (size_t)bound[c] * p->n_hid, hi = (c + 1 == nc) ? ntot : (size_t)bound[c + 1] * p->n_hid;
        if (nc == 1)    // nothing to overlap with: the all-reduce stays on the caller's stream, no hand-off
            return comm_allreduce_sum(comm, o.delta_out + lo, hi - lo, st);
        // this range's sums are complete on `stream`: hand them to the comm stream
        HIP_TRY(hipEventRecord(comm->ev_ready[c], st));
        HIP_TRY(hipStreamWaitEvent(comm->stream, comm->ev_ready[c], 0));
        on_comm = true;
        if (int e = comm_allreduce_sum(comm, o.delta_out + lo, hi - lo, comm->stream)) return e;
        if (apply_ranges)
            return apply_delta_rows(ctx, p, mirror, mirror_bytes, pieces, o.delta_out, o.lr, 7, bound[c], bound[c + 1], c + 1 == nc, comm->stream);
        return KURBM_OK;
    };
    for (int c = 0; c < nc && !err; ++c) err = range(c);
    if (on_comm) {   // (also on the error path: see above)
        const std::string keep = g_err;
        const hipError_t e1 = hipEventRecord(comm->ev_done, comm->stream);
        const hipError_t e2 = (e1 == hipSuccess) ? hipStreamWaitEvent(st, comm->ev_done, 0) : e1;
        if (err) g_err = keep;
        else if (e2 != hipSuccess) err = fail(KURBM_ERR_HIP, "joining the comm stream: %s", hipGetErrorString(e2));
    }
    if (err) return err;
    if (opts->apply && !apply_ranges) {
        if (pieces == 3) return kurbm_x3_apply_delta(ctx, p, mirror, mirror_bytes, o.delta_out, o.lr, 7, stream);
        if ((p->n_hid & 3) == 0) return apply_delta_rows(ctx, p, mirror, mirror_bytes, pieces, o.delta_out, o.lr, 7, 0, p->n_vis, true, st);
        if (int e = kurbm_apply_delta(ctx, p, o.delta_out, o.lr, 7, stream)) return e;
        return mirror_refresh_any(ctx, pieces, p, mirror, mirror_bytes, stream);
    }
    return KURBM_OK;
}

// The data-parallel x3 step on the PEER exchange (kurbm_peer.hip): chain -> statistics -> slab reduce into this rank's exported
// `delta` region -> shot 1 (every rank sums its band of all ranks' deltas, in rank order) -> ONE launch that reads every band
// from its owner's `sum` region and applies it (W, biases, both weight-piece mirrors).  Two launches beyond the local step's own.
int kurbm_cd_step_x3_peer(kurbm_ctx* ctx, kurbm_peer* peer, const kurbm_params* p, void* mirror, size_t mirror_bytes,
                          const float* v_batch, int v_pieces, int rows, int ldv, const kurbm_cd_opts* opts, void* workspace,
                          size_t workspace_bytes, kurbm_stream_t stream) {
    if (!ctx || !peer || !opts) return fail(KURBM_ERR_ARG, "null argument");
    if (int e = check_params(p)) return e;
    if (int e = peer_geometry_ok(peer, ctx->device, p->n_vis, p->n_hid)) return e;
    if (!opts->apply) return fail(KURBM_ERR_ARG, "kurbm_cd_step_x3_peer applies the summed update in place: apply = 1");
    if (rows < 0) return fail(KURBM_ERR_ARG, "rows must be >= 0");
    // everything the step would refuse is refused HERE, before this rank publishes anything a peer waits for
    if (int e = check_cd_args(ctx, 3, v_pieces, p, mirror, mirror_bytes, v_batch, rows, ldv, opts, workspace, workspace_bytes, true)) return e;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t nw = (size_t)p->n_vis * p->n_hid, ntot = nw + p->n_hid + p->n_vis;
    kurbm_cd_opts o = *opts;
    o.apply = 0;
    o.delta_out = peer_delta(peer);
    if (rows == 0) HIP_TRY(hipMemsetAsync(o.delta_out, 0, ntot * sizeof(float), st));
    else if (int e = cd_step_any(ctx, 3, v_pieces, p, mirror, mirror_bytes, v_batch, rows, ldv, &o, 7, workspace, workspace_bytes, stream)) return e;
    PeerSrc src;
    if (int e = peer_exchange_shot1(peer, ctx->status, st, &src)) return e;
    // shot 2 = the apply: the packed total as ONE "slab" whose rows come from the band owners, one row of bias sums each
    const Mirror m = carve_mirror(ctx, mirror, p->n_vis, p->n_hid, 3);
    ReduceArgs a;
    memset(&a, 0, sizeof a);
    a.slab = src.own_sum; a.slab_stride = 0; a.nslab = 1; a.ld_slab = p->n_hid;
    a.n_vis = p->n_vis; a.n_hid = p->n_hid; a.ldw = p->ldw; a.lr = o.lr;
    a.W = p->W;
    a.part_h = src.own_sum + nw; a.nrow_tiles_h = 1; a.ld_part_h = p->n_hid; a.b_h = p->b_h;
    a.part_v = src.own_sum + nw + p->n_hid; a.nrow_tiles_v = 1; a.ld_part_v = p->n_vis; a.b_v = p->b_v;
    a.Wb = m.Wb; a.ldWb = m.ldW; a.planeWb = m.planeW;
    a.Wtb = m.Wtb; a.ldWtb = m.ldWt; a.planeWtb = m.planeWt; a.pieces = 3;
    a.tile_rows = ctx->knob[KN_REDUCE_TR];
    HIP_TRY(launch_reduce_apply_split(a, st, &src));
    return KURBM_OK;
}

int kurbm_cd_step_x3_dp(kurbm_ctx* ctx, kurbm_comm* comm, const kurbm_params* p, void* mirror, size_t mirror_bytes,
                        const float* v_batch, int v_pieces, int rows, int ldv, const kurbm_cd_opts* opts, int n_chunks,
                        void* workspace, size_t workspace_bytes, kurbm_stream_t stream) {
    return cd_step_dp_any(ctx, comm, 3, p, mirror, mirror_bytes, v_batch, v_pieces, rows, ldv, opts, n_chunks, workspace, workspace_bytes, stream);
}

int kurbm_cd_step_bf16_dp(kurbm_ctx* ctx, kurbm_comm* comm, const kurbm_params* p, void* mirror, size_t mirror_bytes,
                          const float* v_batch, int rows, int ldv, const kurbm_cd_opts* opts, int n_chunks, void* workspace,
                          size_t workspace_bytes, kurbm_stream_t stream) {
    return cd_step_dp_any(ctx, comm, 1, p, mirror, mirror_bytes, v_batch, 1, rows, ldv, opts, n_chunks, workspace, workspace_bytes, stream);
}

int kurbm_x3_apply_delta(kurbm_ctx* ctx, const kurbm_params* p, void* mirror, size_t mirror_bytes, const float* delta,
                         float lr, int which, kurbm_stream_t stream) {
    if (!ctx || !delta) return fail(KURBM_ERR_ARG, "null argument");
    if (int e = check_params(p)) return e;
    if (!mirror || !aligned16(mirror) || !aligned16(delta)) return fail(KURBM_ERR_ARG, "mirror / delta null or misaligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if ((p->n_hid & 3) || !(which & 1)) {   // packed rows not 16-byte aligned, or W untouched: two launches
        const Mirror m = carve_mirror(ctx, mirror, p->n_vis, p->n_hid, 3);
        if (m.bytes > mirror_bytes) return fail(KURBM_ERR_WORKSPACE, "mirror too small: need %zu bytes, got %zu", m.bytes, mirror_bytes);
        if (int e = kurbm_apply_delta(ctx, p, delta, lr, which, stream)) return e;
        if (which & 1)
            HIP_TRY(launch_f32_to_bf16(p->W, p->n_vis, p->n_hid, p->ldw, m.Wb, m.ldW, p->n_vis, m.Wtb, m.ldWt, p->n_hid, 3,
                                       m.planeW, m.planeWt, nullptr, 0, st));
        return KURBM_OK;
    }
    return apply_delta_rows(ctx, p, mirror, mirror_bytes, 3, delta, lr, which, 0, p->n_vis, true, st);
}

size_t kurbm_x3_planes_bytes(kurbm_ctx* ctx, int rows, int n_vis, int v_pieces) {
    if (v_pieces == (1 | KURBM_V_BINARY)) v_pieces = 1;
    if (!ctx || rows <= 0 || n_vis <= 0 || (v_pieces != 1 && v_pieces != 3)) return 0;
    return carve_vplanes(ctx, nullptr, rows, n_vis, v_pieces).bytes;
}

int kurbm_x3_convert_rows(kurbm_ctx* ctx, const float* v, int rows, int ldv, int n_vis, int v_pieces, void* planes,
                          size_t planes_bytes, kurbm_stream_t stream) {
    if (!ctx) return fail(KURBM_ERR_ARG, "ctx is null");
    // 0/1 data: the transposed plane (operand of the positive statistics only) as fp8, as the steps will read it
    const bool f8 = (v_pieces == (1 | KURBM_V_BINARY)) && ctx->knob[KN_X3_F8POS] != 0;
    const bool vbytes = (v_pieces == (1 | KURBM_V_BINARY)) && ctx->knob[KN_X3_BYTES] != 0;   // ... and the row-major one as bytes
    if (v_pieces == (1 | KURBM_V_BINARY)) v_pieces = 1;
    if (rows <= 0 || n_vis <= 0 || (v_pieces != 1 && v_pieces != 3)) return fail(KURBM_ERR_ARG, "bad shape / v_pieces");
    if (bad_matrix(v, ldv, n_vis)) return fail(KURBM_ERR_ARG, "v: null, misaligned, ld %% 4 != 0 or ld < n_vis");
    if (!planes || !aligned16(planes)) return fail(KURBM_ERR_ARG, "planes is null or misaligned");
    const VPlanes vp = carve_vplanes(ctx, planes, rows, n_vis, v_pieces);
    if (vp.bytes > planes_bytes) return fail(KURBM_ERR_WORKSPACE, "planes too small: need %zu bytes, got %zu", vp.bytes, planes_bytes);
    const int Kb = round_up(rows, 128), Lv = ld_pad(ctx, round_up(n_vis, 128)), Lb = ld_pad(ctx, Kb);
    HIP_TRY(launch_f32_to_bf16(v, rows, n_vis, ldv, vp.vb, Lv, Kb, vp.vbT, Lb, n_vis, v_pieces, (size_t)Kb * Lv, (size_t)n_vis * Lb,
                               vp.part_v, round_up(n_vis, 4), static_cast<hipStream_t>(stream), f8 ? 1 : 0, vbytes ? 1 : 0));
    return KURBM_OK;
}

int kurbm_bf16_exact(kurbm_ctx* ctx, const float* x, int rows, int cols, int ld, int* flag, kurbm_stream_t stream) {
    if (!ctx || !flag) return fail(KURBM_ERR_ARG, "null argument");
    if (rows <= 0 || cols <= 0 || bad_matrix(x, ld, cols)) return fail(KURBM_ERR_ARG, "x: null, misaligned, ld %% 4 != 0 or ld < cols");
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIP_TRY(hipMemsetAsync(flag, 0, sizeof(int), st));
    HIP_TRY(launch_bf16_exact_check(x, rows, cols, ld, flag, st));
    return KURBM_OK;
}

}  // extern "C"
