/*
 * cd_step_demo.c -- one CD-1 update through the C ABI of include/kurbm.h, with no Python and no PyTorch in the
 * process: plain C host code, device memory from the HIP runtime API, the library doing the rest.
 *
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include examples/c/cd_step_demo.c \
 *       -L keras_unsupervised_amd/csrc -lkurbm -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/keras_unsupervised_amd/csrc -o cd_step_demo
 *
 * Runs the same update (784 x 256, 64 rows: the shape of the reference's example, ku/ebm/rbm.py:117-134) on the fp32-MFMA
 * kernels (kurbm_cd_step), on the x3 kernels (kurbm_cd_step_x3, resident data planes) and as ONE launch (kurbm_cd_step_small)
 * from the same parameters and counters, prints how far the results are apart (fp32 summation order only), and reads the score of
 * fit(verbose = 1) (kurbm_score_small) by polling pinned host memory -- exit code 0 iff the differences are <= 1e-5.
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "kurbm.h"

#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define KU(x) do { int e_ = (x); if (e_ < 0) { fprintf(stderr, "%s: %s\n", #x, kurbm_last_error()); return 3; } } while (0)

static uint32_t lcg_state = 12345u;
static float lcg_unit(void) { lcg_state = lcg_state * 1664525u + 1013904223u; return (float)(lcg_state >> 8) / 16777216.0f; }

int main(void) {
    const int nv = 784, nh = 256, rows = 64, ld = 784, ldw = 256;
    const size_t nW = (size_t)nv * ldw;
    float *hW = malloc(nW * 4), *hv = malloc((size_t)rows * ld * 4), *r1 = malloc(nW * 4), *r2 = malloc(nW * 4);
    for (size_t i = 0; i < nW; ++i) hW[i] = (lcg_unit() - 0.5f) * 0.1f;                       /* U(-0.05, 0.05): rbm.py:30-33 */
    for (size_t i = 0; i < (size_t)rows * ld; ++i) hv[i] = lcg_unit() < 0.19f ? 1.0f : 0.0f;

    kurbm_ctx* ctx;
    KU(kurbm_ctx_create(0, &ctx));
    float *W, *bh, *bv, *v;
    HIP(hipMalloc((void**)&W, nW * 4)); HIP(hipMalloc((void**)&bh, nh * 4)); HIP(hipMalloc((void**)&bv, nv * 4));
    HIP(hipMalloc((void**)&v, (size_t)rows * ld * 4));
    HIP(hipMemcpy(v, hv, (size_t)rows * ld * 4, hipMemcpyHostToDevice));
    kurbm_params p = {nv, nh, ldw, 0, W, bh, bv};
    kurbm_cd_opts o = {0};
    o.k = 1; o.mode = KURBM_MODE_VISIBLE_BERNOULLI; o.lr = 1e-3f; o.apply = 1; o.seed = 42; o.step = 0;

    /* fp32 MFMA kernels */
    size_t ws_bytes = kurbm_workspace_bytes(ctx, rows, nv, nh, 1);
    void* ws;
    HIP(hipMalloc(&ws, ws_bytes));
    HIP(hipMemcpy(W, hW, nW * 4, hipMemcpyHostToDevice)); HIP(hipMemset(bh, 0, nh * 4)); HIP(hipMemset(bv, 0, nv * 4));
    KU(kurbm_cd_step(ctx, &p, v, rows, ld, &o, 7, ws, ws_bytes, NULL));
    HIP(hipDeviceSynchronize());
    HIP(hipMemcpy(r1, W, nW * 4, hipMemcpyDeviceToHost));

    /* x3 kernels: weight-piece mirror, workspace, resident data planes */
    size_t mir_bytes = kurbm_x3_mirror_bytes(ctx, nv, nh), ws3_bytes = kurbm_x3_workspace_bytes(ctx, rows, nv, nh, 1, 1 | KURBM_V_BINARY);
    size_t pl_bytes = kurbm_x3_planes_bytes(ctx, rows, nv, 1 | KURBM_V_BINARY);
    void *mir, *ws3, *planes;
    int* flag;
    HIP(hipMalloc(&mir, mir_bytes)); HIP(hipMalloc(&ws3, ws3_bytes)); HIP(hipMalloc(&planes, pl_bytes)); HIP(hipMalloc((void**)&flag, 4));
    HIP(hipMemcpy(W, hW, nW * 4, hipMemcpyHostToDevice)); HIP(hipMemset(bh, 0, nh * 4)); HIP(hipMemset(bv, 0, nv * 4));
    HIP(hipMemset(flag, 0, 4));
    KU(kurbm_bf16_exact(ctx, v, rows, nv, ld, flag, NULL));
    int bits = 3;   /* bit 0: some value is not a bf16 value; bit 1: some value is neither 0 nor 1 */
    HIP(hipMemcpy(&bits, flag, 4, hipMemcpyDeviceToHost));
    if (bits) { fprintf(stderr, "0/1 data reported as %d\n", bits); return 4; }
    const int vp = 1 | KURBM_V_BINARY;   /* 0/1 data: one bf16 piece, positive statistics on the fp8 matrix cores */
    KU(kurbm_x3_mirror_refresh(ctx, &p, mir, mir_bytes, NULL));
    KU(kurbm_x3_convert_rows(ctx, v, rows, ld, nv, vp, planes, pl_bytes, NULL));
    o.v_planes = planes;
    KU(kurbm_cd_step_x3(ctx, &p, mir, mir_bytes, v, vp, rows, ld, &o, 7, ws3, ws3_bytes, NULL));
    HIP(hipDeviceSynchronize());
    HIP(hipMemcpy(r2, W, nW * 4, hipMemcpyDeviceToHost));

    /* the same update in ONE launch (kurbm_cd_step_small), then the score of fit(verbose = 1) in one more (kurbm_score_small):
     * the score and a 1.0f behind it land in PINNED host memory, which this thread polls -- no copy, no event, no synchronise */
    float* r3 = malloc(nW * 4);
    volatile float* score;
    HIP(hipHostMalloc((void**)&score, 16, hipHostMallocDefault));
    score[0] = 0.0f; score[1] = 0.0f;
    HIP(hipMemcpy(W, hW, nW * 4, hipMemcpyHostToDevice)); HIP(hipMemset(bh, 0, nh * 4)); HIP(hipMemset(bv, 0, nv * 4));
    o.v_planes = NULL;
    KU(kurbm_cd_step_small(ctx, &p, v, rows, ld, &o, 7, ws, ws_bytes, NULL));
    kurbm_cd_opts so = o;
    so.chain = 3;   /* the score's own chain of draws */
    KU(kurbm_score_small(ctx, &p, v, rows, ld, &so, (float*)score, NULL, ws, ws_bytes, NULL));
    long spins = 0;
    while (score[1] != 1.0f && spins < 2000000000L) ++spins;
    if (score[1] != 1.0f) { fprintf(stderr, "the score never arrived\n"); return 5; }
    const float got_score = score[0];
    HIP(hipDeviceSynchronize());
    HIP(hipMemcpy(r3, W, nW * 4, hipMemcpyDeviceToHost));
    int status = -1;
    KU(kurbm_ctx_status(ctx, &status));

    double maxdiff = 0.0, moved = 0.0, maxdiff_small = 0.0;
    for (size_t i = 0; i < nW; ++i) {
        double d = fabs((double)r1[i] - (double)r2[i]), m = fabs((double)r1[i] - (double)hW[i]), d3 = fabs((double)r1[i] - (double)r3[i]);
        if (d > maxdiff) maxdiff = d;
        if (m > moved) moved = m;
        if (d3 > maxdiff_small) maxdiff_small = d3;
    }
    printf("abi %d  max |W_fp32mfma - W_x3| = %.3g  max |W_fp32mfma - W_one_launch| = %.3g  max |W_new - W_old| = %.3g  score %.4f (polled, %ld spins)  status %d\n",
           kurbm_abi_version(), maxdiff, maxdiff_small, moved, got_score, spins, status);
    kurbm_ctx_destroy(ctx);
    return (maxdiff <= 1e-5 && maxdiff_small <= 1e-5 && moved > 1e-4 && status == 0 && got_score > 0.0f && got_score < 1e4f) ? 0 : 1;
}
