/* A caller with nothing but a C compiler: drives liblyssa_hip.so through the library-owned context of
 * include/lyssa_hip.h (no PyTorch, no HIP calls of its own).  Built and run by tests/test_gpu_parity.py::test_c_abi_context.
 * Every signal is an exact two-atom combination x = 2 d_a - 1.5 d_b, so Batch-OMP with k = 2 must return {a, b} with
 * those coefficients. */
#define _POSIX_C_SOURCE 200112L /* setenv */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "lyssa_hip.h"

/* oracle/bomp_oracle.c (float64 restatement of approx_ksvd, lyssa/dict_learning/ksvd.py:98-126): the checker */
int lyso_approx_ksvd(const double* X, double* D, int n, int K, int k, int64_t N, const int32_t* idx, double* coef,
                     const int32_t* nnz, int n_cycles, int32_t* unused, double* err);

static uint32_t lcg(uint32_t* s) { return *s = *s * 1664525u + 1013904223u; }
static float unif(uint32_t* s) { return (float)(lcg(s) >> 8) / 16777216.0f - 0.5f; }

#define CHECK(call)                                                        \
    do {                                                                   \
        int rc_ = (call);                                                  \
        if (rc_ != 0) {                                                    \
            printf("FAIL %s -> %d: %s\n", #call, rc_, lys_last_error());   \
            return 1;                                                      \
        }                                                                  \
    } while (0)

int main(void) {
    const int n = 64, K = 256, k = 2;
    const int64_t N = 3000;
    uint32_t seed = 12345u;
    float* D = malloc(sizeof(float) * K * n);
    float* X = malloc(sizeof(float) * N * n);
    int32_t* idx = malloc(sizeof(int32_t) * N * k);
    float* coef = malloc(sizeof(float) * N * k);
    int32_t* nnz = malloc(sizeof(int32_t) * N);
    int* A = malloc(sizeof(int) * N);
    int* B = malloc(sizeof(int) * N);
    for (int a = 0; a < K; ++a) {
        double s = 0.0;
        for (int f = 0; f < n; ++f) {
            D[a * n + f] = unif(&seed);
            s += (double)D[a * n + f] * D[a * n + f];
        }
        for (int f = 0; f < n; ++f) D[a * n + f] = (float)(D[a * n + f] / sqrt(s));
    }
    for (int64_t i = 0; i < N; ++i) {
        A[i] = (int)(lcg(&seed) >> 8) % K;
        do B[i] = (int)(lcg(&seed) >> 8) % K; while (B[i] == A[i]);
        for (int f = 0; f < n; ++f) X[i * n + f] = 2.0f * D[A[i] * n + f] - 1.5f * D[B[i] * n + f];
    }
    lys_ctx* ctx = NULL;
    CHECK(lys_ctx_create(0, &ctx));
    CHECK(lys_ctx_set_dictionary(ctx, D, n, K));
    CHECK(lys_ctx_bomp_encode(ctx, X, N, k, idx, coef, nnz));
    int bad = 0;
    for (int64_t i = 0; i < N; ++i) {
        const int ok = nnz[i] == 2 && idx[i * k] == A[i] && idx[i * k + 1] == B[i] && fabsf(coef[i * k] - 2.0f) < 1e-4f &&
                       fabsf(coef[i * k + 1] + 1.5f) < 1e-4f;
        bad += !ok;
    }
    double ms[4], st[4];
    CHECK(lys_ctx_timings(ctx, ms));
    printf("encode: %d of %lld signals wrong; h2d %.3f ms, kernels %.3f ms, d2h %.3f ms\n", bad, (long long)N, ms[0], ms[1], ms[2]);
    CHECK(lys_ctx_bomp_encode_synthetic(ctx, 7u, 1000, 100000, 5, st));
    printf("synthetic: %.0f patches, mean nnz %.3f, %.3f ms, %.3g patches/s\n", st[0], st[1], st[2], st[3]);
    /* error path: k out of range must come back as a code + message, not a crash */
    const int rc = lys_ctx_bomp_encode(ctx, X, N, 65, idx, coef, nnz);
    printf("k=65 -> %d (%s)\n", rc, lys_last_error());
    /* ---- dictionary learning through the context: resident signals, one approx K-SVD cycle against the float64 C
     * restatement run from the SAME codes; then the same cycle through the RCCL path (communicator over this one device);
     * then one online-DL mini-batch. */
    const int k3 = 3;
    const int64_t M = 20000;
    float* Y = malloc(sizeof(float) * M * n);
    for (int64_t i = 0; i < M; ++i) {
        const int a = (int)(lcg(&seed) >> 8) % K, b = (int)(lcg(&seed) >> 8) % K;
        for (int f = 0; f < n; ++f) Y[i * n + f] = 2.0f * D[a * n + f] - 1.5f * D[b * n + f] + 0.3f * unif(&seed);
    }
    int32_t* ci = malloc(sizeof(int32_t) * M * k3);
    float* cc = malloc(sizeof(float) * M * k3);
    int32_t* cn = malloc(sizeof(int32_t) * M);
    float* Dg = malloc(sizeof(float) * K * n);
    float* Dg2 = malloc(sizeof(float) * K * n);
    double *Yd = malloc(sizeof(double) * M * n), *Dd = malloc(sizeof(double) * K * n), *cd = malloc(sizeof(double) * M * k3);
    int32_t* unused = malloc(sizeof(int32_t) * K);
    int n_unused = -1;
    double err0 = 0.0, err1 = 0.0, err_ref = 0.0;
    CHECK(lys_ctx_set_signals(ctx, Y, M));
    CHECK(lys_ctx_encode_resident(ctx, k3));
    CHECK(lys_ctx_get_codes(ctx, ci, cc, cn));
    CHECK(lys_ctx_error(ctx, &err0));
    CHECK(lys_ctx_ksvd_sweep(ctx, &n_unused));
    CHECK(lys_ctx_error(ctx, &err1));
    CHECK(lys_ctx_get_dictionary(ctx, Dg));
    for (int64_t i = 0; i < M * n; ++i) Yd[i] = Y[i];
    for (int i = 0; i < K * n; ++i) Dd[i] = D[i];
    for (int64_t i = 0; i < M * k3; ++i) cd[i] = cc[i];
    const int nu_ref = lyso_approx_ksvd(Yd, Dd, n, K, k3, M, ci, cd, cn, 1, unused, &err_ref);
    double worst = 0.0;
    for (int i = 0; i < K * n; ++i) worst = fmax(worst, fabs((double)Dg[i] - Dd[i]));
    printf("ksvd sweep: error %.6g -> %.6g (float64 restatement %.6g), unused %d (ref %d), worst atom entry diff %.3g\n", err0,
           err1, err_ref, n_unused, nu_ref, worst);
    const int sweep_ok = worst < 2e-5 && n_unused == nu_ref && fabs(err1 - err_ref) < 1e-5 * err_ref && err1 < err0;
    /* the same through RCCL: one communicator over this device */
    setenv("LYS_CTX_FORCE_RCCL", "1", 1);
    lys_ctx* mctx = NULL;
    const int dev0 = 0;
    int multi_ok = 0;
    if (lys_ctx_create_multi(1, &dev0, &mctx) == 0) {
        CHECK(lys_ctx_set_dictionary(mctx, D, n, K));
        CHECK(lys_ctx_set_signals(mctx, Y, M));
        CHECK(lys_ctx_encode_resident(mctx, k3));
        int nu2 = -1;
        CHECK(lys_ctx_ksvd_sweep(mctx, &nu2));
        CHECK(lys_ctx_get_dictionary(mctx, Dg2));
        double w2 = 0.0;
        for (int i = 0; i < K * n; ++i) w2 = fmax(w2, fabs((double)Dg2[i] - Dg[i]));
        printf("rccl path (1 device): devices %d, unused %d, worst diff to the plain context %.3g\n",
               lys_ctx_device_count(mctx), nu2, w2);
        multi_ok = (w2 < 1e-6 && nu2 == n_unused);   /* statistics are summed with atomics: reproducible to ~1e-7, not bitwise */
        /* online DL on the multi context: A = Z Z' must have trace = sum of squared coefficients */
        CHECK(lys_ctx_set_dictionary(mctx, D, n, K));
        CHECK(lys_ctx_odl_reset(mctx));
        CHECK(lys_ctx_odl_accumulate(mctx, Y, M, k3, 0.0f));
        float* Ah = malloc(sizeof(float) * K * K);
        float* Bh = malloc(sizeof(float) * K * n);
        CHECK(lys_ctx_get_ab(mctx, Ah, Bh));
        CHECK(lys_ctx_get_codes(mctx, ci, cc, cn));
        double tr = 0.0, ss = 0.0;
        for (int a = 0; a < K; ++a) tr += Ah[a * K + a];
        for (int64_t i = 0; i < M; ++i)
            for (int j = 0; j < cn[i]; ++j) ss += (double)cc[i * k3 + j] * cc[i * k3 + j];
        CHECK(lys_ctx_odl_update(mctx, 0));
        CHECK(lys_ctx_get_dictionary(mctx, Dg2));
        double nmin = 1e9, nmax = 0.0;
        for (int a = 0; a < K; ++a) {
            double s2 = 0.0;
            for (int f = 0; f < n; ++f) s2 += (double)Dg2[a * n + f] * Dg2[a * n + f];
            nmin = fmin(nmin, sqrt(s2));
            nmax = fmax(nmax, sqrt(s2));
        }
        CHECK(lys_ctx_set_ab(mctx, Ah, Bh));
        printf("online DL: trace(A) %.6g vs sum z^2 %.6g; atom norms after the update [%.6f, %.6f]\n", tr, ss, nmin, nmax);
        multi_ok = multi_ok && fabs(tr - ss) < 1e-4 * ss && nmin > 0.999 && nmax < 1.001;
        lys_ctx_destroy(mctx);
    } else {
        printf("lys_ctx_create_multi failed: %s\n", lys_last_error());
    }
    lys_ctx_destroy(ctx);
    if (bad == 0 && st[1] == 5.0 && rc < 0 && ms[1] > 0.0 && sweep_ok && multi_ok) {
        printf("OK\n");
        return 0;
    }
    printf("FAIL (encode %d, sweep %d, multi/odl %d)\n", bad == 0, sweep_ok, multi_ok);
    return 1;
}
