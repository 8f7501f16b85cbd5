/* A caller with nothing but a C compiler: drives liblyssa_hip.so through the library-owned context of
 * include/lyssa_hip.h (no PyTorch, no HIP calls of its own).  Built and run by tests/test_gpu_parity.py::test_c_abi_context.
 * Every signal is an exact two-atom combination x = 2 d_a - 1.5 d_b, so Batch-OMP with k = 2 must return {a, b} with
 * those coefficients. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "lyssa_hip.h"

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
    lys_ctx_destroy(ctx);
    if (bad == 0 && st[1] == 5.0 && rc < 0 && ms[1] > 0.0) {
        printf("OK\n");
        return 0;
    }
    printf("FAIL\n");
    return 1;
}
