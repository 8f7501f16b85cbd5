/*
 * bomp_oracle.c -- plain-C float64 restatement of the reference's Batch-OMP (TEST INFRASTRUCTURE ONLY).
 *
 * Same arithmetic structure as lyssa/sparse_coding.py:302-367 (`batch_omp`) after the precompute of :629-631
 * (Gram = D'D, Alpha = D'X): per signal, greedy argmax|a| (first maximum wins, :322), stop on re-selection
 * (:323-325), incremental Cholesky of G[Dx,Dx] with a HARD-CODED unit diagonal (w = L^-1 g, vs = 1 - w'w, stop if
 * vs < eps, new row [w, sqrt(vs)], :327-349), coefficients by forward + backward substitution (:353-354),
 * a = a0 - G[:,Dx] z (:359).  Also records the minimum relative top-1/top-2 gap along the greedy path (tie
 * classifier of the parity tests).  It exists because the numpy oracle does ~2 k signals/s: this one lets the GPU
 * parity tests compare 10^5..10^6 signals directly.  Pinned against the numpy oracle and the reference-generated
 * golden vectors in tests/test_oracle_golden.py.  Nothing under lyssandra_amd/ links or loads it.
 *
 * Layouts: D column-major-by-atom [K][n] (atom contiguous), X signal-major [N][n], outputs idx [N][k] (-1 padded,
 * selection order), coef [N][k], nnz [N], gap [N].
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define EPS64 2.220446049250313e-16

/* G[K][K] = D D' (atoms are rows of D here) */
void lyso_gram(const double* D, int n, int K, double* G) {
#pragma omp parallel for schedule(static)
    for (int a = 0; a < K; ++a)
        for (int b = 0; b < K; ++b) {
            double s = 0.0;
            for (int f = 0; f < n; ++f) s += D[(size_t)a * n + f] * D[(size_t)b * n + f];
            G[(size_t)a * K + b] = s;
        }
}

static void one_signal(const double* a0, const double* G, int K, int k, int32_t* idx, double* coef, int32_t* nnz,
                       double* gap, double* a, double* L, double* w, double* z, double* y) {
    int m = 0; /* len(Dx) */
    double min_gap = INFINITY;
    memcpy(a, a0, (size_t)K * sizeof(double));
    for (int j = 0; j < k; ++j) {
        /* argmax |a|, first maximum wins; runner-up for the gap */
        int kk = 0;
        double top = fabs(a[0]), second = -1.0;
        for (int c = 1; c < K; ++c) {
            const double v = fabs(a[c]);
            if (v > top) { second = top; top = v; kk = c; }
            else if (v > second) second = v;
        }
        int seen = 0;
        for (int i = 0; i < m; ++i) seen |= (idx[i] == kk);
        if (K > 1 && !seen) {
            const double g_ = (top > 0.0) ? (top - second) / top : 0.0;
            if (g_ < min_gap) min_gap = g_;
        }
        if (seen) break;
        if (j > 0) {
            /* w = L^-1 G[Dx, kk] */
            double ww = 0.0;
            for (int i = 0; i < m; ++i) {
                double s = G[(size_t)idx[i] * K + kk];
                for (int t = 0; t < i; ++t) s -= L[i * k + t] * w[t];
                w[i] = s / L[i * k + i];
                ww += w[i] * w[i];
            }
            const double vs = 1.0 - ww;
            if (vs < EPS64) break;
            for (int i = 0; i < m; ++i) L[m * k + i] = w[i];
            L[m * k + m] = sqrt(vs);
        } else {
            L[0] = 1.0;
        }
        idx[m] = kk;
        ++m;
        /* y = L^-1 a0[Dx];  z = L^-T y */
        for (int i = 0; i < m; ++i) {
            double s = a0[idx[i]];
            for (int t = 0; t < i; ++t) s -= L[i * k + t] * y[t];
            y[i] = s / L[i * k + i];
        }
        for (int i = m - 1; i >= 0; --i) {
            double s = y[i];
            for (int t = i + 1; t < m; ++t) s -= L[t * k + i] * z[t];
            z[i] = s / L[i * k + i];
        }
        /* a = a0 - G[:, Dx] z */
        memcpy(a, a0, (size_t)K * sizeof(double));
        for (int i = 0; i < m; ++i) {
            const double zi = z[i];
            const double* gr = G + (size_t)idx[i] * K; /* symmetric: row == column */
            for (int c = 0; c < K; ++c) a[c] -= gr[c] * zi;
        }
    }
    for (int i = 0; i < k; ++i) {
        coef[i] = (i < m) ? z[i] : 0.0;
        if (i >= m) idx[i] = -1;
    }
    *nnz = m;
    *gap = isfinite(min_gap) ? min_gap : 0.0;
}

/* X [N][n], D [K][n], G [K][K] (from lyso_gram) */
void lyso_bomp(const double* X, const double* D, const double* G, int n, int K, int k, int64_t N, int32_t* idx,
               double* coef, int32_t* nnz, double* gap) {
#pragma omp parallel
    {
        double* a0 = (double*)malloc((size_t)K * sizeof(double));
        double* a = (double*)malloc((size_t)K * sizeof(double));
        double* L = (double*)calloc((size_t)k * k, sizeof(double));
        double* w = (double*)malloc((size_t)k * sizeof(double));
        double* z = (double*)malloc((size_t)k * sizeof(double));
        double* y = (double*)malloc((size_t)k * sizeof(double));
#pragma omp for schedule(dynamic, 64)
        for (int64_t s = 0; s < N; ++s) {
            for (int c = 0; c < K; ++c) { /* Alpha[:, s] = D' x */
                double acc = 0.0;
                const double* d = D + (size_t)c * n;
                const double* x = X + (size_t)s * n;
                for (int f = 0; f < n; ++f) acc += d[f] * x[f];
                a0[c] = acc;
            }
            memset(L, 0, (size_t)k * k * sizeof(double));
            one_signal(a0, G, K, k, idx + s * k, coef + s * k, nnz + s, gap + s, a, L, w, z, y);
        }
        free(a0); free(a); free(L); free(w); free(z); free(y);
    }
}
