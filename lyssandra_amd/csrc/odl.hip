// Online dictionary learning (Mairal) sufficient statistics and block update for gfx950 --
// lyssa/dict_learning/online_dict_learn.py:84-98.
//
//   dA = Z Z'   and   dB = X Z'   are accumulated from the k-sparse codes through the atom-major index
//   (k^2 + k*n scalar updates per signal instead of the reference's dense 2K^2 + 2nK FLOP per signal);
//   one workgroup owns one atom: row `a` of dA lives in LDS, column `a` of dB in registers.
//   The dictionary update is one fp32 MFMA GEMM (DA = D A, computed ONCE per batch like the reference, :91)
//   plus a fused epilogue (:93-98).
#include "common.h"

namespace lys {

int gemm_nt(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int Nc,
            int Kin, hipStream_t stream, bool stream_c = false);
int transpose(const float* src, int rows, int cols, int ld_src, float* dst, int ld_dst, hipStream_t stream);

// One workgroup per atom, 64 teams of 16 lanes; a team takes every 64th signal of the atom (each signal is a chain of
// dependent loads: entry -> coefficient / support row / patch row), so a workgroup keeps 64 chains in flight instead of
// 4 (315 -> 273 us per 65 536-signal mini-batch at K = 1024: the rest is the latency of those chains).
constexpr int ODL_THREADS = 1024;
__global__ __launch_bounds__(ODL_THREADS) void odl_increment_kernel(const float* __restrict__ X, int64_t ldx, int n,
                                                                    int Kp, int ldd, int k,
                                                                    const int32_t* __restrict__ idx,
                                                                    const float* __restrict__ coef,
                                                                    const int32_t* __restrict__ nnz,
                                                                    const int32_t* __restrict__ row_ptr,
                                                                    const int32_t* __restrict__ entry,
                                                                    float* __restrict__ dA, float* __restrict__ dB) {
    extern __shared__ float s_mem[];
    float* s_row = s_mem;       // [Kp]   row `a` of Z Z'
    float* s_b = s_mem + Kp;    // [ldd]  column `a` of X Z'
    const int a = blockIdx.x;
    const int team = threadIdx.x >> 4, q = threadIdx.x & 15;
    for (int x = threadIdx.x; x < Kp + ldd; x += ODL_THREADS) s_mem[x] = 0.f;
    __syncthreads();
    const int beg = row_ptr[a], end = row_ptr[a + 1];
    for (int e = beg + team; e < end; e += ODL_THREADS / 16) {
        const int ss = entry[e];
        const int64_t sig = ss / k;
        const float xa = coef[ss];
        const int m = nnz[sig];
        for (int j = q; j < m; j += 16) atomicAdd(&s_row[idx[sig * k + j]], xa * coef[sig * k + j]);
        for (int f = q; f < n; f += 16) atomicAdd(&s_b[f], xa * X[sig * ldx + f]);
    }
    __syncthreads();
    for (int x = threadIdx.x; x < Kp; x += ODL_THREADS) dA[(int64_t)a * Kp + x] = s_row[x];
    for (int f = threadIdx.x; f < ldd; f += ODL_THREADS) dB[(int64_t)a * ldd + f] = s_b[f];
}

int odl_increments(const float* X, int64_t ldx, int n, int K, int k, const int32_t* idx, const float* coef,
                   const int32_t* nnz, const int32_t* row_ptr, const int32_t* entry, float* dA, float* dB,
                   hipStream_t stream) {
    const int Kp = padded_atoms(K), ldd = padded_features(n);
    const size_t lds = (size_t)(Kp + ldd) * sizeof(float);
    if (lds > 64 * 1024) {
        set_error("odl_increments: K = %d too large for one LDS row", K);
        return LYS_ENOSUP;
    }
    // rows >= K of dA/dB are written by nobody: the caller provides zeroed (or previously zero) buffers
    LYS_CHECK_HIP(hipMemsetAsync(dA, 0, (size_t)Kp * Kp * sizeof(float), stream));
    LYS_CHECK_HIP(hipMemsetAsync(dB, 0, (size_t)Kp * ldd * sizeof(float), stream));
    hipLaunchKernelGGL(odl_increment_kernel, dim3(K), dim3(ODL_THREADS), lds, stream, X, ldx, n, Kp, ldd, k, idx, coef, nnz,
                       row_ptr, entry, dA, dB);
    LYS_LAUNCH_CHECK();
    return LYS_OK;
}

__global__ void axpby_kernel(float* __restrict__ y, float beta, const float* __restrict__ x, int64_t count) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) y[i] = fmaf(beta, y[i], x[i]);
}

int axpby(float* y, float beta, const float* x, int64_t count, hipStream_t stream) {
    if (count <= 0) return LYS_OK;
    hipLaunchKernelGGL(axpby_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, y, beta, x, count);
    LYS_LAUNCH_CHECK();
    return LYS_OK;
}

// ---------------------------------------------------------------------------------------------
// Exchange format of the symmetric statistics matrix A = Z Z' (online_dict_learn.py:84): only the block-upper triangle
// travels.  Row block i = rows [i*blk, min((i+1)*blk, Kp)) keeps its columns [i*blk, Kp); the blocks follow each other in
// one flat buffer (K = 8192, blk = 1024: 144 MB instead of 256 MB).  `sym_pack` gathers, `sym_unpack` scatters the reduced
// buffer back and mirrors it below the block diagonal with an LDS-tiled transpose.
// ---------------------------------------------------------------------------------------------
__host__ __device__ inline int64_t sym_block_offset(int i, int Kp, int blk) {  // floats before row block i
    // sum_{j<i} blk * (Kp - j*blk)   (every block before the last is full)
    return (int64_t)i * blk * Kp - (int64_t)blk * blk * ((int64_t)i * (i - 1) / 2);
}

int64_t sym_packed_count(int Kp, int blk) {
    const int nbk = (Kp + blk - 1) / blk;
    const int last = nbk - 1;
    return sym_block_offset(last, Kp, blk) + (int64_t)(Kp - last * blk) * (Kp - last * blk);
}

template <bool PACK>
__global__ __launch_bounds__(256) void sym_rows_kernel(float* __restrict__ A, int Kp, int blk, float* __restrict__ flat) {
    const int r = blockIdx.x, i = r / blk, c0 = i * blk, w = Kp - c0;
    float* dst = flat + sym_block_offset(i, Kp, blk) + (int64_t)(r - c0) * w;
    float* row = A + (int64_t)r * Kp + c0;
    for (int c = threadIdx.x * 4; c < w; c += 256 * 4) {  // Kp and blk are multiples of 64: whole float4s
        if (PACK) *reinterpret_cast<float4*>(dst + c) = *reinterpret_cast<const float4*>(row + c);
        else *reinterpret_cast<float4*>(row + c) = *reinterpret_cast<const float4*>(dst + c);
    }
}

// A[c][r] = A[r][c] for the 32 x 32 tiles right of the block diagonal (tile row tr in block i, tile column >= (i+1)*blk/32)
__global__ __launch_bounds__(256) void sym_mirror_kernel(float* __restrict__ A, int Kp, int blk) {
    __shared__ float t[32][33];
    const int tr = blockIdx.y, tc = blockIdx.x;
    if (tc * 32 < (tr * 32 / blk + 1) * blk) return;  // inside or left of the diagonal block: already complete
    const int x = threadIdx.x & 31, y = threadIdx.x >> 5;
#pragma unroll
    for (int j = 0; j < 32; j += 8) t[y + j][x] = A[(int64_t)(tr * 32 + y + j) * Kp + tc * 32 + x];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 32; j += 8) A[(int64_t)(tc * 32 + y + j) * Kp + tr * 32 + x] = t[x][y + j];
}

int sym_pack(const float* A, int Kp, int blk, float* flat, hipStream_t stream) {
    if (Kp <= 0 || Kp % 64 || blk <= 0 || blk % 64) {
        set_error("sym_pack: Kp = %d and block = %d must be positive multiples of 64", Kp, blk);
        return LYS_EINVAL;
    }
    hipLaunchKernelGGL(sym_rows_kernel<true>, dim3(Kp), dim3(256), 0, stream, const_cast<float*>(A), Kp, blk, flat);
    LYS_LAUNCH_CHECK();
    return LYS_OK;
}

int sym_unpack(const float* flat, int Kp, int blk, float* A, hipStream_t stream) {
    if (Kp <= 0 || Kp % 64 || blk <= 0 || blk % 64) {
        set_error("sym_unpack: Kp = %d and block = %d must be positive multiples of 64", Kp, blk);
        return LYS_EINVAL;
    }
    hipLaunchKernelGGL(sym_rows_kernel<false>, dim3(Kp), dim3(256), 0, stream, A, Kp, blk, const_cast<float*>(flat));
    if (Kp > blk) hipLaunchKernelGGL(sym_mirror_kernel, dim3(Kp / 32, Kp / 32), dim3(256), 0, stream, A, Kp, blk);
    LYS_LAUNCH_CHECK();
    return LYS_OK;
}

// one wave per atom: d += (B - DA) / (A_aa + eps); clip; d /= (||d|| + eps)     (:93-98)
__global__ __launch_bounds__(64) void odl_update_kernel(float* __restrict__ D, int ldd, int n, int Kp,
                                                        const float* __restrict__ A, const float* __restrict__ B,
                                                        const float* __restrict__ DA, int non_neg) {
    const int a = blockIdx.x, lane = threadIdx.x;
    const float inv = 1.f / (A[(int64_t)a * Kp + a] + EPS64_F);
    float ss = 0.f;
    for (int f = lane; f < n; f += 64) {
        const int64_t o = (int64_t)a * ldd + f;
        float d = fmaf(inv, B[o] - DA[o], D[o]);
        if (non_neg && d < 0.f) d = 0.f;
        D[o] = d;
        ss = fmaf(d, d, ss);
    }
    const float nrm = sqrtf(wave_sum_f(ss)) + EPS64_F;
    for (int f = lane; f < n; f += 64) {
        const int64_t o = (int64_t)a * ldd + f;
        D[o] = D[o] / nrm;
    }
}

int odl_update(float* D, const float* A, const float* B, int n, int K, int non_neg, float* scratch,
               hipStream_t stream) {
    const int Kp = padded_atoms(K), ldd = padded_features(n);
    float* Dt = scratch;                      // [ldd][Kp] feature-major copy of D
    float* DA = scratch + (size_t)Kp * ldd;   // [Kp][ldd] (D A)' atom-major
    int rc = transpose(D, Kp, ldd, ldd, Dt, Kp, stream);
    if (rc) return rc;
    // DA_am[a][f] = sum_j A[a][j] * D_am[j][f] = sum_j A[a][j] * Dt[f][j]   (A symmetric)
    rc = gemm_nt(A, Kp, Dt, Kp, DA, ldd, Kp, ldd, Kp, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(odl_update_kernel, dim3(K), dim3(64), 0, stream, D, ldd, n, Kp, A, B, DA, non_neg);
    LYS_LAUNCH_CHECK();
    return LYS_OK;
}

// ---------------------------------------------------------------------------------------------
// Projected gradient step of lyssa/dict_learning/gradient_descent.py:84-98 (the learner the reference's own
// dictionary-learning test drives):  grad = (D Z - X) Z' = D (ZZ') - X Z';  D <- D - eta*grad + 2 mu D (D'D - I)
// (the incoherence term is ADDED in the reference, :92 -- reproduced);  clip;  norm_cols.
// With W = -eta*ZZ' + 2 mu (G - I) (symmetric) this is  D <- norm_cols(clip(D + D W + eta X Z')): one MFMA GEMM.
// ---------------------------------------------------------------------------------------------
__global__ void pgd_weight_kernel(const float* __restrict__ dA, const float* __restrict__ G, float eta, float mu,
                                  int Kp, float* __restrict__ W) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)Kp * Kp) return;
    float w = -eta * dA[t];
    if (G) w = fmaf(2.f * mu, G[t] - ((t / Kp) == (t % Kp) ? 1.f : 0.f), w);
    W[t] = w;
}

__global__ __launch_bounds__(64) void pgd_apply_kernel(float* __restrict__ D, int ldd, int n, const float* __restrict__ T,
                                                       const float* __restrict__ dB, float eta, int non_neg) {
    const int a = blockIdx.x, lane = threadIdx.x;
    float ss = 0.f;
    for (int f = lane; f < n; f += 64) {
        const int64_t o = (int64_t)a * ldd + f;
        float d = D[o] + T[o] + eta * dB[o];
        if (non_neg && d < 0.f) d = 0.f;
        D[o] = d;
        ss = fmaf(d, d, ss);
    }
    const float nrm = sqrtf(wave_sum_f(ss)) + EPS64_F;
    for (int f = lane; f < n; f += 64) {
        const int64_t o = (int64_t)a * ldd + f;
        D[o] = D[o] / nrm;
    }
}

int pgd_update(float* D, const float* dA, const float* dB, const float* G, int n, int K, float eta, float mu, int non_neg,
               float* scratch, hipStream_t stream) {
    const int Kp = padded_atoms(K), ldd = padded_features(n);
    float* W = scratch;                                   // [Kp][Kp]
    float* Dt = W + (size_t)Kp * Kp;                      // [ldd][Kp]
    float* T = Dt + (size_t)Kp * ldd;                     // [Kp][ldd]
    const int64_t tot = (int64_t)Kp * Kp;
    hipLaunchKernelGGL(pgd_weight_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream, dA,
                       (mu > 0.f) ? G : nullptr, eta, mu, Kp, W);
    LYS_LAUNCH_CHECK();
    int rc = transpose(D, Kp, ldd, ldd, Dt, Kp, stream);
    if (rc) return rc;
    rc = gemm_nt(W, Kp, Dt, Kp, T, ldd, Kp, ldd, Kp, stream);   // T_am[a][f] = sum_j W[a][j] D_am[j][f] = (D W)_am
    if (rc) return rc;
    hipLaunchKernelGGL(pgd_apply_kernel, dim3(K), dim3(64), 0, stream, D, ldd, n, T, dB, eta, non_neg);
    LYS_LAUNCH_CHECK();
    return LYS_OK;
}

// column normalisation of the packed dictionary (utils/math.py:65-71)
__global__ __launch_bounds__(64) void norm_atoms_kernel(float* __restrict__ D, int ldd, int n) {
    const int a = blockIdx.x, lane = threadIdx.x;
    float ss = 0.f;
    for (int f = lane; f < n; f += 64) {
        const float d = D[(int64_t)a * ldd + f];
        ss = fmaf(d, d, ss);
    }
    const float nrm = sqrtf(wave_sum_f(ss)) + EPS64_F;
    for (int f = lane; f < n; f += 64) D[(int64_t)a * ldd + f] /= nrm;
}

int norm_atoms(float* D, int n, int K, hipStream_t stream) {
    hipLaunchKernelGGL(norm_atoms_kernel, dim3(K), dim3(64), 0, stream, D, padded_features(n), n);
    LYS_LAUNCH_CHECK();
    return LYS_OK;
}

// sum of |G[a][b]| over a != b, a,b < K  (average_mutual_coherence, lyssa/dict_learning/utils.py:7-11)
__global__ __launch_bounds__(256) void offdiag_abs_sum_kernel(const float* __restrict__ G, int K, int Kp,
                                                              double* __restrict__ out) {
    __shared__ double s_part[4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    double acc = 0.0;
    for (int a = blockIdx.x; a < K; a += gridDim.x) {
        float s = 0.f;
        for (int b = threadIdx.x; b < K; b += 256)
            if (b != a) s += fabsf(G[(int64_t)a * Kp + b]);
        acc += (double)s;
    }
    // block reduction: per-wave DPP sum of the float parts is not enough (acc is per thread): go through LDS
    float f = (float)acc;
    const float w = wave_sum_f(f);
    if (lane == 0) s_part[wid] = (double)w;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]));
}

int offdiag_abs_sum(const float* G, int K, double* out, hipStream_t stream) {
    const int Kp = padded_atoms(K);
    LYS_CHECK_HIP(hipMemsetAsync(out, 0, sizeof(double), stream));
    const int grid = K < 1024 ? K : 1024;
    hipLaunchKernelGGL(offdiag_abs_sum_kernel, dim3(grid), dim3(256), 0, stream, G, K, Kp, out);
    LYS_LAUNCH_CHECK();
    return LYS_OK;
}

__global__ void densify_kernel(const int32_t* __restrict__ idx, const float* __restrict__ coef,
                               const int32_t* __restrict__ nnz, int k, int64_t N, double* __restrict__ Z) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N * k) return;
    const int64_t s = t / k;
    const int j = (int)(t % k);
    if (j < nnz[s]) Z[(int64_t)idx[t] * N + s] = (double)coef[t];
}

int densify_f64(const int32_t* idx, const float* coef, const int32_t* nnz, int K, int k, int64_t N, double* Z,
                hipStream_t stream) {
    LYS_CHECK_HIP(hipMemsetAsync(Z, 0, (size_t)K * (size_t)N * sizeof(double), stream));
    const int64_t tot = N * k;
    if (tot <= 0) return LYS_OK;
    hipLaunchKernelGGL(densify_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream, idx, coef, nnz, k, N,
                       Z);
    LYS_LAUNCH_CHECK();
    return LYS_OK;
}

}  // namespace lys
