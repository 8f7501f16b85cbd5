// Producer of the signal matrix for the encode path: dense grid of overlapping patches straight
// into the engine's signal-major fp32 layout, with the per-patch preprocessing fused.
//   grid_patches        lyssa/utils/img.py:420-489  (patch (i,j) of the grid at (i*step, j*step); feature order
//                       = C-order flatten of (patch_size, patch_size, channels); patches in row-major grid order)
//   preproc             lyssa/feature_extract/preproc.py:46-80: 'scaling' (x/255), 'local_centering' (minus the
//                       patch mean), 'contrast_normalization' (centre, then x/(||x||+eps)), 'normalization'
// HBM-bound byte shuffling: every pixel is read patch_size^2/step^2 times (from L2), every patch written once.
#include "common.h"

namespace lys {

__device__ __forceinline__ float row16_sum_p(float x) {
    x += dpp_f<0xB1>(x);
    x += dpp_f<0x4E>(x);
    x += dpp_f<0x124>(x);
    x += dpp_f<0x128>(x);
    return x;
}

// one 16-lane team per patch; lane q owns features q, q+16, q+32, ... (dim <= 16*MAXF)
template <typename PIX, int MAXF>
__global__ __launch_bounds__(256) void grid_patches_kernel(const PIX* __restrict__ img, int H, int W, int C, int patch,
                                                           int step, int n_pw, int64_t n_patches, float scale,
                                                           int center, int normalize, float* __restrict__ X,
                                                           int64_t ldx) {
    const int team = threadIdx.x >> 4, q = threadIdx.x & 15;
    const int dim = patch * patch * C, rowlen = patch * C;
    for (int64_t p = (int64_t)blockIdx.x * 16 + team; p < n_patches; p += (int64_t)gridDim.x * 16) {
        const int y0 = (int)(p / n_pw) * step, x0 = (int)(p % n_pw) * step;
        float v[MAXF];
        float sum = 0.f;
#pragma unroll
        for (int m = 0; m < MAXF; ++m) {
            const int f = q + 16 * m;
            v[m] = 0.f;
            if (f < dim) {
                const int r = f / rowlen, rem = f % rowlen;   // rem = c*C + ch
                v[m] = (float)img[((int64_t)(y0 + r) * W + x0) * C + rem] * scale;
                sum += v[m];
            }
        }
        if (center) {
            const float mean = row16_sum_p(sum) / (float)dim;
#pragma unroll
            for (int m = 0; m < MAXF; ++m) v[m] -= mean;
        }
        if (normalize) {
            float ss = 0.f;
#pragma unroll
            for (int m = 0; m < MAXF; ++m)
                if (q + 16 * m < dim) ss = fmaf(v[m], v[m], ss);
            const float nrm = sqrtf(row16_sum_p(ss)) + EPS64_F;
#pragma unroll
            for (int m = 0; m < MAXF; ++m) v[m] = v[m] / nrm;
        }
#pragma unroll
        for (int m = 0; m < MAXF; ++m) {
            const int f = q + 16 * m;
            if (f < dim) X[p * ldx + f] = v[m];
        }
    }
}

template <typename PIX>
static int launch_grid(const PIX* img, int H, int W, int C, int patch, int step, float scale, int center, int normalize,
                       float* X, int64_t ldx, hipStream_t stream) {
    const int n_ph = (H - patch) / step + 1, n_pw = (W - patch) / step + 1;
    const int64_t n_patches = (int64_t)n_ph * n_pw;
    const int dim = patch * patch * C;
    int64_t blocks = (n_patches + 15) / 16;
    const int64_t cap = (int64_t)num_cus() * 16;
    if (blocks > cap) blocks = cap;
    const dim3 g((unsigned)blocks), b(256);
    if (dim <= 64)
        hipLaunchKernelGGL((grid_patches_kernel<PIX, 4>), g, b, 0, stream, img, H, W, C, patch, step, n_pw, n_patches, scale, center, normalize, X, ldx);
    else if (dim <= 256)
        hipLaunchKernelGGL((grid_patches_kernel<PIX, 16>), g, b, 0, stream, img, H, W, C, patch, step, n_pw, n_patches, scale, center, normalize, X, ldx);
    else if (dim <= 1024)
        hipLaunchKernelGGL((grid_patches_kernel<PIX, 64>), g, b, 0, stream, img, H, W, C, patch, step, n_pw, n_patches, scale, center, normalize, X, ldx);
    else {
        set_error("grid_patches: patch dimension %d > 1024 not supported", dim);
        return LYS_ENOSUP;
    }
    LYS_LAUNCH_CHECK();
    return LYS_OK;
}

int grid_patches(const void* img, int dtype, int H, int W, int C, int patch, int step, float scale, int center,
                 int normalize, float* X, int64_t ldx, hipStream_t stream) {
    if (H < patch || W < patch || patch < 1 || step < 1 || C < 1) {
        set_error("grid_patches: bad geometry H=%d W=%d C=%d patch=%d step=%d", H, W, C, patch, step);
        return LYS_EINVAL;
    }
    if (dtype == 0) return launch_grid(static_cast<const unsigned char*>(img), H, W, C, patch, step, scale, center, normalize, X, ldx, stream);
    if (dtype == 1) return launch_grid(static_cast<const float*>(img), H, W, C, patch, step, scale, center, normalize, X, ldx, stream);
    set_error("grid_patches: dtype %d (0 = uint8, 1 = float32)", dtype);
    return LYS_EINVAL;
}

// in-place per-signal preprocessing of an existing signal-major matrix (one wave per signal)
__global__ __launch_bounds__(256) void preproc_kernel(float* __restrict__ X, int64_t ldx, int n, int64_t N, float scale,
                                                      int center, int normalize) {
    const int lane = threadIdx.x & 63;
    for (int64_t s = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); s < N; s += (int64_t)gridDim.x * 4) {
        float sum = 0.f;
        for (int f = lane; f < n; f += 64) sum += X[s * ldx + f] * scale;
        const float mean = center ? wave_sum_f(sum) / (float)n : 0.f;
        float ss = 0.f;
        for (int f = lane; f < n; f += 64) {
            const float v = X[s * ldx + f] * scale - mean;
            ss = fmaf(v, v, ss);
        }
        const float nrm = normalize ? sqrtf(wave_sum_f(ss)) + EPS64_F : 1.f;
        for (int f = lane; f < n; f += 64) {
            const float v = X[s * ldx + f] * scale - mean;
            X[s * ldx + f] = normalize ? v / nrm : v;
        }
    }
}

int preproc_signals(float* X, int64_t ldx, int n, int64_t N, float scale, int center, int normalize,
                    hipStream_t stream) {
    if (N <= 0) return LYS_OK;
    int64_t blocks = (N + 3) / 4;
    const int64_t cap = (int64_t)num_cus() * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(preproc_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, X, ldx, n, N, scale, center,
                       normalize);
    LYS_LAUNCH_CHECK();
    return LYS_OK;
}

// ---------------------------------------------------------------------------------------------
// Dataset-level preprocessing (lyssa/feature_extract/preproc.py:18-31,55-62,77-78): per-FEATURE mean / std over all
// signals ('global_centering', 'global_standarization') and ZCA 'whitening'.  All HBM-rate streaming passes over the
// signal-major matrix; the n x n eigen-decomposition of ZCA stays on the host (scipy eigh, like the reference).
// ---------------------------------------------------------------------------------------------
// sum and sum of squares of every feature, fp64 (one wave walks rows, lane = feature % 64; per-workgroup LDS reduce)
__global__ __launch_bounds__(256) void feature_stats_kernel(const float* __restrict__ X, int64_t ldx, int n, int64_t N,
                                                            double* __restrict__ sum, double* __restrict__ sumsq) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int f0 = 0; f0 < n; f0 += 64) {
        const int f = f0 + lane;
        double s = 0.0, q = 0.0;
        if (f < n) {
            for (int64_t i = (int64_t)blockIdx.x * 4 + wid; i < N; i += (int64_t)gridDim.x * 4) {
                const double v = (double)X[i * ldx + f];
                s += v;
                q = fma(v, v, q);
            }
            atomicAdd(sum + f, s);
            atomicAdd(sumsq + f, q);
        }
    }
}

// X[i][f] = (X[i][f] - shift[f]) * scale[f]
__global__ __launch_bounds__(256) void feature_affine_kernel(float* __restrict__ X, int64_t ldx, int n, int64_t N,
                                                             const float* __restrict__ shift,
                                                             const float* __restrict__ scale) {
    const int64_t tot = N * (int64_t)n;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < tot; t += (int64_t)gridDim.x * 256) {
        const int64_t i = t / n;
        const int f = (int)(t - i * n);
        X[i * ldx + f] = (X[i * ldx + f] - shift[f]) * scale[f];
    }
}

// C += X' X restricted to the 64 x 64 block (bi, bj) of features, over this workgroup's slab of signals: the slab is
// staged 64 rows at a time in LDS, the four waves own the four 32 x 32 MFMA tiles of the block
// (v_mfma_f32_32x32x2_f32: A[f][i] and B[i][g] both read from the same staged rows), fp64 atomics at the end.
__global__ __launch_bounds__(256) void covariance_kernel(const float* __restrict__ X, int64_t ldx, int n, int64_t N,
                                                         int64_t rows_per_wg, double* __restrict__ C) {
    __shared__ float s_a[64][65], s_b[64][65];
    const int nb = (n + 63) / 64;
    int bi = 0, bj = blockIdx.y;  // blockIdx.y enumerates the upper-triangular blocks bi <= bj
    while (bj >= nb - bi) {
        bj -= nb - bi;
        ++bi;
    }
    bj += bi;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int ti = wid >> 1, tj = wid & 1;  // this wave's 32 x 32 tile inside the block
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_wg;
    const int64_t r1 = (r0 + rows_per_wg < N) ? r0 + rows_per_wg : N;
    using f16v = __attribute__((ext_vector_type(16))) float;
    f16v acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int64_t c0 = r0; c0 < r1; c0 += 64) {
        __syncthreads();
        for (int e = tid; e < 64 * 64; e += 256) {
            const int i = e >> 6, f = e & 63;
            const int64_t row = c0 + i;
            const int fa = bi * 64 + f, fb = bj * 64 + f;
            s_a[i][f] = (row < r1 && fa < n) ? X[row * ldx + fa] : 0.f;
            s_b[i][f] = (row < r1 && fb < n) ? X[row * ldx + fb] : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int i = 0; i < 64; i += 2) {
            const int kk = i + (lane >> 5);
            const float av = s_a[kk][32 * ti + (lane & 31)];
            const float bv = s_b[kk][32 * tj + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r >> 2) * 8 + (lane >> 5) * 4 + (r & 3), col = lane & 31;
        const int f = bi * 64 + 32 * ti + row, g = bj * 64 + 32 * tj + col;
        if (f < n && g < n) {
            atomicAdd(C + (int64_t)f * n + g, (double)acc[r]);
            if (bi != bj) atomicAdd(C + (int64_t)g * n + f, (double)acc[r]);
        }
    }
}

int feature_stats(const float* X, int64_t ldx, int n, int64_t N, double* sum, double* sumsq, hipStream_t stream) {
    LYS_CHECK_HIP(hipMemsetAsync(sum, 0, (size_t)n * sizeof(double), stream));
    LYS_CHECK_HIP(hipMemsetAsync(sumsq, 0, (size_t)n * sizeof(double), stream));
    if (N <= 0) return LYS_OK;
    int64_t blocks = (N + 3) / 4;
    const int64_t cap = (int64_t)num_cus() * 4;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(feature_stats_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, X, ldx, n, N, sum, sumsq);
    LYS_LAUNCH_CHECK();
    return LYS_OK;
}

int feature_affine(float* X, int64_t ldx, int n, int64_t N, const float* shift, const float* scale, hipStream_t stream) {
    if (N <= 0) return LYS_OK;
    int64_t blocks = (N * n + 255) / 256;
    const int64_t cap = (int64_t)num_cus() * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(feature_affine_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, X, ldx, n, N, shift, scale);
    LYS_LAUNCH_CHECK();
    return LYS_OK;
}

int covariance(const float* X, int64_t ldx, int n, int64_t N, double* C, hipStream_t stream) {
    LYS_CHECK_HIP(hipMemsetAsync(C, 0, (size_t)n * n * sizeof(double), stream));
    if (N <= 0) return LYS_OK;
    const int nb = (n + 63) / 64;
    int64_t wgs = (int64_t)num_cus() * 2;
    int64_t rows = (N + wgs - 1) / wgs;
    rows = ((rows + 63) / 64) * 64;
    wgs = (N + rows - 1) / rows;
    hipLaunchKernelGGL(covariance_kernel, dim3((unsigned)wgs, nb * (nb + 1) / 2), dim3(256), 0, stream, X, ldx, n, N, rows, C);
    LYS_LAUNCH_CHECK();
    return LYS_OK;
}

// ---------------------------------------------------------------------------------------------
// Consumer right after the encode: ScSPM spatial-pyramid max pooling of |z|
// (lyssa/feature_extract/spatial_pyramid.py:57-97 with pooling.py:4-7) straight from the sparse triplet:
// out[cell][atom] = max over the patches of the cell of |coef|.  |x| >= 0, so the float max is an unsigned max on
// the bit patterns (global atomicMax).  `cell` holds one cell id per (level, patch), already offset per level.
// ---------------------------------------------------------------------------------------------
__global__ void pool_max_abs_kernel(const int32_t* __restrict__ idx, const float* __restrict__ coef,
                                    const int32_t* __restrict__ nnz, int k, int64_t N,
                                    const int32_t* __restrict__ cell, int L, int K, unsigned* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N * k) return;
    const int64_t s = t / k;
    const int j = (int)(t % k);
    if (j >= nnz[s]) return;
    const int a = idx[t];
    const unsigned v = __builtin_bit_cast(unsigned, fabsf(coef[t]));
    if (v == 0u || a < 0 || a >= K) return;
    for (int l = 0; l < L; ++l) {
        const int c = cell[(int64_t)l * N + s];
        if (c >= 0) atomicMax(&out[(int64_t)c * K + a], v);
    }
}

// x / (||x|| + eps) per row (l2_normalizer, feature_extract/preproc.py:8-16); one wave per row
__global__ __launch_bounds__(64) void l2_rows_kernel(float* __restrict__ M, int cols) {
    const int lane = threadIdx.x;
    float* row = M + (int64_t)blockIdx.x * cols;
    float ss = 0.f;
    for (int c = lane; c < cols; c += 64) ss = fmaf(row[c], row[c], ss);
    const float nrm = sqrtf(wave_sum_f(ss)) + EPS64_F;
    for (int c = lane; c < cols; c += 64) row[c] = row[c] / nrm;
}

int pool_max_abs(const int32_t* idx, const float* coef, const int32_t* nnz, int k, int64_t N, const int32_t* cell, int L,
                 int K, int n_cells, float* out, int l2_normalize, hipStream_t stream) {
    LYS_CHECK_HIP(hipMemsetAsync(out, 0, (size_t)n_cells * K * sizeof(float), stream));
    const int64_t tot = N * k;
    if (tot > 0) {
        hipLaunchKernelGGL(pool_max_abs_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream, idx, coef, nnz,
                           k, N, cell, L, K, reinterpret_cast<unsigned*>(out));
        LYS_LAUNCH_CHECK();
    }
    if (l2_normalize && n_cells > 0) {
        hipLaunchKernelGGL(l2_rows_kernel, dim3(n_cells), dim3(64), 0, stream, out, K);
        LYS_LAUNCH_CHECK();
    }
    return LYS_OK;
}

}  // namespace lys
