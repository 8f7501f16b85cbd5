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
