// l1-penalised sparse coding on the Gram matrix: min_a 0.5 ||x - D a||^2 + lambda ||a||_1 per signal.
//
// Reference: lyssa/sparse_coding.py:487-509 (`lasso`, a wrapper of spams.lasso(mode=2, lambda2=0); SPAMS solves it
// with a LARS homotopy).  SPAMS is not vendored, so parity is on the optimisation problem itself (unique minimiser in
// general position): the kernel runs greedy (Gauss-Southwell) coordinate descent on the correlations
// c = D'x - G a, the same data the Batch-OMP kernel works on:
//     v_j = c_j + G_jj a_j;  a_j^new = soft(v_j, lambda) / G_jj;  pick j with the largest |a_j^new - a_j|;
//     a_j += delta;  c -= delta * G[j, :]      (one Gram row per step, like one Batch-OMP step)
// until the largest change is below tol * max|D'x| (then every KKT condition holds to that accuracy).
// One workgroup per signal, 16 atoms per lane (fewer for small dictionaries): a, c, diag(G) stay in registers.
#include "common.h"

namespace lys {

template <int R>
struct LLay {  // element e of lane-register r, for T threads: float4-coalesced rows when R >= 4
    static constexpr int VEC = (R >= 4) ? 4 : R;
    __device__ static __forceinline__ int elem(int r, int t, int T) { return (r / VEC) * (VEC * T) + t * VEC + (r % VEC); }
};

template <int R>
__device__ __forceinline__ void load_vec(const float* __restrict__ row, int t, int T, float (&v)[R]) {
    using L = LLay<R>;
    if constexpr (L::VEC == 4) {
#pragma unroll
        for (int g = 0; g < R / 4; ++g) {
            const float4 x = *reinterpret_cast<const float4*>(row + g * 4 * T + t * 4);
            v[4 * g + 0] = x.x;
            v[4 * g + 1] = x.y;
            v[4 * g + 2] = x.z;
            v[4 * g + 3] = x.w;
        }
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = row[L::elem(r, t, T)];
    }
}

struct Best {
    float score;
    int e;
    float delta;
};

__device__ __forceinline__ Best better(const Best& a, const Best& b) {  // larger score; ties -> lower atom index
    return (b.score > a.score || (b.score == a.score && b.e < a.e)) ? b : a;
}

__device__ __forceinline__ Best wave_best(Best x) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        Best o;
        o.score = __shfl_xor(x.score, off, 64);
        o.e = __shfl_xor(x.e, off, 64);
        o.delta = __shfl_xor(x.delta, off, 64);
        x = better(x, o);
    }
    return x;
}

constexpr int LASSO_MAX_WAVES = 16;

template <int R>
__global__ __launch_bounds__(64 * LASSO_MAX_WAVES) void lasso_cd_kernel(const float* __restrict__ alpha0,
                                                                         const float* __restrict__ G, int Kp, int K,
                                                                         float lambda, float tol_rel, int max_steps,
                                                                         int kcap, int64_t N, int32_t* __restrict__ idx,
                                                                         float* __restrict__ coef,
                                                                         int32_t* __restrict__ nnz,
                                                                         int32_t* __restrict__ steps_out) {
    using L = LLay<R>;
    __shared__ float s_score[LASSO_MAX_WAVES], s_delta[LASSO_MAX_WAVES];
    __shared__ int s_e[LASSO_MAX_WAVES], s_cnt[LASSO_MAX_WAVES];
    const int64_t sig = blockIdx.x;
    if (sig >= N) return;
    const int t = threadIdx.x, T = blockDim.x, W = T >> 6, wid = t >> 6, lane = t & 63;
    float c[R], a[R], gd[R], ginv[R];
    load_vec<R>(alpha0 + sig * Kp, t, T, c);
    float amax = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int e = L::elem(r, t, T);
        a[r] = 0.f;
        gd[r] = (e < K) ? G[(int64_t)e * Kp + e] : 0.f;
        ginv[r] = (gd[r] > 0.f) ? 1.f / gd[r] : 0.f;
        amax = fmaxf(amax, fabsf(c[r]));
    }
    amax = wave_max_f(amax);
    if (W > 1) {
        if (lane == 0) s_score[wid] = amax;
        __syncthreads();
        for (int w = 0; w < W; ++w) amax = fmaxf(amax, s_score[w]);
        __syncthreads();
    }
    const float tol_abs = tol_rel * amax;
    int steps = 0, total = 0;
    // Rounds of [coordinate descent to the stopping rule] + [compaction of the non-zeros] + [refresh of c from scratch].
    // c -= delta * G[j,:] accumulates fp32 rounding over hundreds of steps; recomputing c = D'x - G a from the compacted
    // code (one Gram row per non-zero) and resuming removes that drift.  The kernel returns when a round that started
    // from fresh correlations needs no step at all (or at max_steps / after MAX_REFRESH rounds).
    constexpr int MAX_REFRESH = 8;
    for (int round = 0;; ++round) {
        const int steps_before = steps;
        while (steps < max_steps) {
            Best b{-1.f, 0x7fffffff, 0.f};
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float v = fmaf(gd[r], a[r], c[r]);
                const float s = copysignf(fmaxf(fabsf(v) - lambda, 0.f), v) * ginv[r];
                const float d = s - a[r];
                const Best cand{fabsf(d), L::elem(r, t, T), d};
                b = better(b, cand);
            }
            b = wave_best(b);
            if (W > 1) {
                if (lane == 0) {
                    s_score[wid] = b.score;
                    s_e[wid] = b.e;
                    s_delta[wid] = b.delta;
                }
                __syncthreads();
                b = Best{s_score[0], s_e[0], s_delta[0]};
                for (int w = 1; w < W; ++w) b = better(b, Best{s_score[w], s_e[w], s_delta[w]});
                __syncthreads();
            }
            if (!(b.score > tol_abs)) break;
            float g[R];
            load_vec<R>(G + (int64_t)b.e * Kp, t, T, g);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (L::elem(r, t, T) == b.e) a[r] += b.delta;
                c[r] = fmaf(-b.delta, g[r], c[r]);
            }
            ++steps;
        }
        // ---- snap: a coefficient whose own update target is exactly zero (it sits below the stopping tolerance) is
        // set to zero, so the returned support is the support of the minimiser and not "zero + 1e-7"
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float v = fmaf(gd[r], a[r], c[r]);
            if (a[r] != 0.f && !(fabsf(v) > lambda)) {
                c[r] = v;  // own-coordinate part of the correlation update (the other coordinates move by < tol)
                a[r] = 0.f;
            }
        }
        // ---- compact the non-zeros: (thread, register) order, at most kcap entries
        int cnt = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) cnt += (a[r] != 0.f) ? 1 : 0;
        int incl = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        int base = 0;
        total = __shfl(incl, 63, 64);
        if (W > 1) {
            __syncthreads();
            if (lane == 63) s_cnt[wid] = incl;
            __syncthreads();
            total = 0;
            for (int w = 0; w < W; ++w) {
                if (w < wid) base += s_cnt[w];
                total += s_cnt[w];
            }
        }
        int pos = base + incl - cnt;
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (a[r] != 0.f) {
                if (pos < kcap) {
                    idx[sig * kcap + pos] = L::elem(r, t, T);
                    coef[sig * kcap + pos] = a[r];
                }
                ++pos;
            }
        const bool fresh_and_quiet = (round > 0) && (steps == steps_before);
        if (fresh_and_quiet || steps >= max_steps || round >= MAX_REFRESH || total > kcap || total == 0) break;
        // ---- refresh: c = alpha0 - sum_e coef_e * G[idx_e, :] from the list just written
        __threadfence_block();
        __syncthreads();
        load_vec<R>(alpha0 + sig * Kp, t, T, c);
        for (int e = 0; e < total; ++e) {
            const int j = idx[sig * kcap + e];
            const float aj = coef[sig * kcap + e];
            float g[R];
            load_vec<R>(G + (int64_t)j * Kp, t, T, g);
#pragma unroll
            for (int r = 0; r < R; ++r) c[r] = fmaf(-aj, g[r], c[r]);
        }
        __syncthreads();  // the list is overwritten by the next compaction
    }
    for (int p = total + t; p < kcap; p += T) {  // unused slots: (-1, 0) like the other encoders
        idx[sig * kcap + p] = -1;
        coef[sig * kcap + p] = 0.f;
    }
    if (t == 0) {
        nnz[sig] = total < kcap ? total : kcap;
        if (steps_out) steps_out[sig] = (total > kcap) ? -(steps + 1) : steps;  // negative: support truncated to kcap
    }
}

int lasso_from_alpha0(const float* alpha0, const float* G, int Kp, int K, float lambda, float tol, int max_steps,
                      int kcap, int64_t N, int32_t* idx, float* coef, int32_t* nnz, int32_t* steps,
                      hipStream_t stream) {
    if (N <= 0) return LYS_OK;
    if (N > 0x7fffffffLL) {
        set_error("lasso: too many signals in one tile");
        return LYS_ENOSUP;
    }
    const dim3 grid((unsigned)N);
#define LYS_LASSO(RR, TT)                                                                                              \
    hipLaunchKernelGGL(lasso_cd_kernel<RR>, grid, dim3(TT), 0, stream, alpha0, G, Kp, K, lambda, tol, max_steps, kcap, \
                       N, idx, coef, nnz, steps)
    if (Kp == 64) LYS_LASSO(1, 64);
    else if (Kp == 128) LYS_LASSO(2, 64);
    else if (Kp == 256) LYS_LASSO(4, 64);
    else if (Kp == 512) LYS_LASSO(8, 64);
    else if (Kp % 1024 == 0 && Kp / 16 <= 64 * LASSO_MAX_WAVES) LYS_LASSO(16, Kp / 16);
    else {
        set_error("lasso: padded atom count %d not supported (max %d)", Kp, 1024 * LASSO_MAX_WAVES);
        return LYS_ENOSUP;
    }
#undef LYS_LASSO
    LYS_LAUNCH_CHECK();
    return LYS_OK;
}

}  // namespace lys
