// Dictionary-update kernels for gfx950: sparse residual / error, atom-major index of the non-zeros,
// and the approximate K-SVD atom update (lyssa/dict_learning/ksvd.py:98-126).
//
// Data layout (HBM-bound byte shuffling, no MFMA here): residual R is signal-major [N][ldr] so that one
// signal is one contiguous 4*n-byte row; a 16-lane DPP row ("team") owns one signal at a time and moves it
// with one dwordx4 per lane per 64 features.  Per (atom, signal) non-zero the sweep reads and writes the
// residual row once per phase: algorithmic traffic 2*4n B (phase 1 read + phase 2 read/write = 3*4n B moved).
#include <stdlib.h>

#include "common.h"
#include <vector>
#include <utility>
#include <algorithm>

namespace lys {

__device__ __forceinline__ float row16_sum(float x) {  // sum over the 16 lanes of a DPP row, in every lane
    x += dpp_f<0xB1>(x);
    x += dpp_f<0x4E>(x);
    x += dpp_f<0x124>(x);
    x += dpp_f<0x128>(x);
    return x;
}

// lane j (a constant after unrolling) of every 16-lane row, in all lanes of that row
__device__ __forceinline__ int csr_row_bcast_res(int x, int j) {
    switch (j & 15) {
        case 0: return __builtin_amdgcn_update_dpp(0, x, 0x150, 0xf, 0xf, true);
        case 1: return __builtin_amdgcn_update_dpp(0, x, 0x151, 0xf, 0xf, true);
        case 2: return __builtin_amdgcn_update_dpp(0, x, 0x152, 0xf, 0xf, true);
        case 3: return __builtin_amdgcn_update_dpp(0, x, 0x153, 0xf, 0xf, true);
        case 4: return __builtin_amdgcn_update_dpp(0, x, 0x154, 0xf, 0xf, true);
        case 5: return __builtin_amdgcn_update_dpp(0, x, 0x155, 0xf, 0xf, true);
        case 6: return __builtin_amdgcn_update_dpp(0, x, 0x156, 0xf, 0xf, true);
        case 7: return __builtin_amdgcn_update_dpp(0, x, 0x157, 0xf, 0xf, true);
        case 8: return __builtin_amdgcn_update_dpp(0, x, 0x158, 0xf, 0xf, true);
        case 9: return __builtin_amdgcn_update_dpp(0, x, 0x159, 0xf, 0xf, true);
        case 10: return __builtin_amdgcn_update_dpp(0, x, 0x15A, 0xf, 0xf, true);
        case 11: return __builtin_amdgcn_update_dpp(0, x, 0x15B, 0xf, 0xf, true);
        case 12: return __builtin_amdgcn_update_dpp(0, x, 0x15C, 0xf, 0xf, true);
        case 13: return __builtin_amdgcn_update_dpp(0, x, 0x15D, 0xf, 0xf, true);
        case 14: return __builtin_amdgcn_update_dpp(0, x, 0x15E, 0xf, 0xf, true);
        default: return __builtin_amdgcn_update_dpp(0, x, 0x15F, 0xf, 0xf, true);
    }
}

// ---------------------------------------------------------------------------------------------
// R = X - D Z  and  err += ||R||^2      (ksvd.py:103, dict_learning/utils.py:14-19)
// A 16-lane DPP row ("team") owns a signal: lane q moves features 64b + 4q .. +3 with one dwordx4 per 64 features
// (a 64-dim patch = one 256-B row per team), the k selected atoms are gathered as 256-B rows of the L2-resident
// dictionary.  Falls back to one feature per lane for n > 256 or unaligned rows.
// ---------------------------------------------------------------------------------------------
template <int FB>
__global__ __launch_bounds__(256) void residual_team_kernel(const float* __restrict__ X, int64_t ldx,
                                                            const float* __restrict__ D, int ldd, int n, int k,
                                                            int64_t N, const int32_t* __restrict__ idx,
                                                            const float* __restrict__ coef,
                                                            const int32_t* __restrict__ nnz, float* __restrict__ R,
                                                            int64_t ldr, double* __restrict__ err) {
    __shared__ double s_part[16];
    const int team = threadIdx.x >> 4, q = threadIdx.x & 15;
    const int64_t gteam = (int64_t)blockIdx.x * 16 + team, nteams = (int64_t)gridDim.x * 16;
    double acc = 0.0;
    for (int64_t s = gteam; s < N; s += nteams) {
        const int m = nnz[s];
        float4 r[FB];
#pragma unroll
        for (int b = 0; b < FB; ++b) {
            const int f = 64 * b + 4 * q;
            r[b] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f < n) {
                if (f + 3 < n) {
                    r[b] = *reinterpret_cast<const float4*>(X + s * ldx + f);
                } else {  // ragged tail of an n that is not a multiple of 4
                    const float* p = X + s * ldx + f;
                    r[b].x = p[0];
                    if (f + 1 < n) r[b].y = p[1];
                    if (f + 2 < n) r[b].z = p[2];
                }
            }
        }
        for (int j = 0; j < m; ++j) {
            const int a = idx[s * k + j];
            const float c = coef[s * k + j];
#pragma unroll
            for (int b = 0; b < FB; ++b) {
                const int f = 64 * b + 4 * q;
                if (f < ldd) {  // packed dictionary: ldd is a multiple of 8, columns >= n are zero
                    const float4 d = *reinterpret_cast<const float4*>(D + (int64_t)a * ldd + f);
                    r[b].x = fmaf(-c, d.x, r[b].x);
                    r[b].y = fmaf(-c, d.y, r[b].y);
                    r[b].z = fmaf(-c, d.z, r[b].z);
                    r[b].w = fmaf(-c, d.w, r[b].w);
                }
            }
        }
        float e2 = 0.f;
#pragma unroll
        for (int b = 0; b < FB; ++b) {
            const int f = 64 * b + 4 * q;
            if (f < n) {
                if (R) *reinterpret_cast<float4*>(R + s * ldr + f) = r[b];  // padded columns of R receive zeros
                e2 = fmaf(r[b].x, r[b].x, e2);
                e2 = fmaf(r[b].y, r[b].y, e2);
                e2 = fmaf(r[b].z, r[b].z, e2);
                e2 = fmaf(r[b].w, r[b].w, e2);
            }
        }
        acc += (double)row16_sum(e2);
    }
    if (err) {
        if (q == 0) s_part[team] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            double tot = 0.0;
#pragma unroll
            for (int t = 0; t < 16; ++t) tot += s_part[t];
            atomicAdd(err, tot);
        }
    }
}

// The same for n <= 64, k <= 16 (the block sweep's shapes), one signal per team and iteration with NOTHING dependent inside the
// iteration but the dictionary rows: lane q of the team holds slot q of the support (one load per lane instead of k broadcast
// loads one after the other), the row, the support and the count of the team's NEXT signal are requested before the current
// one is processed, and the k dictionary rows of a signal are requested together (slots past the count read atom 0 with a zero
// coefficient).  residual_team_kernel walks idx[j] -> D[idx[j]] k times per signal, 32 signals per team one after the other.
__global__ __launch_bounds__(256) void residual_team16_kernel(const float* __restrict__ X, int64_t ldx,
                                                              const float* __restrict__ D, int ldd, int n, int k,
                                                              int64_t N, const int32_t* __restrict__ idx,
                                                              const float* __restrict__ coef,
                                                              const int32_t* __restrict__ nnz, float* __restrict__ R,
                                                              int64_t ldr, double* __restrict__ err) {
    __shared__ double s_part[16];
    const int team = threadIdx.x >> 4, q = threadIdx.x & 15;
    const int64_t gteam = (int64_t)blockIdx.x * 16 + team, nteams = (int64_t)gridDim.x * 16;
    const int f = 4 * q;
    double acc = 0.0;
    float4 rN = make_float4(0.f, 0.f, 0.f, 0.f);
    int aN = 0, mN = 0;
    float cN = 0.f;
    auto fetch = [&](int64_t s) __attribute__((always_inline)) {
        mN = nnz[s];
        const int64_t off = s * k + ((q < k) ? q : k - 1);
        aN = idx[off];
        cN = coef[off];
        rN = make_float4(0.f, 0.f, 0.f, 0.f);
        if (f + 3 < n) {
            rN = *reinterpret_cast<const float4*>(X + s * ldx + f);
        } else if (f < n) {  // ragged tail of an n that is not a multiple of 4
            const float* p = X + s * ldx + f;
            rN.x = p[0];
            if (f + 1 < n) rN.y = p[1];
            if (f + 2 < n) rN.z = p[2];
        }
    };
    if (gteam < N) fetch(gteam);
    for (int64_t s = gteam; s < N; s += nteams) {
        float4 r = rN;
        const int m = mN;
        const int a = (q < m && q < k) ? aN : 0;
        const float c = (q < m && q < k) ? cN : 0.f;
        fetch((s + nteams < N) ? s + nteams : s);
        float4 d[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            d[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < k) {  // uniform
                const int aj = csr_row_bcast_res(a, j);
                if (f < ldd) d[j] = *reinterpret_cast<const float4*>(D + (int64_t)aj * ldd + f);
            }
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (j < k) {
                const float cj = __builtin_bit_cast(float, csr_row_bcast_res(__builtin_bit_cast(int, c), j));
                r.x = fmaf(-cj, d[j].x, r.x);
                r.y = fmaf(-cj, d[j].y, r.y);
                r.z = fmaf(-cj, d[j].z, r.z);
                r.w = fmaf(-cj, d[j].w, r.w);
            }
        }
        float e2 = 0.f;
        if (f < n) {
            if (R) *reinterpret_cast<float4*>(R + s * ldr + f) = r;  // padded columns of R receive zeros
            e2 = fmaf(r.x, r.x, e2);
            e2 = fmaf(r.y, r.y, e2);
            e2 = fmaf(r.z, r.z, e2);
            e2 = fmaf(r.w, r.w, e2);
        }
        acc += (double)row16_sum(e2);
    }
    if (err) {
        if (q == 0) s_part[team] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            double tot = 0.0;
#pragma unroll
            for (int t = 0; t < 16; ++t) tot += s_part[t];
            atomicAdd(err, tot);
        }
    }
}

__global__ __launch_bounds__(256) void residual_kernel(const float* __restrict__ X, int64_t ldx,
                                                       const float* __restrict__ D, int ldd, int n, int k, int64_t N,
                                                       const int32_t* __restrict__ idx,
                                                       const float* __restrict__ coef,
                                                       const int32_t* __restrict__ nnz, float* __restrict__ R,
                                                       int64_t ldr, double* __restrict__ err) {
    __shared__ double s_part[4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int64_t wave = (int64_t)blockIdx.x * 4 + wid, nwaves = (int64_t)gridDim.x * 4;
    double acc = 0.0;
    for (int64_t s = wave; s < N; s += nwaves) {
        const int m = nnz[s];
        float e2 = 0.f;
        for (int f = lane; f < n; f += 64) {
            float r = X[s * ldx + f];
            for (int j = 0; j < m; ++j) {
                const int a = idx[s * k + j];
                r = fmaf(-coef[s * k + j], D[(int64_t)a * ldd + f], r);
            }
            if (R) R[s * ldr + f] = r;
            e2 = fmaf(r, r, e2);
        }
        acc += (double)wave_sum_f(e2);
    }
    if (err) {
        if (lane == 0) s_part[wid] = acc;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(err, (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]));
    }
}

int residual(const float* X, int64_t ldx, const float* D, int n, int K, int k, int64_t N, const int32_t* idx,
             const float* coef, const int32_t* nnz, float* R, int64_t ldr, double* err, hipStream_t stream) {
    if (N <= 0) return LYS_OK;
    const int ldd = padded_features(n);
    const bool aligned = ((ldx & 3) == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0) &&
                         (R == nullptr || (((ldr & 3) == 0) && ((reinterpret_cast<uintptr_t>(R) & 15) == 0)));
    const int64_t cap = (int64_t)num_cus() * 8;
    if (aligned && n <= 256) {
        int64_t blocks = (N + 15) / 16;
        if (blocks > cap) blocks = cap;
        const dim3 g((unsigned)blocks), b(256);
        if (n <= 64 && k <= 16)
            hipLaunchKernelGGL(residual_team16_kernel, g, b, 0, stream, X, ldx, D, ldd, n, k, N, idx, coef, nnz, R, ldr, err);
        else if (n <= 64)
            hipLaunchKernelGGL(residual_team_kernel<1>, g, b, 0, stream, X, ldx, D, ldd, n, k, N, idx, coef, nnz, R, ldr, err);
        else if (n <= 128)
            hipLaunchKernelGGL(residual_team_kernel<2>, g, b, 0, stream, X, ldx, D, ldd, n, k, N, idx, coef, nnz, R, ldr, err);
        else
            hipLaunchKernelGGL(residual_team_kernel<4>, g, b, 0, stream, X, ldx, D, ldd, n, k, N, idx, coef, nnz, R, ldr, err);
        LYS_LAUNCH_CHECK();
        return LYS_OK;
    }
    int64_t blocks = (N + 3) / 4;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(residual_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, X, ldx, D, ldd, n, k, N, idx,
                       coef, nnz, R, ldr, err);
    LYS_LAUNCH_CHECK();
    return LYS_OK;
}

// ---------------------------------------------------------------------------------------------
// CSR by atom: deterministic counting sort of (signal, slot) entries by atom id.
// The signal range is cut into T chunks; ONE wave walks a chunk signal by signal (so the order inside an
// atom is the signal order), counting into LDS.  Pass A counts, pass B scans (atom-major, chunk-minor),
// pass C scatters.  Entries with slot >= nnz or coef == 0 are dropped (omega_k = X[k,:] != 0, ksvd.py:111).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void csr_count_or_fill_kernel(const int32_t* __restrict__ idx,
                                                               const float* __restrict__ coef,
                                                               const int32_t* __restrict__ nnz, int K, int k, int64_t N,
                                                               int T, int64_t S, int32_t* __restrict__ counts,
                                                               const int32_t* __restrict__ row_ptr,
                                                               int32_t* __restrict__ entry, int fill,
                                                               int32_t* __restrict__ emeta = nullptr,
                                                               float* __restrict__ ecoef = nullptr, int logb = 0,
                                                               int32_t* __restrict__ cg_cur = nullptr,
                                                               int32_t* __restrict__ cg_entry = nullptr) {
    extern __shared__ int s_cnt[];
    const int lane = threadIdx.x;
    const int chunk = blockIdx.x;
    // counts is chunk-major [T][K]: a wave reads / writes its K counters as one contiguous run
    for (int a = lane; a < K; a += 64) s_cnt[a] = fill ? row_ptr[a] + counts[(int64_t)chunk * K + a] : 0;
    __syncthreads();
    const int64_t s0 = (int64_t)chunk * S;
    const int64_t s1 = (s0 + S < N) ? s0 + S : N;
    if (k <= 64) {
        // the usual case: one lane per slot.  The loads of U signals are issued together (slots past nnz hold idx = -1,
        // coef = 0 and filter themselves out, so nnz is not on the dependency chain); the LDS adds follow signal by
        // signal: atoms inside one signal are distinct, so the adds of one step never collide, and the next signal's
        // adds are issued after these (same wave, in order) => signal order is preserved.
        constexpr int U = 8;
        for (int64_t sb = s0; sb < s1; sb += U) {
            int av[U];
            float cv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t s = sb + u;
                const bool in = (s < s1) && (lane < k);
                av[u] = in ? idx[s * k + lane] : -1;
                cv[u] = in ? coef[s * k + lane] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool live = cv[u] != 0.f && av[u] >= 0 && av[u] < K;
                int meta = lane;
                if (logb > 0) {
                    // block sweep (ksvd_block.hip): entry = signal id + the entry's coefficient + its slot and flags
                    // about the signal's atoms: another atom in the same block of 2^logb atoms (bit 8), in the previous
                    // block (9), in the next (10); bit 11: this entry is the LEADER (smallest atom of its block in the
                    // signal).  Leaders of coupled signals that do not touch the previous block are also listed in a second
                    // index sorted by (block, in-block atom mask): their tuple moments are summed group by group.
                    const int a = live ? av[u] : -1;
                    const int blk = a >> logb;
                    const int bsz = 1 << logb;
                    unsigned msk = 0;
                    for (int j = 0; j < k; ++j) {  // wave-uniform loop over the signal's slots
                        const int aj = __builtin_amdgcn_readlane(a, j);
                        if (aj >= 0) {
                            const int bj = aj >> logb;
                            msk |= (bj == blk) ? (1u << (aj & (bsz - 1))) : 0u;
                            meta |= (bj == blk - 1) ? 0x200 : 0;
                            meta |= (bj == blk + 1) ? 0x400 : 0;
                        }
                    }
                    const bool coupled = (msk & (msk - 1)) != 0;
                    const bool leader = live && ((a & (bsz - 1)) == __ffs(msk) - 1);
                    meta |= coupled ? 0x100 : 0;
                    meta |= leader ? 0x800 : 0;
                    if (cg_cur && leader && coupled && !(meta & 0x200)) {
                        const int key = (blk << bsz) | (int)msk;
                        const int pos = atomicAdd(&cg_cur[key], 1);  // count pass: a counter; fill pass: the position
                        if (fill) cg_entry[pos] = (int32_t)(sb + u);
                    }
                }
                if (live) {
                    const int pos = atomicAdd(&s_cnt[av[u]], 1);
                    if (fill == 2) {
                        reinterpret_cast<int4*>(entry)[pos] = make_int4((int)(sb + u), meta, __builtin_bit_cast(int, cv[u]), 0);
                    } else if (fill) {
                        entry[pos] = (int32_t)((sb + u) * k + lane);
                    }
                }
            }
        }
    } else {
        for (int64_t s = s0; s < s1; ++s) {
            const int m = nnz[s];
            for (int j = lane; j < m; j += 64) {
                const int a = idx[s * k + j];
                const float c = coef[s * k + j];
                if (c != 0.f && a >= 0 && a < K) {
                    const int pos = atomicAdd(&s_cnt[a], 1);
                    if (fill) entry[pos] = (int32_t)(s * k + j);
                }
            }
        }
    }
    if (!fill) {
        __syncthreads();
        for (int a = lane; a < K; a += 64) counts[(int64_t)chunk * K + a] = s_cnt[a];
    }
}

// Block-sweep index, k <= 16: the same two passes with FOUR signals per wave instruction (one 16-lane DPP row per
// signal, lane = slot).  The kernel above spends its time in VALU instructions that use 10 of 64 lanes; the order of the
// signals inside an atom, which it preserves, does not matter to the block sweep (ksvd_block.hip walks contiguous
// chunks and sums with atomics).  Flags / coupled-leader index: see csr_count_or_fill_kernel.
template <int L>
__device__ __forceinline__ int csr_row_bcast(int x) {  // lane L of every 16-lane row, in all lanes of that row
    return __builtin_amdgcn_update_dpp(0, x, 0x150 + L, 0xf, 0xf, true);
}
// Predecessor of an entry for the LAZY block sweep (ksvd_block.hip): the signal's atoms in the last block BEFORE this
// entry's block -- the update of those atoms is still pending when the sweep reaches this entry.  pb = that block (-1:
// none), pa / ps / pc = atom, slot and coefficient of its first atom, pn = how many atoms of the signal it holds.
struct CsrPred {
    int pb, pa, ps, pn;
    float pc;
};
// Round 5b: the J loop only finds the LEADER slot of the pending block -- the maximum of a packed key (block << 3 | 7 - atom's
// position in its block: largest block below this entry's, then its smallest atom) with the slot that holds it -- and the
// leader's atom, coefficient and "its block holds several atoms of the signal" (= the leader lane's own in-block mask has two
// bits) are fetched from that lane afterwards (ds_bpermute).  The five per-slot selects of the tracked form (pn, pa, ps, pc, pb)
// were a quarter of the 340 VALU instructions both index passes spend per four signals.
template <int J>
__device__ __forceinline__ void csr_flag_step(int a, int blk, int k, int logb, unsigned& msk, int& meta, int& best, int& ps) {
    if constexpr (J < 16) {
        if (J < k) {  // uniform
            const int aj = csr_row_bcast<J>(a);
            const int bj = aj >> logb;
            const int lo = aj & ((1 << logb) - 1);
            const bool on = aj >= 0;
            msk |= (on && bj == blk) ? (1u << lo) : 0u;
            meta |= (on && bj == blk - 1) ? 0x200 : 0;
            meta |= (on && bj == blk + 1) ? 0x400 : 0;
            const int cand = (on && bj < blk) ? ((bj << 3) | (7 - lo)) : -1;
            ps = (cand > best) ? J : ps;
            best = (cand > best) ? cand : best;
            csr_flag_step<J + 1>(a, blk, k, logb, msk, meta, best, ps);
        }
    }
}
// flags of one entry (lane = slot of a 16-lane row): in-block mask, PREV / NEXT bits and the predecessor (see CsrPred)
__device__ __forceinline__ void csr_entry_flags(int a, float cv, int blk, int k, int logb, int lane, unsigned& msk, int& meta,
                                                CsrPred& pr) {
    int best = -1, ps = 63;
    csr_flag_step<0>(a, blk, k, logb, msk, meta, best, ps);
    // the leader lane of the pending block (same 16-lane row): its atom, coefficient and in-block mask
    const int src = ((lane & 48) | (ps & 15)) << 2;
    const int pa = __builtin_amdgcn_ds_bpermute(src, a);
    const float pc = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, cv)));
    const unsigned pm = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)msk);
    pr.pb = (best >= 0) ? (best >> 3) : -1;
    pr.ps = ps;
    pr.pa = pa;
    pr.pc = pc;
    pr.pn = (best >= 0) ? (((pm & (pm - 1u)) != 0u) ? 2 : 1) : 0;  // only "one" / "several" is used
}

template <bool FILL>
__global__ __launch_bounds__(64) void bksvd_index_kernel(const int32_t* __restrict__ idx, const float* __restrict__ coef,
                                                         int K, int k, int64_t N, int T, int64_t S,
                                                         int32_t* __restrict__ counts,
                                                         const int32_t* __restrict__ row_ptr,
                                                         int4* __restrict__ erec, int logb,
                                                         int32_t* __restrict__ cg_cur, int32_t* __restrict__ cg_entry) {
    extern __shared__ int s_cnt[];
    const int lane = threadIdx.x, row = lane >> 4, j = lane & 15;
    const int chunk = blockIdx.x;
    for (int a = lane; a < K; a += 64) s_cnt[a] = FILL ? row_ptr[a] + counts[(int64_t)chunk * K + a] : 0;
    __syncthreads();
    const int64_t s0 = (int64_t)chunk * S;
    const int64_t s1 = (s0 + S < N) ? s0 + S : N;
    const int bsz = 1 << logb;
    constexpr int U = 4;  // quads of signals in flight
    for (int64_t sb = s0; sb < s1; sb += 4 * U) {
        int av[U];
        float cv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t sig = sb + 4 * u + row;
            const bool in = (sig < s1) && (j < k);
            av[u] = in ? idx[sig * k + j] : -1;
            cv[u] = in ? coef[sig * k + j] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t sig = sb + 4 * u + row;
            const bool live = cv[u] != 0.f && av[u] >= 0 && av[u] < K;
            const int a = live ? av[u] : -1;
            const int blk = a >> logb;
            unsigned msk = 0;
            int meta = j;
            CsrPred pr{-1, -1, 63, 0, 0.f};
            csr_entry_flags(a, live ? cv[u] : 0.f, blk, k, logb, lane, msk, meta, pr);
            // bits 12-17: predecessor slot (63 = none), 18-30: predecessor atom (K <= 8192), 31: the predecessor block holds
            // several atoms of the signal (slow path); word 3 of the record: the predecessor's coefficient
            if (pr.pb >= 0 && K <= 8192) {
                meta |= (pr.ps & 63) << 12;
                meta |= (pr.pa & 0x1fff) << 18;
                meta |= (pr.pn > 1) ? (int)0x80000000 : 0;
            } else {
                meta |= 63 << 12;
            }
            const bool coupled = (msk & (msk - 1)) != 0;
            const bool leader = live && ((a & (bsz - 1)) == __ffs(msk) - 1);
            meta |= coupled ? 0x100 : 0;
            meta |= leader ? 0x800 : 0;
            // the group phase of X(c) also takes the UNCOUPLED entries whose pending block holds several atoms of the signal
            // (bit 31: they need the support, which that phase loads anyway; single-bit mask keys, empty before round 5b) --
            // as queued slow-path entries of the entry walk they kept every walking workgroup for a barrier-separated drain
            // of ~3 us after its last fast entry
            const bool several = live && !coupled && (meta < 0);
            if (((leader && coupled) || several) && !(meta & 0x200)) {
                const int key = (blk << bsz) | (int)msk;
                const int pos = atomicAdd(&cg_cur[key], 1);  // count pass: a counter; fill pass: the position
                if (FILL) cg_entry[pos] = (int32_t)sig;
            }
            if (live) {
                const int pos = atomicAdd(&s_cnt[a], 1);
                // one 16-byte record per entry {signal, slot | flags, coefficient bits, 0}: one scattered store here, one
                // coalesced load in the sweep (three separate 4-byte stores made this pass twice as long)
                if (FILL) erec[pos] = make_int4((int)sig, meta, __builtin_bit_cast(int, cv[u]), __builtin_bit_cast(int, pr.pc));
            }
        }
    }
    if (!FILL) {
        __syncthreads();
        for (int a = lane; a < K; a += 64) counts[(int64_t)chunk * K + a] = s_cnt[a];
    }
}

// Exclusive scan over the chunks, in place, of the chunk-major table counts[T][K], atom totals out.  Three launches so that
// the 2 x 4 K T bytes of the table (64 MB at config 2) move through 256 workgroups instead of K / 64 = 16 (round 4: one
// launch of 16 workgroups, 100 us of the 0.43-ms index build):
//   csr_scan_partial_kernel  grid (K / 64, 16): workgroup (g, z) = 64 atoms x chunk range z; its 16 sub-segments' sums
//   csr_scan_bases_kernel    the 256 sub-segment sums of every atom -> exclusive prefix in place, atom totals out
//   csr_scan_apply_kernel    same grid as the first: the sub-segment rewritten as running offsets from its base
// (1024 threads = 16 sub-segments x 64 atoms: every access is a coalesced 256-byte run.)
constexpr int CSR_ZSEG = 16;   // chunk ranges (gridDim.y)
constexpr int CSR_SSEG = 16;   // sub-segments per range (threadIdx / 64)
__device__ __forceinline__ void csr_scan_range(int T, int z, int seg, int& c0, int& c1) {
    const int per = (T + CSR_ZSEG * CSR_SSEG - 1) / (CSR_ZSEG * CSR_SSEG);
    c0 = (z * CSR_SSEG + seg) * per;
    c1 = (c0 + per < T) ? c0 + per : T;
    if (c0 > T) c0 = T;
}
__global__ __launch_bounds__(1024) void csr_scan_partial_kernel(const int32_t* __restrict__ counts, int T, int K,
                                                                int32_t* __restrict__ part) {
    const int a = blockIdx.x * 64 + (threadIdx.x & 63), seg = threadIdx.x >> 6, z = blockIdx.y;
    int c0, c1;
    csr_scan_range(T, z, seg, c0, c1);
    int sum = 0;
    if (a < K) {
#pragma unroll 8
        for (int c = c0; c < c1; ++c) sum += counts[(int64_t)c * K + a];
        part[(int64_t)(z * CSR_SSEG + seg) * K + a] = sum;
    }
}
// one WAVE per atom: lane l owns the sub-segments 4 l .. 4 l + 3 (round 5b; one thread per atom walked its 256 sums one
// after the other on K / 256 = 4 workgroups: 36 us of dependent load / store pairs at configs[1])
__global__ __launch_bounds__(256) void csr_scan_bases_kernel(int32_t* __restrict__ part, int K, int32_t* __restrict__ totals) {
    static_assert(CSR_ZSEG * CSR_SSEG == 256, "four sub-segment sums per lane");
    const int a = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (a >= K) return;
    int v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = part[(int64_t)(4 * lane + i) * K + a];
    const int sum = v[0] + v[1] + v[2] + v[3];
    int inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(inc, off, 64);
        inc += (lane >= off) ? o : 0;
    }
    int run = inc - sum;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        part[(int64_t)(4 * lane + i) * K + a] = run;
        run += v[i];
    }
    if (lane == 63) totals[a] = inc;
}
__global__ __launch_bounds__(1024) void csr_scan_apply_kernel(int32_t* __restrict__ counts, int T, int K,
                                                              const int32_t* __restrict__ part) {
    const int a = blockIdx.x * 64 + (threadIdx.x & 63), seg = threadIdx.x >> 6, z = blockIdx.y;
    int c0, c1;
    csr_scan_range(T, z, seg, c0, c1);
    if (a >= K) return;
    int run = part[(int64_t)(z * CSR_SSEG + seg) * K + a];
    for (int c = c0; c < c1; c += 8) {   // eight loads in flight, then the dependent running sum
        int v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = (c + u < c1) ? counts[(int64_t)(c + u) * K + a] : 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (c + u < c1) counts[(int64_t)(c + u) * K + a] = run;
            run += v[u];
        }
    }
}

// row_ptr = exclusive scan of the K totals (single workgroup; K atoms, or the (block, mask) keys of the coupled-leader index:
// 32 768 at configs[1]).  Every thread owns a contiguous run of `per` elements.  Round 5b: the run is loaded ONCE with 16-byte
// loads and kept in registers (per <= 32, 16-byte aligned: both callers at configs[1]), the thread sums are scanned per wave
// with shuffles and across the 16 waves through LDS (two barriers instead of the twenty of a 1024-wide Hillis-Steele scan) --
// the old form read every element twice with 4-byte loads a cache line apart per lane: 50 us for the 32 768 keys.
__global__ __launch_bounds__(1024) void csr_scan_totals_kernel(const int32_t* __restrict__ totals, int K,
                                                               int32_t* __restrict__ row_ptr) {
    __shared__ int s_wave[16];
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int per = (K + 1023) / 1024;
    const int b = t * per, e = (b + per < K) ? b + per : K;
    const bool fast = (per <= 32) && ((per & 3) == 0) && ((reinterpret_cast<uintptr_t>(totals) & 15) == 0) &&
                      ((reinterpret_cast<uintptr_t>(row_ptr) & 15) == 0);  // uniform
    int4 v[8];
    int sum = 0;
    if (fast) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int at = b + 4 * i;
            v[i] = make_int4(0, 0, 0, 0);
            if (4 * i < per && at + 3 < K) {
                v[i] = *reinterpret_cast<const int4*>(totals + at);
            } else if (4 * i < per && at < K) {  // the last, ragged quad of the array
                v[i].x = totals[at];
                v[i].y = (at + 1 < K) ? totals[at + 1] : 0;
                v[i].z = (at + 2 < K) ? totals[at + 2] : 0;
            }
            sum += v[i].x + v[i].y + v[i].z + v[i].w;
        }
    } else {
        for (int i = b; i < e; ++i) sum += totals[i];
    }
    // inclusive scan of the thread sums: shuffles inside a wave, the 16 wave totals through LDS
    int inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(inc, off, 64);
        inc += (lane >= off) ? o : 0;
    }
    if (lane == 63) s_wave[wid] = inc;
    __syncthreads();
    int wbase = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const int x = s_wave[w];
        wbase += (w < wid) ? x : 0;
        total += x;
    }
    int run = wbase + inc - sum;  // exclusive prefix of this thread's run
    if (fast) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int at = b + 4 * i;
            if (4 * i < per && at < K) {
                int4 o;
                o.x = run;
                o.y = o.x + v[i].x;
                o.z = o.y + v[i].y;
                o.w = o.z + v[i].z;
                run = o.w + v[i].w;
                if (at + 3 < K) {
                    *reinterpret_cast<int4*>(row_ptr + at) = o;
                } else {
                    row_ptr[at] = o.x;
                    if (at + 1 < K) row_ptr[at + 1] = o.y;
                    if (at + 2 < K) row_ptr[at + 2] = o.z;
                }
            }
        }
    } else {
        for (int i = b; i < e; ++i) {
            const int x = totals[i];
            row_ptr[i] = run;
            run += x;
        }
    }
    if (t == 1023) row_ptr[K] = total;
}

// One wave walks a chunk of signals in order; the number of chunks T sets the parallelism of the build (the per-signal
// work is a serial chain per wave), bounded by the K x T counter table (<= 64 MB).
static void csr_plan(int64_t N, int K, int& T, int64_t& S) {
    int64_t t = (N + 63) / 64;
    int64_t cap = (16ll << 20) / (K > 0 ? K : 1);
    if (cap > 8192) cap = 8192;
    if (cap < 256) cap = 256;
    if (t < 1) t = 1;
    if (t > cap) t = cap;
    T = (int)t;
    S = (N + T - 1) / T;
}

size_t csr_workspace_bytes(int K, int k, int64_t N) {
    int T;
    int64_t S;
    csr_plan(N, K, T, S);
    return ((size_t)K * (size_t)T + (size_t)K + (size_t)K * CSR_ZSEG * CSR_SSEG) * sizeof(int32_t);   // counts, totals, scan partials
}

// logb == 0: entry = int32 signal*k + slot (per-atom kernels, online DL).  logb = 2 / 3 (block sweep, k <= 64): `entry`
// is an array of 16-byte records {signal, slot | flags, coefficient bits, 0} (see csr_count_or_fill_kernel), blocks of
// 2^logb atoms, plus the coupled-leader index cg_ptr [nb * 2^B + 1] / cg_entry [<= N*k/2] keyed by (block << B) | mask.
size_t csr_block_workspace_bytes(int K, int k, int64_t N, int B) {
    const size_t nkey = (size_t)((K + B - 1) / B) << B;
    return csr_workspace_bytes(K, k, N) + (nkey + 1) * sizeof(int32_t);
}

int csr_by_atom(const int32_t* idx, const float* coef, const int32_t* nnz, int K, int k, int64_t N, int32_t* row_ptr,
                int32_t* entry, void* ws, size_t ws_bytes, hipStream_t stream, int logb, int32_t* cg_ptr,
                int32_t* cg_entry) {
    const bool emeta = logb != 0;  // block-sweep mode
    if (emeta && (k > 64 || !cg_ptr || !cg_entry || (logb != 2 && logb != 3))) {
        set_error("csr_by_atom: block-sweep index needs k <= 64, B in {4, 8} and the coupled-leader buffers (k = %d)", k);
        return LYS_ENOSUP;
    }
    int T;
    int64_t S;
    csr_plan(N, K, T, S);
    const int B = emeta ? (1 << logb) : 1;
    const size_t nkey = emeta ? ((size_t)((K + B - 1) / B) << B) : 0;
    const size_t need = csr_workspace_bytes(K, k, N) + (emeta ? (nkey + 1) * sizeof(int32_t) : 0);
    if (ws_bytes < need) {
        set_error("csr_by_atom: workspace %zu < %zu", ws_bytes, need);
        return LYS_EWORKSPACE;
    }
    if ((int64_t)N * k > 0x7fffffffLL) {
        set_error("csr_by_atom: N*k = %lld exceeds int32 entries", (long long)N * k);
        return LYS_ENOSUP;
    }
    if ((size_t)K * sizeof(int) > 64 * 1024) {
        set_error("csr_by_atom: K = %d too large for the LDS counters", K);
        return LYS_ENOSUP;
    }
    int32_t* counts = static_cast<int32_t*>(ws);
    int32_t* totals = counts + (size_t)K * T;
    int32_t* part = totals + K;                                  // [CSR_ZSEG * CSR_SSEG][K] sub-segment sums of the chunk scan
    int32_t* cg_cur = emeta ? part + (size_t)K * CSR_ZSEG * CSR_SSEG : nullptr;
    const size_t lds = (size_t)K * sizeof(int);
    if (emeta) LYS_CHECK_HIP(hipMemsetAsync(cg_cur, 0, (nkey + 1) * sizeof(int32_t), stream));
    const bool quad = emeta && k <= 16;  // block-sweep index with four signals per wave instruction
    if (quad)
        hipLaunchKernelGGL(bksvd_index_kernel<false>, dim3(T), dim3(64), lds, stream, idx, coef, K, k, N, T, S, counts,
                           row_ptr, reinterpret_cast<int4*>(entry), logb, cg_cur, cg_entry);
    else
        hipLaunchKernelGGL(csr_count_or_fill_kernel, dim3(T), dim3(64), lds, stream, idx, coef, nnz, K, k, N, T, S,
                           counts, row_ptr, entry, 0, (int32_t*)nullptr, (float*)nullptr, emeta ? logb : 0, cg_cur,
                           (int32_t*)nullptr);
    LYS_LAUNCH_CHECK();
    hipLaunchKernelGGL(csr_scan_partial_kernel, dim3((K + 63) / 64, CSR_ZSEG), dim3(1024), 0, stream, counts, T, K, part);
    LYS_LAUNCH_CHECK();
    hipLaunchKernelGGL(csr_scan_bases_kernel, dim3((K + 3) / 4), dim3(256), 0, stream, part, K, totals);
    LYS_LAUNCH_CHECK();
    hipLaunchKernelGGL(csr_scan_apply_kernel, dim3((K + 63) / 64, CSR_ZSEG), dim3(1024), 0, stream, counts, T, K, part);
    LYS_LAUNCH_CHECK();
    hipLaunchKernelGGL(csr_scan_totals_kernel, dim3(1), dim3(1024), 0, stream, totals, K, row_ptr);
    LYS_LAUNCH_CHECK();
    if (emeta) {
        hipLaunchKernelGGL(csr_scan_totals_kernel, dim3(1), dim3(1024), 0, stream, cg_cur, (int)nkey, cg_ptr);
        LYS_LAUNCH_CHECK();
        LYS_CHECK_HIP(hipMemcpyAsync(cg_cur, cg_ptr, (nkey + 1) * sizeof(int32_t), hipMemcpyDeviceToDevice, stream));
    }
    if (quad)
        hipLaunchKernelGGL(bksvd_index_kernel<true>, dim3(T), dim3(64), lds, stream, idx, coef, K, k, N, T, S, counts,
                           row_ptr, reinterpret_cast<int4*>(entry), logb, cg_cur, cg_entry);
    else
        hipLaunchKernelGGL(csr_count_or_fill_kernel, dim3(T), dim3(64), lds, stream, idx, coef, nnz, K, k, N, T, S,
                           counts, row_ptr, entry, emeta ? 2 : 1, (int32_t*)nullptr, (float*)nullptr, emeta ? logb : 0,
                           cg_cur, cg_entry);
    LYS_LAUNCH_CHECK();
    return LYS_OK;
}

// ---------------------------------------------------------------------------------------------
// approximate K-SVD, one atom (ksvd.py:111-123).  A "team" = one 16-lane DPP row; lane q of a team owns
// features 64*b + 4*q .. +3 for b < FB (n <= 64*FB).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double row16_sum_d(double x) {
    for (int off = 1; off < 16; off <<= 1) x += __shfl_xor(x, off, 16);
    return x;
}

// Fixed grid (graph-capturable): 256 workgroups x 16 teams = 4096 signal teams per atom; a workgroup without entries
// exits at once.  At config 2 (about 10k signals per atom) every team handles 2-3 signals, so all residual-row
// loads of an atom are in flight together (the kernels are latency-, not bandwidth-limited: 2.5 MB per atom).
constexpr int KSVD_BLOCKS = 256;

template <int FB>
__global__ __launch_bounds__(256) void ksvd_accumulate_kernel(int atom, const float* __restrict__ R, int64_t ldr, int n,
                                                              int k, const int32_t* __restrict__ row_ptr,
                                                              const int32_t* __restrict__ entry,
                                                              const float* __restrict__ coef,
                                                              double* __restrict__ sbuf) {
    __shared__ float s_acc[16][FB * 64 + 1];
    const int beg = row_ptr[atom], end = row_ptr[atom + 1];
    const int team = threadIdx.x >> 4, q = threadIdx.x & 15;
    const int gteam = blockIdx.x * 16 + team, nteams = gridDim.x * 16;
    if (beg + blockIdx.x * 16 >= end) return;  // block has no entry (uniform per block)
    float4 acc[FB];
    float sq = 0.f;
#pragma unroll
    for (int b = 0; b < FB; ++b) acc[b] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int e = beg + gteam; e < end; e += nteams) {
        const int ss = entry[e];
        const int64_t sig = ss / k;
        const float x = coef[ss];
        sq = fmaf(x, x, sq);
#pragma unroll
        for (int b = 0; b < FB; ++b) {
            const int f = 64 * b + 4 * q;
            if (f < n) {  // ldr is padded to a multiple of 4 and padded columns hold zeros
                const float4 r = *reinterpret_cast<const float4*>(R + sig * ldr + f);
                acc[b].x = fmaf(r.x, x, acc[b].x);
                acc[b].y = fmaf(r.y, x, acc[b].y);
                acc[b].z = fmaf(r.z, x, acc[b].z);
                acc[b].w = fmaf(r.w, x, acc[b].w);
            }
        }
    }
#pragma unroll
    for (int b = 0; b < FB; ++b) {
        s_acc[team][64 * b + 4 * q + 0] = acc[b].x;
        s_acc[team][64 * b + 4 * q + 1] = acc[b].y;
        s_acc[team][64 * b + 4 * q + 2] = acc[b].z;
        s_acc[team][64 * b + 4 * q + 3] = acc[b].w;
    }
    if (q == 0) s_acc[team][FB * 64] = sq;
    __syncthreads();
    double* dst = sbuf + (int64_t)atom * (n + 1);
    for (int f = threadIdx.x; f <= n; f += 256) {
        const int src = (f == n) ? FB * 64 : f;
        double tot = 0.0;
#pragma unroll
        for (int t = 0; t < 16; ++t) tot += (double)s_acc[t][src];
        atomicAdd(dst + f, tot);
    }
}

template <int FB>
__global__ __launch_bounds__(256) void ksvd_apply_kernel(int atom, float* __restrict__ R, int64_t ldr, int n, int k,
                                                         const int32_t* __restrict__ row_ptr,
                                                         const int32_t* __restrict__ entry, float* __restrict__ coef,
                                                         const double* __restrict__ sbuf,
                                                         const float* __restrict__ D, int ldd,
                                                         float* __restrict__ Dnext) {
    const int beg = row_ptr[atom], end = row_ptr[atom + 1];
    const int team = threadIdx.x >> 4, q = threadIdx.x & 15;
    const int gteam = blockIdx.x * 16 + team, nteams = gridDim.x * 16;
    // block 0 always runs: it publishes d_new even when THIS shard holds no non-zero of the atom (multi-GPU:
    // the statistics in sbuf are already all-reduced, every rank must end up with the same dictionary)
    if (blockIdx.x != 0 && beg + blockIdx.x * 16 >= end) return;
    // every team rebuilds d_new = normalize(sum R_i x_i + d_old * sum x_i^2) in fp64 (n values, trivial)
    const double* s = sbuf + (int64_t)atom * (n + 1);
    const double sumsq = s[n];
    float4 dold[FB], dnew[FB];
    double v[FB][4];
    double nrm2 = 0.0;
#pragma unroll
    for (int b = 0; b < FB; ++b) {
        const int f = 64 * b + 4 * q;
        dold[b] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (f < n) dold[b] = *reinterpret_cast<const float4*>(D + (int64_t)atom * ldd + f);
        const float od[4] = {dold[b].x, dold[b].y, dold[b].z, dold[b].w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            v[b][c] = (f + c < n) ? s[f + c] + (double)od[c] * sumsq : 0.0;
            nrm2 += v[b][c] * v[b][c];
        }
    }
    nrm2 = row16_sum_d(nrm2);
    const double scale = 1.0 / (sqrt(nrm2) + 2.220446049250313e-16);  // normalize(): x / (||x|| + eps)
    float dd = 0.f;
#pragma unroll
    for (int b = 0; b < FB; ++b) {
        dnew[b] = make_float4((float)(v[b][0] * scale), (float)(v[b][1] * scale), (float)(v[b][2] * scale),
                              (float)(v[b][3] * scale));
        dd = fmaf(dold[b].x, dnew[b].x, dd);
        dd = fmaf(dold[b].y, dnew[b].y, dd);
        dd = fmaf(dold[b].z, dnew[b].z, dd);
        dd = fmaf(dold[b].w, dnew[b].w, dd);
    }
    dd = row16_sum(dd);  // d_old . d_new
    if (blockIdx.x == 0 && team == 0) {
#pragma unroll
        for (int b = 0; b < FB; ++b) {
            const int f = 64 * b + 4 * q;
            if (f < n) *reinterpret_cast<float4*>(Dnext + (int64_t)atom * ldd + f) = dnew[b];
        }
    }
    for (int e = beg + gteam; e < end; e += nteams) {
        const int ss = entry[e];
        const int64_t sig = ss / k;
        const float xo = coef[ss];
        float4 r[FB];
        float dot = 0.f;
#pragma unroll
        for (int b = 0; b < FB; ++b) {
            const int f = 64 * b + 4 * q;
            r[b] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f < n) r[b] = *reinterpret_cast<const float4*>(R + sig * ldr + f);
            dot = fmaf(r[b].x, dnew[b].x, dot);
            dot = fmaf(r[b].y, dnew[b].y, dot);
            dot = fmaf(r[b].z, dnew[b].z, dot);
            dot = fmaf(r[b].w, dnew[b].w, dot);
        }
        dot = row16_sum(dot);
        const float xn = fmaf(xo, dd, dot);  // (R_i + d_old x_old)' d_new
#pragma unroll
        for (int b = 0; b < FB; ++b) {
            const int f = 64 * b + 4 * q;
            if (f < n) {
                float4 o;
                o.x = fmaf(-dnew[b].x, xn, fmaf(dold[b].x, xo, r[b].x));
                o.y = fmaf(-dnew[b].y, xn, fmaf(dold[b].y, xo, r[b].y));
                o.z = fmaf(-dnew[b].z, xn, fmaf(dold[b].z, xo, r[b].z));
                o.w = fmaf(-dnew[b].w, xn, fmaf(dold[b].w, xo, r[b].w));
                *reinterpret_cast<float4*>(R + sig * ldr + f) = o;
            }
        }
        if (q == 0) coef[ss] = xn;
    }
}

// ---------------------------------------------------------------------------------------------
// Fused single-GPU step: [phase 2 of atom a-1] + [phase 1 of atom a] in ONE launch, so a sweep is K+1 dependent
// launches instead of 2K.  A signal that uses both atoms is owned by the team that meets it in omega_a (it applies
// the pending update of atom a-1 to its registers first, then accumulates for atom a); the team that meets it in
// omega_{a-1} skips it.  Membership is decided from the signal's own k-entry support row, no merged list needed.
// ---------------------------------------------------------------------------------------------
// TEAMS 16-lane teams per workgroup: every team walks its signals one after the other (three dependent memory round
// trips per signal), so the kernel time is (signals per team) x latency -- 64 teams per workgroup instead of 16 keep
// the number of fp64 atomics (one set per workgroup) and cut the chain from ~5 signals to 1-2.
template <int FB, int TEAMS>
__global__ __launch_bounds__(16 * TEAMS) void ksvd_fused_kernel(int atom, int K, int4 hb, int2 hn, float* __restrict__ R,
                                                         int64_t ldr, int n, int k,
                                                         const int32_t* __restrict__ row_ptr,
                                                         const int32_t* __restrict__ entry,
                                                         const int32_t* __restrict__ idx, float* __restrict__ coef,
                                                         double* __restrict__ sbuf, const float* __restrict__ D,
                                                         int ldd, float* __restrict__ Dnext) {
    __shared__ float s_acc[TEAMS][FB * 64 + 1];
    const int prev = atom - 1;
    const bool have_prev = prev >= 0, have_cur = atom < K;
    // segment bounds: passed by value when the host knows row_ptr (hb.x >= 0), which removes one dependent load from
    // the head of every launch; read from row_ptr otherwise (multi-GPU stepping)
    const bool byval = hb.x >= 0;
    const int pbeg = !have_prev ? 0 : byval ? hb.x : row_ptr[prev], pend = !have_prev ? 0 : byval ? hb.y : row_ptr[prev + 1];
    const int cbeg = !have_cur ? 0 : byval ? hb.z : row_ptr[atom], cend = !have_cur ? 0 : byval ? hb.w : row_ptr[atom + 1];
    const int team = threadIdx.x >> 4, q = threadIdx.x & 15;
    // ---- extra workgroups: warm L2 for the NEXT launch (atom+1): its entry list, coefficients, support rows and
    // residual rows are touched now, so that the next kernel's dependent loads hit L2 instead of HBM
    if ((int)blockIdx.x >= KSVD_BLOCKS) {
        const int nxt = atom + 1;
        if (nxt >= K) return;
        const int nb = byval ? hn.x : row_ptr[nxt], ne = byval ? hn.y : row_ptr[nxt + 1];
        const int pteam = ((int)blockIdx.x - KSVD_BLOCKS) * TEAMS + team, pteams = ((int)gridDim.x - KSVD_BLOCKS) * TEAMS;
        for (int e = nb + pteam; e < ne; e += pteams) {
            const int ss = entry[e];
            const int64_t sig = ss / k;
            float sink = coef[ss];
            if (q < k) sink += (float)idx[sig * k + q];
#pragma unroll
            for (int b = 0; b < FB; ++b) {
                const int f = 64 * b + 4 * q;
                if (f < n) sink += R[sig * ldr + f];
            }
            asm volatile("" ::"v"(sink));  // keep the loads
        }
        return;
    }
    // the two passes own disjoint signals, so they run side by side: the lower half of the grid does pass A
    // (pending updates of atom a-1), the upper half pass B (accumulation for atom a) -- one latency chain, not two
    const int half = KSVD_BLOCKS / 2;
    const bool both = have_prev && have_cur;
    const bool do_a = have_prev && (!both || (int)blockIdx.x < half);
    const bool do_b = have_cur && (!both || (int)blockIdx.x >= half);
    const int blk = both ? ((int)blockIdx.x % half) : (int)blockIdx.x;
    const int nblk = both ? half : KSVD_BLOCKS;
    const int gteam = blk * TEAMS + team, nteams = nblk * TEAMS;
    const bool blk_prev = do_a && (pbeg + blk * TEAMS < pend), blk_cur = do_b && (cbeg + blk * TEAMS < cend);
    if (!blk_prev && !blk_cur && !(blockIdx.x == 0 && have_prev)) return;  // uniform per block

    // ---- d_new of the previous atom (fp64, per team; block 0 / team 0 publishes it).  Pass-A teams need it at once;
    // a pass-B team needs it only for a signal that also uses the previous atom (about 1 % of them), so there it is
    // computed on first use and the statistics load + fp64 normalisation leave the head of that pass' latency chain.
    float4 dold[FB], dnew[FB];
    float dd = 0.f;
    bool dnew_ready = false;
    auto compute_dnew = [&]() {
        dnew_ready = true;
        const double* s = sbuf + (int64_t)prev * (n + 1);
        const double sumsq = s[n];
        double v[FB][4];
        double nrm2 = 0.0;
#pragma unroll
        for (int b = 0; b < FB; ++b) {
            const int f = 64 * b + 4 * q;
            dold[b] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f < n) dold[b] = *reinterpret_cast<const float4*>(D + (int64_t)prev * ldd + f);
            const float od[4] = {dold[b].x, dold[b].y, dold[b].z, dold[b].w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                v[b][c] = (f + c < n) ? s[f + c] + (double)od[c] * sumsq : 0.0;
                nrm2 += v[b][c] * v[b][c];
            }
        }
        nrm2 = row16_sum_d(nrm2);
        const double scale = 1.0 / (sqrt(nrm2) + 2.220446049250313e-16);
#pragma unroll
        for (int b = 0; b < FB; ++b) {
            dnew[b] = make_float4((float)(v[b][0] * scale), (float)(v[b][1] * scale), (float)(v[b][2] * scale),
                                  (float)(v[b][3] * scale));
            dd = fmaf(dold[b].x, dnew[b].x, dd);
            dd = fmaf(dold[b].y, dnew[b].y, dd);
            dd = fmaf(dold[b].z, dnew[b].z, dd);
            dd = fmaf(dold[b].w, dnew[b].w, dd);
        }
        dd = row16_sum(dd);
        if (blockIdx.x == 0 && team == 0) {
#pragma unroll
            for (int b = 0; b < FB; ++b) {
                const int f = 64 * b + 4 * q;
                if (f < n) *reinterpret_cast<float4*>(Dnext + (int64_t)prev * ldd + f) = dnew[b];
            }
        }
    };
    // (deferring it behind the first residual-row load in pass A as well was measured slower: 10.0 vs 9.2 ms/sweep)
    if (have_prev && (do_a || blockIdx.x == 0)) compute_dnew();
    // apply the pending update of `prev` to the residual row held in r[] (coefficient slot ss_prev)
    auto apply_prev = [&](float4 (&r)[FB], int ss_prev) {
        const float xo = coef[ss_prev];
        float dot = 0.f;
#pragma unroll
        for (int b = 0; b < FB; ++b) {
            dot = fmaf(r[b].x, dnew[b].x, dot);
            dot = fmaf(r[b].y, dnew[b].y, dot);
            dot = fmaf(r[b].z, dnew[b].z, dot);
            dot = fmaf(r[b].w, dnew[b].w, dot);
        }
        dot = row16_sum(dot);
        const float xn = fmaf(xo, dd, dot);
#pragma unroll
        for (int b = 0; b < FB; ++b) {
            r[b].x = fmaf(-dnew[b].x, xn, fmaf(dold[b].x, xo, r[b].x));
            r[b].y = fmaf(-dnew[b].y, xn, fmaf(dold[b].y, xo, r[b].y));
            r[b].z = fmaf(-dnew[b].z, xn, fmaf(dold[b].z, xo, r[b].z));
            r[b].w = fmaf(-dnew[b].w, xn, fmaf(dold[b].w, xo, r[b].w));
        }
        if (q == 0) coef[ss_prev] = xn;
    };
    auto load_row = [&](float4 (&r)[FB], int64_t sig) {
#pragma unroll
        for (int b = 0; b < FB; ++b) {
            const int f = 64 * b + 4 * q;
            r[b] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f < n) r[b] = *reinterpret_cast<const float4*>(R + sig * ldr + f);
        }
    };
    auto store_row = [&](const float4 (&r)[FB], int64_t sig) {
#pragma unroll
        for (int b = 0; b < FB; ++b) {
            const int f = 64 * b + 4 * q;
            if (f < n) *reinterpret_cast<float4*>(R + sig * ldr + f) = r[b];
        }
    };
    // slot of atom `a` (with a non-zero coefficient) in the support row of `sig`, or -1.  Lane q scans slots q, q+16, ..
    auto find_slot = [&](int64_t sig, int a) -> int {
        int found = -1;
        for (int j = q; j < k; j += 16) {
            if (idx[sig * k + j] == a && coef[sig * k + j] != 0.f) found = j;
        }
        // max over the 16 lanes of the team (row-local DPP)
        found = max(found, dpp_i<0xB1>(found));
        found = max(found, dpp_i<0x4E>(found));
        found = max(found, dpp_i<0x124>(found));
        found = max(found, dpp_i<0x128>(found));
        return found;
    };

    // ---- pass A: signals of omega_prev that do NOT use `atom` (those are owned by pass B)
    if (do_a) {
        for (int e = pbeg + gteam; e < pend; e += nteams) {
            const int ss = entry[e];
            const int64_t sig = ss / k;
            float4 r[FB];
            load_row(r, sig);  // issued before the ownership test: the row and the support row travel together
            if (have_cur && find_slot(sig, atom) >= 0) continue;  // uniform per team
            apply_prev(r, ss);
            store_row(r, sig);
        }
    }
    // ---- pass B: signals of omega_atom: pending update first (if the signal also used `prev`), then accumulate
    if (!do_b) return;
    float4 acc[FB];
    float sq = 0.f;
#pragma unroll
    for (int b = 0; b < FB; ++b) acc[b] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int e = cbeg + gteam; e < cend; e += nteams) {
        const int ss = entry[e];
        const int64_t sig = ss / k;
        const float x = coef[ss];
        float4 r[FB];
        load_row(r, sig);
        if (have_prev) {
            const int sp = find_slot(sig, prev);
            if (sp >= 0) {
                if (!dnew_ready) compute_dnew();
                apply_prev(r, (int)(sig * k + sp));
                store_row(r, sig);
            }
        }
        sq = fmaf(x, x, sq);
#pragma unroll
        for (int b = 0; b < FB; ++b) {
            acc[b].x = fmaf(r[b].x, x, acc[b].x);
            acc[b].y = fmaf(r[b].y, x, acc[b].y);
            acc[b].z = fmaf(r[b].z, x, acc[b].z);
            acc[b].w = fmaf(r[b].w, x, acc[b].w);
        }
    }
    if (!blk_cur) return;  // uniform per block: no team of this block had an entry of `atom`
#pragma unroll
    for (int b = 0; b < FB; ++b) {
        s_acc[team][64 * b + 4 * q + 0] = acc[b].x;
        s_acc[team][64 * b + 4 * q + 1] = acc[b].y;
        s_acc[team][64 * b + 4 * q + 2] = acc[b].z;
        s_acc[team][64 * b + 4 * q + 3] = acc[b].w;
    }
    if (q == 0) s_acc[team][FB * 64] = sq;
    __syncthreads();
    double* dst = sbuf + (int64_t)atom * (n + 1);
    for (int f = threadIdx.x; f <= n; f += 16 * TEAMS) {
        const int src = (f == n) ? FB * 64 : f;
        double tot = 0.0;
#pragma unroll 16
        for (int t = 0; t < TEAMS; ++t) tot += (double)s_acc[t][src];
        atomicAdd(dst + f, tot);
    }
}

int ksvd_fused_step(int atom, int K, float* R, int64_t ldr, int n, int k, const int32_t* row_ptr, const int32_t* entry,
                    const int32_t* idx, float* coef, double* sbuf, const float* D, float* Dnext, hipStream_t stream,
                    const int32_t* row_ptr_host) {
    int4 hb = make_int4(-1, -1, -1, -1);
    int2 hn = make_int2(0, 0);
    if (row_ptr_host) {
        hb.x = (atom >= 1) ? row_ptr_host[atom - 1] : 0;
        hb.y = (atom >= 1) ? row_ptr_host[atom] : 0;
        hb.z = (atom < K) ? row_ptr_host[atom] : 0;
        hb.w = (atom < K) ? row_ptr_host[atom + 1] : 0;
        if (atom + 1 < K) hn = make_int2(row_ptr_host[atom + 1], row_ptr_host[atom + 2]);
    }
    const int ldd = padded_features(n);
    const int fb = (n <= 64) ? 1 : (n <= 128) ? 2 : (n <= 256) ? 4 : 0;
    constexpr int pf = 256;  // workgroups that warm L2 for the next atom (measured: 13.1 -> 12.0 ms per sweep at config 2)
    switch (fb) {
        case 1: hipLaunchKernelGGL((ksvd_fused_kernel<1, 64>), dim3(KSVD_BLOCKS + pf), dim3(1024), 0, stream, atom, K, hb, hn, R, ldr, n, k, row_ptr, entry, idx, coef, sbuf, D, ldd, Dnext); break;
        case 2: hipLaunchKernelGGL((ksvd_fused_kernel<2, 64>), dim3(KSVD_BLOCKS + pf), dim3(1024), 0, stream, atom, K, hb, hn, R, ldr, n, k, row_ptr, entry, idx, coef, sbuf, D, ldd, Dnext); break;
        case 4: hipLaunchKernelGGL((ksvd_fused_kernel<4, 32>), dim3(KSVD_BLOCKS + pf), dim3(512), 0, stream, atom, K, hb, hn, R, ldr, n, k, row_ptr, entry, idx, coef, sbuf, D, ldd, Dnext); break;
        default: set_error("ksvd: n = %d > 256 not supported", n); return LYS_ENOSUP;
    }
    LYS_LAUNCH_CHECK();
    return LYS_OK;
}

__device__ __forceinline__ double block_sum_d(double x, double* pr, double* red);

// ---- n > 256 (colour patches, stacked features): the same two phases per atom with the features spread over threads
// instead of a 16-lane team per signal.  Accumulate: workgroup (feature slab of 1024, signal chunk) keeps four features
// per thread in registers while it walks its signals (every thread reads the same row: coalesced), fp64 atomics at the
// end.  Apply: `ksvd_wide_finalize_kernel` publishes d_new = normalize(s + d_old sum x^2) once, then one workgroup per
// signal does the two passes of ksvd.py:121-123 over the row.
constexpr int WIDE_SLAB = 1024;    // features per workgroup of the accumulate kernel
constexpr int WIDE_CHUNKS = 64;    // signal chunks (grid.y)

__global__ __launch_bounds__(256) void ksvd_accumulate_wide_kernel(int atom, const float* __restrict__ R, int64_t ldr, int n,
                                                                   int k, const int32_t* __restrict__ row_ptr,
                                                                   const int32_t* __restrict__ entry,
                                                                   const float* __restrict__ coef,
                                                                   double* __restrict__ sbuf) {
    __shared__ int64_t s_off[256];
    __shared__ float s_x[256];
    const int beg = row_ptr[atom], m = row_ptr[atom + 1] - beg;
    const int per = (m + (int)gridDim.y - 1) / (int)gridDim.y;
    const int e0 = blockIdx.y * per, e1 = min(m, e0 + per);
    if (e0 >= e1) return;
    const int tid = threadIdx.x, f0 = blockIdx.x * WIDE_SLAB + tid;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    double sq = 0.0;
    for (int c0 = e0; c0 < e1; c0 += 256) {
        __syncthreads();
        if (c0 + tid < e1) {
            const int ss = entry[beg + c0 + tid];
            s_off[tid] = (int64_t)(ss / k) * ldr;
            s_x[tid] = coef[ss];
        }
        __syncthreads();
        const int cnt = min(256, e1 - c0);
        for (int i = 0; i < cnt; ++i) {
            const float x = s_x[i];
            const float* row = R + s_off[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int f = f0 + 256 * j;
                acc[j] = fmaf((f < n) ? row[f] : 0.f, x, acc[j]);
            }
            if (blockIdx.x == 0 && tid == 0) sq = fma((double)x, (double)x, sq);
        }
    }
    double* dst = sbuf + (int64_t)atom * (n + 1);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int f = f0 + 256 * j;
        if (f < n) atomicAdd(dst + f, (double)acc[j]);
    }
    if (blockIdx.x == 0 && tid == 0) atomicAdd(dst + n, sq);
}

// d_new = normalize(s + d_old sum x^2) -> Dnext[atom] (one workgroup)
__global__ __launch_bounds__(256) void ksvd_wide_finalize_kernel(int atom, int n, const double* __restrict__ sbuf,
                                                                 const float* __restrict__ D, int ldd,
                                                                 float* __restrict__ Dnext) {
    __shared__ double pr[256], red[16];
    const double* s = sbuf + (int64_t)atom * (n + 1);
    const double sumsq = s[n];
    double nrm2 = 0.0;
    for (int f = threadIdx.x; f < n; f += 256) {
        const double v = s[f] + (double)D[(int64_t)atom * ldd + f] * sumsq;
        nrm2 = fma(v, v, nrm2);
    }
    const double scale = 1.0 / (sqrt(block_sum_d(nrm2, pr, red)) + 2.220446049250313e-16);
    for (int f = threadIdx.x; f < ldd; f += 256)
        Dnext[(int64_t)atom * ldd + f] = (f < n) ? (float)((s[f] + (double)D[(int64_t)atom * ldd + f] * sumsq) * scale) : 0.f;
}

__global__ __launch_bounds__(256) void ksvd_apply_wide_kernel(int atom, float* __restrict__ R, int64_t ldr, int n, int k,
                                                              const int32_t* __restrict__ row_ptr,
                                                              const int32_t* __restrict__ entry, float* __restrict__ coef,
                                                              const float* __restrict__ D, int ldd,
                                                              const float* __restrict__ Dnext) {
    __shared__ double pr[256], red[16];
    const int beg = row_ptr[atom], end = row_ptr[atom + 1];
    const float* dold = D + (int64_t)atom * ldd;
    const float* dnew = Dnext + (int64_t)atom * ldd;
    for (int e = beg + blockIdx.x; e < end; e += gridDim.x) {
        const int ss = entry[e];
        float* Ri = R + (int64_t)(ss / k) * ldr;
        const float xo = coef[ss];
        double dot = 0.0;
        for (int f = threadIdx.x; f < n; f += 256) dot += (double)fmaf(dold[f], xo, Ri[f]) * (double)dnew[f];
        const float xn = (float)block_sum_d(dot, pr, red);  // (R_i + d_old x_old)' d_new
        for (int f = threadIdx.x; f < n; f += 256) Ri[f] = fmaf(-dnew[f], xn, fmaf(dold[f], xo, Ri[f]));
        if (threadIdx.x == 0) coef[ss] = xn;
    }
}

static int fb_of(int n) { return (n <= 64) ? 1 : (n <= 128) ? 2 : (n <= 256) ? 4 : 0; }

int ksvd_atom_accumulate(int atom, const float* R, int64_t ldr, int n, int k, const int32_t* row_ptr,
                         const int32_t* entry, const float* coef, double* sbuf, hipStream_t stream) {
    switch (fb_of(n)) {
        case 1: hipLaunchKernelGGL(ksvd_accumulate_kernel<1>, dim3(KSVD_BLOCKS), dim3(256), 0, stream, atom, R, ldr, n, k, row_ptr, entry, coef, sbuf); break;
        case 2: hipLaunchKernelGGL(ksvd_accumulate_kernel<2>, dim3(KSVD_BLOCKS), dim3(256), 0, stream, atom, R, ldr, n, k, row_ptr, entry, coef, sbuf); break;
        case 4: hipLaunchKernelGGL(ksvd_accumulate_kernel<4>, dim3(KSVD_BLOCKS), dim3(256), 0, stream, atom, R, ldr, n, k, row_ptr, entry, coef, sbuf); break;
        default:
            hipLaunchKernelGGL(ksvd_accumulate_wide_kernel, dim3((unsigned)((n + WIDE_SLAB - 1) / WIDE_SLAB), WIDE_CHUNKS), dim3(256), 0,
                               stream, atom, R, ldr, n, k, row_ptr, entry, coef, sbuf);
            break;
    }
    LYS_LAUNCH_CHECK();
    return LYS_OK;
}

int ksvd_atom_apply(int atom, float* R, int64_t ldr, int n, int k, const int32_t* row_ptr, const int32_t* entry,
                    float* coef, const double* sbuf, const float* D, float* Dnext, hipStream_t stream) {
    const int ldd = padded_features(n);
    switch (fb_of(n)) {
        case 1: hipLaunchKernelGGL(ksvd_apply_kernel<1>, dim3(KSVD_BLOCKS), dim3(256), 0, stream, atom, R, ldr, n, k, row_ptr, entry, coef, sbuf, D, ldd, Dnext); break;
        case 2: hipLaunchKernelGGL(ksvd_apply_kernel<2>, dim3(KSVD_BLOCKS), dim3(256), 0, stream, atom, R, ldr, n, k, row_ptr, entry, coef, sbuf, D, ldd, Dnext); break;
        case 4: hipLaunchKernelGGL(ksvd_apply_kernel<4>, dim3(KSVD_BLOCKS), dim3(256), 0, stream, atom, R, ldr, n, k, row_ptr, entry, coef, sbuf, D, ldd, Dnext); break;
        default:
            hipLaunchKernelGGL(ksvd_wide_finalize_kernel, dim3(1), dim3(256), 0, stream, atom, n, sbuf, D, ldd, Dnext);
            hipLaunchKernelGGL(ksvd_apply_wide_kernel, dim3(1024), dim3(256), 0, stream, atom, R, ldr, n, k, row_ptr, entry, coef, D,
                               ldd, Dnext);
            break;
    }
    LYS_LAUNCH_CHECK();
    return LYS_OK;
}

__global__ void ksvd_commit_kernel(int ldd, int K, const int32_t* __restrict__ row_ptr, const float* __restrict__ Dnext,
                                   float* __restrict__ D) {
    const int a = blockIdx.x;
    if (a >= K || row_ptr[a] >= row_ptr[a + 1]) return;  // unused atom keeps its old column (ksvd.py:112-115)
    for (int f = threadIdx.x; f < ldd; f += blockDim.x) D[(int64_t)a * ldd + f] = Dnext[(int64_t)a * ldd + f];
}

int ksvd_commit(int n, int K, const int32_t* row_ptr, const float* Dnext, float* D, hipStream_t stream) {
    hipLaunchKernelGGL(ksvd_commit_kernel, dim3(K), dim3(64), 0, stream, padded_features(n), K, row_ptr, Dnext, D);
    LYS_LAUNCH_CHECK();
    return LYS_OK;
}

// ---------------------------------------------------------------------------------------------
// Exact rank-1 K-SVD atom update (lyssa/dict_learning/ksvd.py:19-43).  The reference takes the leading singular
// triplet of Rk = R[:, omega] + d_old x_omega with sklearn's randomized_svd(n_components=1, n_iter=10).  Here, per atom:
//   1. ksvd_gram_kernel   C = Rk Rk' (n x n, fp64 atomics of fp32 64x64 register tiles; one pass over the atom's rows)
//   2. ksvd_eig_kernel    leading eigenvector u of C: Lanczos with full re-orthogonalisation (<= 24 steps, one
//                         workgroup, basis in LDS, C read from L2) + Rayleigh-Ritz on the small projected matrix
//   3. ksvd_exact_apply_kernel   x_i = rk_i . u (= sigma v_i), R_i = rk_i - u x_i
// Sign convention: u . d_old >= 0 (the reference's sign is arbitrary, flip_sign=False).
// ---------------------------------------------------------------------------------------------
constexpr int GRAM_TS = 32;   // signals staged per LDS tile
constexpr int GRAM_SPB = 256; // signals per workgroup (bounds the number of fp64 atomics per atom)

__global__ __launch_bounds__(256) void ksvd_gram_kernel(int atom, const float* __restrict__ R, int64_t ldr, int n, int k,
                                                        const int32_t* __restrict__ row_ptr,
                                                        const int32_t* __restrict__ entry,
                                                        const float* __restrict__ coef, const float* __restrict__ D,
                                                        int ldd, double* __restrict__ C) {
    __shared__ __attribute__((aligned(16))) float As[GRAM_TS][68];
    __shared__ __attribute__((aligned(16))) float Bs[GRAM_TS][68];
    const int beg = row_ptr[atom], end = row_ptr[atom + 1];
    const int s0 = beg + blockIdx.x * GRAM_SPB;
    if (s0 >= end) return;
    const int s1 = min(end, s0 + GRAM_SPB);
    // blockIdx.y enumerates the upper-triangular 64x64 blocks (bi <= bj) of C
    int bi = 0, bj = blockIdx.y;
    const int nb = (n + 63) >> 6;
    while (bj >= nb - bi) {
        bj -= nb - bi;
        ++bi;
    }
    bj += bi;
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    const int lsig = tid >> 3, lf = (tid & 7) * 8;  // loader: signal lsig of the tile, features lf..lf+7 of the block
    float acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] = 0.f;
    for (int t0 = s0; t0 < s1; t0 += GRAM_TS) {
        const int e = t0 + lsig;
        float va[8], vb[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) va[c] = vb[c] = 0.f;
        if (e < s1) {
            const int ss = entry[e];
            const int64_t sig = ss / k;
            const float x = coef[ss];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int fa = bi * 64 + lf + c, fb = bj * 64 + lf + c;
                if (fa < n) va[c] = fmaf(D[(int64_t)atom * ldd + fa], x, R[sig * ldr + fa]);
                if (fb < n) vb[c] = fmaf(D[(int64_t)atom * ldd + fb], x, R[sig * ldr + fb]);
            }
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            As[lsig][lf + c] = va[c];
            Bs[lsig][lf + c] = vb[c];
        }
        __syncthreads();
#pragma unroll 8
        for (int sgl = 0; sgl < GRAM_TS; ++sgl) {
            const float4 a4 = *reinterpret_cast<const float4*>(&As[sgl][4 * ty]);
            const float4 b4 = *reinterpret_cast<const float4*>(&Bs[sgl][4 * tx]);
            const float av[4] = {a4.x, a4.y, a4.z, a4.w}, bv[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[a][c] = fmaf(av[a], bv[c], acc[a][c]);
        }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int r = bi * 64 + 4 * ty + a, cc = bj * 64 + 4 * tx + c;
            if (r < n && cc < n) {
                atomicAdd(C + (int64_t)r * n + cc, (double)acc[a][c]);
                if (bi != bj) atomicAdd(C + (int64_t)cc * n + r, (double)acc[a][c]);
            }
        }
}

// phase timestamps (100 MHz wall clock) of the last exact-update launches (lys_debug_timestamps slots 16..31):
// [0..7] eigen-solver (start, C loaded, Lanczos done, end, steps used), [8..15] workgroup 0 of the Gram kernel
__device__ unsigned long long g_exact_stamp[16];
int exact_debug_stamps(unsigned long long* out16) {
    LYS_CHECK_HIP(hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_exact_stamp), 16 * sizeof(unsigned long long)));
    return LYS_OK;
}

// n <= 64: C = Rk Rk' on the matrix cores, without atomics.  Workgroup b of P owns a contiguous slice of the atom's
// signals, stages 64 restricted-residual rows at a time in LDS (next round's rows prefetched into registers) and
// accumulates the 64 x 64 product with v_mfma_f32_32x32x2_f32 (one 32 x 32 quadrant per wave); the fp32 partial goes to
// part[b][64][64] with plain stores and the eigen-solver sums the P partials in fp64 while it loads C.
constexpr int G64_MAX_PARTS = 256;
constexpr int G64_ROWS = 768;  // most signals one workgroup takes (descriptor staging area)
constexpr int G64_SKIP = 0x7fc00001;

// eflag (may be null): per-entry flags of the pipelined sweep below; an entry with bit 0 set (its signal also uses the previous
// used atom) contributes nothing here.
__device__ __forceinline__ void gram64_part(int atom, int beg, int m, int bx, int P, const float* __restrict__ R, int64_t ldr,
                                            int n, int k, const int32_t* __restrict__ entry,
                                            const uint8_t* __restrict__ eflag, const float* __restrict__ coef,
                                            const float* __restrict__ D, int ldd, float* __restrict__ part) {
    __shared__ float s_a[64][65];
    __shared__ int64_t s_off[G64_ROWS];
    __shared__ float s_x[G64_ROWS];
    if (m <= 0) return;
    int chunk = (m + P - 1) / P;
    chunk = ((chunk + 63) >> 6) << 6;
    const int e0 = bx * chunk, cnt = max(0, min(chunk, m - e0));
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ti = wid >> 1, tj = wid & 1;
    const unsigned long long ts0 = wall_clock64();
    using f16v = __attribute__((ext_vector_type(16))) float;
    f16v acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = tid; i < cnt; i += 256) {
        const int ss = entry[beg + e0 + i];
        const int fl = eflag ? eflag[beg + e0 + i] : 0;
        s_off[i] = (int64_t)(ss / k) * ldr;
        // this NaN pattern marks a row that is left out (x_i is finite: the coder's output); compared as bits below
        s_x[i] = (fl & 1) ? __builtin_bit_cast(float, G64_SKIP) : coef[ss];
    }
    const float d = (lane < n) ? D[(int64_t)atom * ldd + lane] : 0.f;
    __syncthreads();
    // wave w stages rows w, w + 4, .. of each 64-row round; lane = feature.  The prefetch holds RAW residual values
    // (unconditional loads from clamped addresses: a select on the loaded value would make the compiler wait for every
    // load in turn); x_i d is added and out-of-range rows / features are zeroed when the round is written to LDS.
    float cur[16], nxt[16];
    const int lf = min(lane, n - 1);
    auto fetch = [&](int r0, float* v) {
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = R[s_off[min(r0 + wid + 4 * q, cnt - 1)] + lf];
    };
    unsigned long long ts1 = wall_clock64();
    if (cnt > 0) fetch(0, cur);
    for (int r0 = 0; r0 < cnt; r0 += 64) {
        if (r0 + 64 < cnt) fetch(r0 + 64, nxt);
        // LDS-only barriers: __syncthreads() would also wait for the prefetch loads just issued (vmcnt(0))
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the previous round's MFMAs have read s_a
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int i = r0 + wid + 4 * q;
            const float xi = s_x[min(i, cnt - 1)];
            s_a[wid + 4 * q][lane] =
                (i < cnt && lane < n && __builtin_bit_cast(int, xi) != G64_SKIP) ? fmaf(d, xi, cur[q]) : 0.f;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll 8
        for (int i = 0; i < 64; i += 2) {
            const int kk = i + (lane >> 5);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(s_a[kk][32 * ti + (lane & 31)], s_a[kk][32 * tj + (lane & 31)], acc,
                                                       0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) cur[q] = nxt[q];
    }
    const unsigned long long ts2 = wall_clock64();
    float* out = part + (int64_t)bx * 4096;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r >> 2) * 8 + (lane >> 5) * 4 + (r & 3), col = lane & 31;
        out[(32 * ti + row) * 64 + 32 * tj + col] = acc[r];
    }
    if (bx == 0 && tid == 0) {
        g_exact_stamp[8] = ts0;
        g_exact_stamp[9] = ts1;
        g_exact_stamp[10] = ts2;
        g_exact_stamp[11] = wall_clock64();
        g_exact_stamp[12] = (unsigned long long)cnt;
    }
}

__global__ __launch_bounds__(256) void ksvd_gram64_kernel(int atom, const float* __restrict__ R, int64_t ldr, int n, int k,
                                                          const int32_t* __restrict__ row_ptr,
                                                          const int32_t* __restrict__ entry,
                                                          const float* __restrict__ coef, const float* __restrict__ D,
                                                          int ldd, float* __restrict__ part) {
    const int beg = row_ptr[atom];
    gram64_part(atom, beg, row_ptr[atom + 1] - beg, (int)blockIdx.x, (int)gridDim.x, R, ldr, n, k, entry, nullptr, coef, D, ldd,
                part);
}

__device__ __forceinline__ int prev_used_atom(const int32_t* __restrict__ row_ptr, int a) {
    int p = a - 1;
    while (p >= 0 && row_ptr[p] >= row_ptr[p + 1]) --p;  // uniform; unused atoms are rare
    return p;  // -1: none
}
__device__ __forceinline__ int next_used_atom(const int32_t* __restrict__ row_ptr, int a, int K) {
    int q = a + 1;
    while (q < K && row_ptr[q] >= row_ptr[q + 1]) ++q;
    return q;  // K: none
}

constexpr int EIG_M = 24;  // Lanczos steps (Krylov dimension)

// sum over the 256 threads of the workgroup through LDS (two levels of 16; no cross-lane fp64 shuffles)
__device__ __forceinline__ double block_sum_d(double x, double* pr /* [256] */, double* red /* [16] */) {
    __syncthreads();
    pr[threadIdx.x] = x;
    __syncthreads();
    if (threadIdx.x < 16) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += pr[threadIdx.x * 16 + q];
        red[threadIdx.x] = t;
    }
    __syncthreads();
    double tot = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) tot += red[q];
    return tot;
}

// Leading eigenpair of the m x m projected matrix S = Q'CQ (S_ij = H[min(i,j)][max(i,j)]) on ONE wave: S^(2^10) by
// repeated squaring (rescaled by the largest diagonal entry each time), c = its column with the largest diagonal,
// normalised.  out[0] = theta = c'Sc, out[1] = ||S c - theta c||.  LDS traffic of one wave is in order, so the lanes
// see each other's writes without a workgroup barrier.
__device__ void ritz_wave(int m, const double (*H)[EIG_M], double* Ta, double* Tb, double* cvec, double* ys, double* out,
                          int lane) {
    // every reduction below is done redundantly by every lane from LDS broadcast reads: m <= 24 values, against six
    // two-instruction ds_bpermute rounds per fp64 cross-lane reduction
    for (int i = lane; i < m * m; i += 64) {
        const int r = i / m, c = i - r * m;
        Ta[r * EIG_M + c] = H[min(r, c)][max(r, c)];
    }
    double* A = Ta;
    double* B = Tb;
    for (int it = 0; it < 10; ++it) {
        __builtin_amdgcn_wave_barrier();
        double dg = 0.0;
        for (int l = 0; l < m; ++l) dg = fmax(dg, fabs(A[l * EIG_M + l]));
        if (!(dg > 0.0)) break;
        const double inv2 = (1.0 / dg) * (1.0 / dg);
        for (int i = lane; i < m * m; i += 64) {
            const int r = i / m, c = i - r * m;
            double t = 0.0;
            for (int l = 0; l < m; ++l) t = fma(A[r * EIG_M + l], A[l * EIG_M + c], t);
            B[r * EIG_M + c] = t * inv2;
        }
        double* tmp = A;
        A = B;
        B = tmp;
    }
    __builtin_amdgcn_wave_barrier();
    int jb = 0;
    for (int j = 1; j < m; ++j)
        if (A[j * EIG_M + j] > A[jb * EIG_M + jb]) jb = j;
    double nrm2 = 0.0;
    for (int l = 0; l < m; ++l) nrm2 = fma(A[l * EIG_M + jb], A[l * EIG_M + jb], nrm2);
    double c = (lane < m) ? A[lane * EIG_M + jb] : 0.0;
    c = (nrm2 > 0.0) ? c / sqrt(nrm2) : (lane == 0 ? 1.0 : 0.0);
    if (lane < EIG_M) cvec[lane] = (lane < m) ? c : 0.0;
    __builtin_amdgcn_wave_barrier();
    double y = 0.0;
    if (lane < m)
        for (int j = 0; j < m; ++j) y = fma(H[min(lane, j)][max(lane, j)], cvec[j], y);
    if (lane < EIG_M) ys[lane] = y;
    __builtin_amdgcn_wave_barrier();
    double theta = 0.0, r2 = 0.0;
    for (int l = 0; l < m; ++l) theta = fma(cvec[l], ys[l], theta);
    for (int l = 0; l < m; ++l) {
        const double r = ys[l] - theta * cvec[l];
        r2 = fma(r, r, r2);
    }
    if (lane == 0) {
        out[0] = theta;
        out[1] = sqrt(r2);
    }
}

// One workgroup (256 threads).  Dynamic LDS: Q[(EIG_M + 1) * n] doubles.
// Tall mode (entry != nullptr; n > 256 features, |omega_a| <= 256): the matrix is the |omega| x |omega| Gram matrix of
// the restricted residual's ROWS, the start vector the atom's current coefficients, the result v goes to `vout`.
__global__ __launch_bounds__(256) void ksvd_eig_kernel(int atom, int n, const int32_t* __restrict__ row_ptr,
                                                       const double* __restrict__ Cg, const float* __restrict__ D,
                                                       int ldd, float* __restrict__ Dnext, int c_in_lds,
                                                       const int32_t* __restrict__ entry = nullptr,
                                                       const float* __restrict__ coef = nullptr,
                                                       float* __restrict__ vout = nullptr, int parts = 0) {
    extern __shared__ __attribute__((aligned(16))) double Q[];  // [EIG_M + 1][n], then (c_in_lds) a copy of C [n][n]
    __shared__ double H[EIG_M][EIG_M], T[EIG_M][EIG_M], T2[EIG_M][EIG_M];
    __shared__ double hh[EIG_M + 1], red[16], wv[256], pr[256], cvec[EIG_M], ritz[2], rys[EIG_M];
    __shared__ int m_used;
    if (row_ptr[atom] >= row_ptr[atom + 1]) return;
    const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
    const bool tall = entry != nullptr;
    const unsigned long long ts0 = wall_clock64();
    if (tall) {
        n = row_ptr[atom + 1] - row_ptr[atom];
        c_in_lds = n <= 64;
    }
    // n <= 256: thread tid owns component tid of every n-vector
    const double d0 = (tid < n) ? (tall ? (double)coef[entry[row_ptr[atom] + tid]] : (double)D[(int64_t)atom * ldd + tid]) : 0.0;
    for (int i = tid; i < EIG_M * EIG_M; i += 256) (&H[0][0])[i] = 0.0;
    double nrm2 = block_sum_d(d0 * d0, pr, red);
    {
        double q0 = (nrm2 > 0.0) ? d0 / sqrt(nrm2) : (tid == 0 ? 1.0 : 0.0);
        if (tid < n) Q[tid] = q0;
    }
    if (tid == 0) m_used = EIG_M;
    // small n: C (32 KB at n = 64) is copied into LDS once, the 32 matrix-vector products then never leave the CU
    const double* C = Cg;
    if (c_in_lds) {
        double* Cl = Q + (int64_t)(EIG_M + 1) * n;
        if (parts > 0) {
            // n <= 64: C arrives as `parts` fp32 partial 64 x 64 sums of ksvd_gram64_kernel; summed here in fp64
            // (thread t owns entries 4t..4t+3 of each 1024-entry quarter: four 16-byte loads per partial, eight
            // partials unrolled = 32 loads in flight per lane; one pass over parts x 16 KB)
            const float4* P = reinterpret_cast<const float4*>(Cg);
            double t[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) t[q] = 0.0;
            for (int p0 = 0; p0 < parts; p0 += 8) {
                float4 v[8][4];  // all 32 loads are issued before the first add (clamped partial index, masked below)
#pragma unroll
                for (int pp = 0; pp < 8; ++pp)
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[pp][q] = P[(int64_t)min(p0 + pp, parts - 1) * 1024 + tid + 256 * q];
#pragma unroll
                for (int pp = 0; pp < 8; ++pp) {
                    const double on = (p0 + pp < parts) ? 1.0 : 0.0;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        t[4 * q + 0] = fma(on, (double)v[pp][q].x, t[4 * q + 0]);
                        t[4 * q + 1] = fma(on, (double)v[pp][q].y, t[4 * q + 1]);
                        t[4 * q + 2] = fma(on, (double)v[pp][q].z, t[4 * q + 2]);
                        t[4 * q + 3] = fma(on, (double)v[pp][q].w, t[4 * q + 3]);
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int e = 4 * tid + 1024 * (q >> 2) + (q & 3), r = e >> 6, c = e & 63;
                if (r < n && c < n) Cl[r * n + c] = t[q];
            }
        } else {
            for (int i = tid; i < n * n; i += 256) Cl[i] = Cg[i];
        }
        C = Cl;
    }
    __syncthreads();
    const unsigned long long ts1 = wall_clock64();
    double scale0 = 0.0;
    int m = 0;
    for (int j = 0; j < EIG_M; ++j) {
        // w = C q_j (C symmetric: column access is coalesced).  The 256 threads split the sum over c into 256/npad
        // parts (npad = n rounded up to 64/128/256), independent loads unrolled 8-deep: the matvec is L2-latency bound
        double w = 0.0;
        {
            const int npad = (n <= 64) ? 64 : (n <= 128) ? 128 : 256;
            const int parts = 256 / npad, part = tid / npad, r = tid & (npad - 1);
            const int c0 = (n * part) / parts, c1 = (n * (part + 1)) / parts;
            const double* qj = Q + (int64_t)j * n;
            double acc = 0.0;
            if (r < n) {
#pragma unroll 8
                for (int c = c0; c < c1; ++c) acc = fma(C[(int64_t)c * n + r], qj[c], acc);
            }
            __syncthreads();  // wv is free again (read by the previous step's Gram-Schmidt)
            wv[tid] = acc;
            __syncthreads();
            if (tid < n)
                for (int pp = 0; pp < parts; ++pp) w += wv[pp * npad + tid];
            __syncthreads();
        }
        // classical Gram-Schmidt, twice, against q_0..q_j; the first round's coefficients are column j of H.
        // Dot products: thread (i = tid/8, part = tid%8) sums one eighth of q_i . w, the 8 partials meet in LDS
        // (wave-wide fp64 shuffles through ds_bpermute made this the slowest part of the kernel).
        for (int round = 0; round < 2; ++round) {
            wv[tid] = w;
            __syncthreads();
            {
                const int i = tid >> 3, part = tid & 7;
                double d = 0.0;
                if (i <= j) {
                    const int c0 = (n * part) >> 3, c1 = (n * (part + 1)) >> 3;
                    const double* qi = Q + (int64_t)i * n;
#pragma unroll 4
                    for (int c = c0; c < c1; ++c) d = fma(qi[c], wv[c], d);
                }
                pr[tid] = d;
            }
            __syncthreads();
            if (tid <= j) {
                double d = 0.0;
#pragma unroll
                for (int q8 = 0; q8 < 8; ++q8) d += pr[tid * 8 + q8];
                hh[tid] = d;
                if (round == 0) H[tid][j] = d;
                else H[tid][j] += d;
            }
            __syncthreads();
            if (tid < n) {
#pragma unroll 4
                for (int i = 0; i <= j; ++i) w = fma(-hh[i], Q[(int64_t)i * n + tid], w);
            }
        }
        const double beta2 = block_sum_d(w * w, pr, red);
        const double beta = sqrt(beta2);
        m = j + 1;
        if (j == 0) scale0 = fabs(H[0][0]) + beta;
        const bool last = !(beta > 1e-13 * scale0) || j == EIG_M - 1 || j + 1 >= n;  // Krylov space exhausted
        // Rayleigh-Ritz on the m x m projection (one wave; every step up to m = 12, every second one after that) and
        // the residual of the leading Ritz pair, ||C u - theta u||^2 = ||S c - theta c||^2 + (beta c_m)^2: converged
        // pairs stop the recurrence -- K-SVD's restricted residuals have one dominant direction, typically 6-10 steps
        if (last || m <= 12 || (m & 1) == 0) {
            __syncthreads();  // H column j complete
            if (wid == 0) ritz_wave(m, H, &T[0][0], &T2[0][0], cvec, rys, ritz, lane);
            __syncthreads();
            const double tail = beta * cvec[m - 1];
            if (last || sqrt(ritz[1] * ritz[1] + tail * tail) <= 1e-9 * fabs(ritz[0])) break;
        }
        if (tid < n) Q[(int64_t)(j + 1) * n + tid] = w / beta;
        __syncthreads();
    }
    __syncthreads();
    const unsigned long long ts2 = wall_clock64();
    double u = 0.0;
    if (tid < n)
        for (int j = 0; j < m; ++j) u = fma(cvec[j], Q[(int64_t)j * n + tid], u);
    const double un2 = block_sum_d(u * u, pr, red);
    const double sg = block_sum_d(u * d0, pr, red);
    if (un2 > 0.0) u *= (sg < 0.0 ? -1.0 : 1.0) / sqrt(un2);
    else u = d0;
    if (tid < n) {
        if (tall) vout[tid] = (float)u;
        else Dnext[(int64_t)atom * ldd + tid] = (float)u;
    }
    if (tid == 0) {
        g_exact_stamp[0] = ts0;
        g_exact_stamp[1] = ts1;
        g_exact_stamp[2] = ts2;
        g_exact_stamp[3] = wall_clock64();
        g_exact_stamp[4] = (unsigned long long)m;
    }
}

// Cheap convergence test for the single-wave solver: with full re-orthogonalisation the projected matrix is tridiagonal
// (a_i = H[i][i], b_i = H[i][i+1]) up to rounding.  Largest eigenvalue by Newton's iteration on the characteristic
// polynomial from the Gershgorin bound (monotone from the right of the largest root), eigenvector by the BACKWARD
// three-term recurrence (stable for the decaying leading vector); rows 1..m-1 of (T - theta) s = 0 then hold by
// construction and the whole residual sits in row 0, so the test verifies itself: a Newton iteration that has not
// converged just fails the test.  Every lane computes the same scalars; O(m) per Newton step.  Returns true when the
// Ritz pair's residual sqrt(r_0^2 + (beta s_m)^2) <= tol * theta, with the normalised s in cvec.
// Hardware reciprocal / reciprocal square root refined by two Newton steps each (full double precision for normal
// arguments): a fraction of the IEEE division / sqrt sequences, which sat in the eigen-solver's serial chain a dozen times per
// Lanczos step (round 4).
__device__ __forceinline__ double fast_rcp(double a) {
    double r = __builtin_amdgcn_rcp(a);
    r = fma(fma(-a, r, 1.0), r, r);
    r = fma(fma(-a, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ double fast_rsqrt(double a) {
    double r = __builtin_amdgcn_rsq(a);
    r = fma(fma(-0.5 * a * r, r, 0.5), r, r);
    r = fma(fma(-0.5 * a * r, r, 0.5), r, r);
    return r;
}

// Newton's iteration on the characteristic polynomial of the scaled M x M tridiagonal (da: diagonal, db2: squared
// sub-diagonal, db2[i] couples i-1 and i) from x >= its largest root
template <int M>
__device__ __forceinline__ double tridiag_newton(const double (&da)[6], const double (&db2)[6], double x) {
    for (int it = 0; it < 48; ++it) {
        double p0 = 1.0, d0 = 0.0, p1 = da[0] - x, d1 = -1.0;
#pragma unroll
        for (int i = 1; i < M; ++i) {
            const double a = da[i] - x;
            const double p2 = a * p1 - db2[i] * p0, d2 = a * d1 - p1 - db2[i] * d0;
            p0 = p1;
            d0 = d1;
            p1 = p2;
            d1 = d2;
        }
        if (d1 == 0.0) break;
        const double dx = p1 * fast_rcp(d1);
        x -= dx;
        if (fabs(dx) <= 4e-16) break;
    }
    return x;
}

// theta_prev (in / out): the largest eigenvalue of the leading (m-1) x (m-1) block from the previous call, 0: not known.
// With it the iteration starts from the secular-equation bound lambda <= ((alpha + theta') + sqrt((alpha - theta')^2 +
// 4 b^2)) / 2 (alpha, b: the new row; every term of the secular sum is below its largest-pole term) instead of the
// Gershgorin bound: two or three Newton steps instead of eight (round 4: the test was 1.5 of a Lanczos step's 3.5 us).
__device__ bool ritz_tridiag(int m, const double (*H)[EIG_M], double beta, double* cvec, double tol, int lane,
                             double& theta_prev) {
    double sc = 0.0, x = 1.0;  // x: largest eigenvalue of the scaled matrix, Newton from an upper bound
    constexpr int RM = 6;      // K-SVD's restricted residuals stop at m = 3 .. 5
    if (m <= RM) {
        // small m: the tridiagonal goes to REGISTERS first (one batch of LDS broadcast reads) -- the Newton iteration of the
        // general branch re-reads H from LDS inside its inner loop (two dependent round trips per row and iteration: ~2 us
        // per call at m = 4).  (All EIG_M = 24 rows in registers with guarded steps was slower than the LDS loop.)
        double da[RM], db[RM];  // diagonal, sub-diagonal (db[i] = H[i-1][i])
#pragma unroll
        for (int i = 0; i < RM; ++i) {
            da[i] = (i < m) ? H[i][i] : 0.0;
            db[i] = (i >= 1 && i < m) ? H[i - 1][i] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < RM; ++i) {
            double row = fabs(da[i]) + fabs(db[i]);          // rows >= m are all zero
            if (i + 1 < RM) row += fabs(db[i + 1]);
            sc = fmax(sc, row);
        }
        if (!(sc > 0.0) || !(sc < 1e300)) return false;
        // power-of-two scale: sc becomes 2^e >= the Gershgorin bound, 1 / sc is exact
        const int ex = ilogb(sc) + 1;
        sc = ldexp(1.0, ex);
        const double isc = ldexp(1.0, -ex);
        if (m >= 3) {
            double tp = theta_prev;
            if (!(tp > 0.0) && m == 3) {  // leading 2 x 2 block in closed form (inflated: any tp >= the true value keeps the bound)
                const double h = da[0] - da[1], g = h * h + 4.0 * db[1] * db[1];
                tp = 0.5 * ((da[0] + da[1]) + g * fast_rsqrt(fmax(g, 1e-300)) * (1.0 + 1e-12));
            }
            if (tp > 0.0) {
                double al = da[2], b = db[2];
#pragma unroll
                for (int i = 3; i < RM; ++i) {
                    al = (i == m - 1) ? da[i] : al;
                    b = (i == m - 1) ? db[i] : b;
                }
                const double h = al - tp, g = h * h + 4.0 * b * b;
                const double up = 0.5 * ((al + tp) + g * fast_rsqrt(fmax(g, 1e-300)) * (1.0 + 1e-12));
                x = fmin(1.0, up * isc * (1.0 + 1e-12));
            }
        }
#pragma unroll
        for (int i = 0; i < RM; ++i) {
            da[i] *= isc;
            db[i] = (db[i] * isc) * (db[i] * isc);  // b^2 of the scaled matrix
        }
        switch (m) {
            case 1: x = da[0]; break;
            case 2: x = tridiag_newton<2>(da, db, x); break;
            case 3: x = tridiag_newton<3>(da, db, x); break;
            case 4: x = tridiag_newton<4>(da, db, x); break;
            case 5: x = tridiag_newton<5>(da, db, x); break;
            default: x = tridiag_newton<6>(da, db, x); break;
        }
    } else {
        for (int i = 0; i < m; ++i) {
            double row = fabs(H[i][i]);
            if (i > 0) row += fabs(H[i - 1][i]);
            if (i < m - 1) row += fabs(H[i][i + 1]);
            sc = fmax(sc, row);
        }
        if (!(sc > 0.0)) return false;
        const double isc = 1.0 / sc;
        if (theta_prev > 0.0) {
            const double al = H[m - 1][m - 1], b = H[m - 2][m - 1], h = al - theta_prev;
            x = fmin(1.0, 0.5 * ((al + theta_prev) + sqrt(h * h + 4.0 * b * b)) * isc * (1.0 + 1e-14));
        }
        for (int it = 0; it < 48; ++it) {
            double p0 = 1.0, d0 = 0.0, p1 = H[0][0] * isc - x, d1 = -1.0;
            for (int i = 1; i < m; ++i) {
                const double a = H[i][i] * isc - x, b = H[i - 1][i] * isc, b2 = b * b;
                const double p2 = a * p1 - b2 * p0, d2 = a * d1 - p1 - b2 * d0;
                p0 = p1;
                d0 = d1;
                p1 = p2;
                d1 = d2;
            }
            if (d1 == 0.0) break;
            const double dx = p1 / d1;
            x -= dx;
            if (fabs(dx) <= 4e-16) break;
        }
    }
    const double theta = x * sc;
    theta_prev = theta;
    // backward recurrence, s_{m-1} = 1
    double s1 = 1.0, s2 = 0.0, mine = (lane == m - 1) ? 1.0 : 0.0, nrm2 = 1.0;  // s1 = s_i, s2 = s_{i+1}
    for (int i = m - 1; i >= 1; --i) {
        const double b = H[i - 1][i];
        if (b == 0.0) return false;
        const double bi = (i < m - 1) ? H[i][i + 1] : 0.0;
        const double s0 = ((theta - H[i][i]) * s1 - bi * s2) * fast_rcp(b);  // (1 / b does not depend on s: off the chain)
        s2 = s1;
        s1 = s0;
        if (lane == i - 1) mine = s0;
        nrm2 = fma(s0, s0, nrm2);
    }
    if (!(nrm2 > 0.0) || !(nrm2 < 1e300)) return false;
    const double inv = fast_rsqrt(nrm2);
    const double r0 = ((H[0][0] - theta) * s1 + ((m > 1) ? H[0][1] * s2 : 0.0)) * inv;
    const double tail = beta * inv;  // beta * s_{m-1}, s_{m-1} = 1 before normalisation
    if (lane < EIG_M) cvec[lane] = (lane < m) ? mine * inv : 0.0;
    return r0 * r0 + tail * tail <= (tol * theta) * (tol * theta);
}

// C = sum of the P fp32 partial Gram matrices, in fp64, spread over 16 workgroups (one matrix element per thread): the
// eigen-solver's single workgroup used to pull all P x 16 KB through one CU (8.7 us of its 24 at P = 32).
__global__ __launch_bounds__(256) void ksvd_gram64_reduce_kernel(int atom, const int32_t* __restrict__ row_ptr,
                                                                 const float* __restrict__ part, int parts,
                                                                 double* __restrict__ C) {
    if (row_ptr[atom] >= row_ptr[atom + 1]) return;
    const int e = blockIdx.x * 256 + threadIdx.x;  // 0 .. 4095
    // 16 loads in flight per thread (clamped partial index, masked add): the loop is a chain of dependent HBM/L2 round
    // trips otherwise (4 in flight: 8 us per launch at 54 partials)
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int p0 = 0; p0 < parts; p0 += 16) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = part[(int64_t)min(p0 + u, parts - 1) * 4096 + e];
#pragma unroll
        for (int u = 0; u < 16; ++u) acc[u & 3] += (p0 + u < parts) ? (double)v[u] : 0.0;
    }
    C[e] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
}

// Wave-wide sum of a double, every lane gets the result: four DPP row steps on the two halves, then the four row sums
// through readlane.
__device__ __forceinline__ double wave_sum_d(double x) {
#define LYS_DSTEP(CTRL)                                                                            \
    do {                                                                                           \
        const int lo_ = dpp_i<CTRL>(__double2loint(x)), hi_ = dpp_i<CTRL>(__double2hiint(x));      \
        x += __hiloint2double(hi_, lo_);                                                           \
    } while (0)
    LYS_DSTEP(0xB1);
    LYS_DSTEP(0x4E);
    LYS_DSTEP(0x124);
    LYS_DSTEP(0x128);
#undef LYS_DSTEP
    const int lo = __double2loint(x), hi = __double2hiint(x);
    const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
    const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
    const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
    const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
    return (r0 + r1) + (r2 + r3);
}

// n <= 64 with the partial Gram sums of ksvd_gram64_kernel: the same Lanczos / Rayleigh-Ritz recurrence on ONE wave
// (lane = vector component).  The four-wave kernel above spends its time in ~15 workgroup barriers per step; here the
// matrix-vector product is a loop of LDS broadcast reads and the dot products / norms are DPP reductions inside the wave (no
// barrier at all).  The 256
// threads only share the load of C (sum of the fp32 partials in fp64), then waves 1..3 retire.
constexpr int E64_QS = 65;  // row stride of the Krylov basis (doubles): lanes reading different rows hit different banks

constexpr int g_eig_mmin = 2;  // first Lanczos step the Ritz test runs at
constexpr int g_eig_pre = 4;   // power steps before the Lanczos recurrence of the single-wave solver
constexpr int XK1_RED = 64;            // workgroups of K1's partial-sum role (64 matrix elements each)
constexpr int XK1_APPLY_BLOCKS = 1024;  // most apply workgroups of a K1 launch (16 entries each per pass)
constexpr int XL_SH = 4;  // workgroups (= fp32 partials) of the shared-row Gram part of the pipelined sweep below

// spart / ns: ns <= XL_SH more fp32 partials (64 x 64 each) to add to the fp64 sum (parts < 0 only)
__device__ __forceinline__ void eig64_body(int atom, int n, const int32_t* __restrict__ row_ptr,
                                           const float* __restrict__ part, int parts, const float* __restrict__ spart, int ns,
                                           const float* __restrict__ D, int ldd, float* __restrict__ Dnext) {
    __shared__ double Cl[64 * 64];
    __shared__ double Q[(EIG_M + 1) * E64_QS];
    __shared__ double H[EIG_M][EIG_M], T[EIG_M][EIG_M], T2[EIG_M][EIG_M];
    __shared__ double hh[EIG_M + 1], wv[64], cvec[EIG_M], ritz[2], rys[EIG_M];
    if (row_ptr && row_ptr[atom] >= row_ptr[atom + 1]) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const unsigned long long ts0 = wall_clock64();
    // the start vector's load is issued with the loads of C (after the barrier below it cost a round trip of its own)
    const float d0f = (lane < n) ? D[(int64_t)atom * ldd + lane] : 0.f;
    if (parts < 0) {
        // C already summed in fp64 by ksvd_gram64_reduce_kernel (64 x 64 doubles behind the partials): 32 KB
        const double2* C2 = reinterpret_cast<const double2*>(part);
        double2 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = C2[tid + 256 * q];
        if (ns > 0) {  // uniform; all loads issued from clamped partial indices, masked adds in a fixed order
            const float2* S2 = reinterpret_cast<const float2*>(spart);
            float2 w[XL_SH][8];
#pragma unroll
            for (int t = 0; t < XL_SH; ++t)
#pragma unroll
                for (int q = 0; q < 8; ++q) w[t][q] = S2[(int64_t)min(t, ns - 1) * 2048 + tid + 256 * q];
#pragma unroll
            for (int t = 0; t < XL_SH; ++t)
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    v[q].x += (t < ns) ? (double)w[t][q].x : 0.0;
                    v[q].y += (t < ns) ? (double)w[t][q].y : 0.0;
                }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            Cl[2 * (tid + 256 * q)] = v[q].x;
            Cl[2 * (tid + 256 * q) + 1] = v[q].y;
        }
    } else {
        const float4* P = reinterpret_cast<const float4*>(part);
        double t[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) t[q] = 0.0;
        for (int p0 = 0; p0 < parts; p0 += 8) {
            float4 v[8][4];  // all 32 loads are issued before the first add (clamped partial index, masked below)
#pragma unroll
            for (int pp = 0; pp < 8; ++pp)
#pragma unroll
                for (int q = 0; q < 4; ++q) v[pp][q] = P[(int64_t)min(p0 + pp, parts - 1) * 1024 + tid + 256 * q];
#pragma unroll
            for (int pp = 0; pp < 8; ++pp) {
                const double on = (p0 + pp < parts) ? 1.0 : 0.0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    t[4 * q + 0] = fma(on, (double)v[pp][q].x, t[4 * q + 0]);
                    t[4 * q + 1] = fma(on, (double)v[pp][q].y, t[4 * q + 1]);
                    t[4 * q + 2] = fma(on, (double)v[pp][q].z, t[4 * q + 2]);
                    t[4 * q + 3] = fma(on, (double)v[pp][q].w, t[4 * q + 3]);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) Cl[4 * tid + 1024 * (q >> 2) + (q & 3)] = t[q];  // 64 x 64, rows/cols >= n are zero
    }
    for (int i = tid; i < EIG_M * EIG_M; i += 256) (&H[0][0])[i] = 0.0;
    __syncthreads();
    if (tid >= 64) return;
    const unsigned long long ts1 = wall_clock64();
    const double d0 = (double)d0f;
    auto sumsq64 = [&](double x) { return wave_sum_d(x * x); };  // (round 2: a 64-step loop of LDS broadcast reads, ~0.45 us)
    {
        // start vector: g_eig_pre power steps on d_old first (a matrix-vector product + one reduction is ~0.6 us, a Lanczos
        // step 2-3 us)
        double sv = d0;
        double nrm2 = sumsq64(sv);
        for (int ps = 0; ps < g_eig_pre && nrm2 > 0.0; ++ps) {
            __builtin_amdgcn_wave_barrier();
            wv[lane] = sv * fast_rsqrt(nrm2);
            __builtin_amdgcn_wave_barrier();
            double w0 = 0.0, w1 = 0.0, w2 = 0.0, w3 = 0.0;
#pragma unroll 8
            for (int c = 0; c < 64; c += 4) {
                w0 = fma(Cl[c * 64 + lane], wv[c], w0);
                w1 = fma(Cl[(c + 1) * 64 + lane], wv[c + 1], w1);
                w2 = fma(Cl[(c + 2) * 64 + lane], wv[c + 2], w2);
                w3 = fma(Cl[(c + 3) * 64 + lane], wv[c + 3], w3);
            }
            const double nv = (w0 + w1) + (w2 + w3);
            const double n2 = sumsq64(nv);
            if (!(n2 > 0.0)) break;  // C = 0 (or the vector in its null space): keep the previous one
            sv = nv;
            nrm2 = n2;
        }
        Q[lane] = (nrm2 > 0.0) ? sv * fast_rsqrt(nrm2) : (lane == 0 ? 1.0 : 0.0);
    }
    double scale0 = 0.0, theta_prev = 0.0;
    int m = 0;
    for (int j = 0; j < EIG_M; ++j) {
        __builtin_amdgcn_wave_barrier();
        const double* qj = Q + j * E64_QS;
        double w = 0.0;
        {
            double w0 = 0.0, w1 = 0.0, w2 = 0.0, w3 = 0.0;  // C symmetric: column access, conflict-free
#pragma unroll 8
            for (int c = 0; c < 64; c += 4) {
                w0 = fma(Cl[c * 64 + lane], qj[c], w0);
                w1 = fma(Cl[(c + 1) * 64 + lane], qj[c + 1], w1);
                w2 = fma(Cl[(c + 2) * 64 + lane], qj[c + 2], w2);
                w3 = fma(Cl[(c + 3) * 64 + lane], qj[c + 3], w3);
            }
            w = (w0 + w1) + (w2 + w3);
        }
        // classical Gram-Schmidt, twice: lane i <= j takes the dot product q_i . w
        if (j < 6) {
            // few basis vectors (K-SVD's restricted residuals converge in ~4 steps): the lane's components of q_0 .. q_j are
            // read from LDS ONCE per step, every dot product is one DPP reduction (branch-free: the slots past j carry
            // zeros), and H is written once after both rounds (the first version re-read Q four times per vector and did a
            // read-modify-write of H in LDS per round: 1.04 us per step)
            double qr[6], hs[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                qr[i] = (i <= j) ? Q[i * E64_QS + lane] : 0.0;
                hs[i] = 0.0;
            }
#pragma unroll
            for (int round = 0; round < 2; ++round) {
                double hj[6];
                // j is uniform: only the j + 1 live dot products are reduced (round 4; six every step before -- a
                // reduction is ~25 dependent instructions)
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    hj[i] = 0.0;
                    if (i <= j) hj[i] = wave_sum_d(qr[i] * w);
                }
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    w = fma(-hj[i], qr[i], w);
                    hs[i] += hj[i];
                }
            }
            if (lane == 0) {
#pragma unroll
                for (int i = 0; i < 6; ++i)
                    if (i <= j) H[i][j] = hs[i];
            }
        } else
        for (int round = 0; round < 2; ++round) {
            __builtin_amdgcn_wave_barrier();
            wv[lane] = w;
            __builtin_amdgcn_wave_barrier();
            if (lane <= j) {
                const double* qi = Q + lane * E64_QS;
                double h0 = 0.0, h1 = 0.0, h2 = 0.0, h3 = 0.0;
#pragma unroll 4
                for (int c = 0; c < 64; c += 4) {
                    h0 = fma(qi[c], wv[c], h0);
                    h1 = fma(qi[c + 1], wv[c + 1], h1);
                    h2 = fma(qi[c + 2], wv[c + 2], h2);
                    h3 = fma(qi[c + 3], wv[c + 3], h3);
                }
                const double h = (h0 + h1) + (h2 + h3);
                hh[lane] = h;
                if (round == 0) H[lane][j] = h;
                else H[lane][j] += h;
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll 4
            for (int i = 0; i <= j; ++i) w = fma(-hh[i], Q[i * E64_QS + lane], w);
        }
        const double w2 = sumsq64(w);
        const double ibeta = (w2 > 1e-300) ? fast_rsqrt(w2) : 0.0;
        const double beta = w2 * ibeta;
        m = j + 1;
        if (j == 0) scale0 = fabs(H[0][0]) + beta;
        const bool last = !(beta > 1e-13 * scale0) || j == EIG_M - 1 || j + 1 >= n;
        __builtin_amdgcn_wave_barrier();
        if (last) {  // the robust dense Rayleigh-Ritz (repeated squaring) closes the recurrence
            ritz_wave(m, H, &T[0][0], &T2[0][0], cvec, rys, ritz, lane);
            __builtin_amdgcn_wave_barrier();
            break;
        }
        if (m >= g_eig_mmin && ritz_tridiag(m, H, beta, cvec, 1e-9, lane, theta_prev)) {
            __builtin_amdgcn_wave_barrier();
            break;
        }
        Q[(j + 1) * E64_QS + lane] = w * ibeta;
#ifdef LYS_EXACT_STEP_STAMPS
        if (j < 3 && lane == 0) g_exact_stamp[5 + j] = wall_clock64();
#endif
    }
    __builtin_amdgcn_wave_barrier();
    const unsigned long long ts2 = wall_clock64();
    double u = 0.0;
    for (int j = 0; j < m; ++j) u = fma(cvec[j], Q[j * E64_QS + lane], u);
    const double un2 = sumsq64(u);
    const double sg = wave_sum_d(u * d0);
    if (un2 > 0.0) u *= (sg < 0.0 ? -1.0 : 1.0) / sqrt(un2);
    else u = d0;
    if (lane < n) Dnext[(int64_t)atom * ldd + lane] = (float)u;
    if (lane == 0) {
        g_exact_stamp[0] = ts0;
        g_exact_stamp[1] = ts1;
        g_exact_stamp[2] = ts2;
        g_exact_stamp[3] = wall_clock64();
        g_exact_stamp[4] = (unsigned long long)m;
        g_exact_stamp[13] += (unsigned long long)m;  // running sum of the steps and number of solves (tools/exact_stamps.py)
        g_exact_stamp[14] += 1ull;
    }
}

__global__ __launch_bounds__(256) void ksvd_eig64_kernel(int atom, int n, const int32_t* __restrict__ row_ptr,
                                                         const float* __restrict__ part, int parts,
                                                         const float* __restrict__ D, int ldd, float* __restrict__ Dnext) {
    eig64_body(atom, n, row_ptr, part, parts, nullptr, 0, D, ldd, Dnext);
}

// ---------------------------------------------------------------------------------------------
// Round 4: the PIPELINED exact sweep (n <= 64, k <= 16, the codes' atom indices at hand): TWO dependent launches per atom
// instead of four, with the Gram products of the NEXT atom beside the eigen-solve of this one.
//   The Gram matrix of atom a only depends on the update of the previous used atom p through the signals that use BOTH
//   (about 1 % of either list at configs[1]).  So the sweep splits every atom's list by two per-entry flags, set once per
//   sweep (exact_flag_kernel; bit 0: the signal also uses p, bit 1: it also uses the next used atom), and runs
//     K2(p):  [eigen-solve of p -> u_p]  beside  [Gram partials of a over the rows WITHOUT bit 0]
//     K1(a):  [fp64 sum of those partials]  beside  [the shared rows (a, p): p's pending update rk = R_i + d_p x_p,
//             x_p' = rk . u_p, R_i = rk - u_p x_p' (ksvd.py:36-40) applied in place, then their Gram partials]  beside
//             [the apply of p on its rows WITHOUT bit 1]
//     K2(a):  [eigen-solve of a on sum + shared partials]  beside  [Gram partials of the next used atom] ...
//   Every residual row is written by exactly one workgroup of one launch, and read by a Gram part only in a launch after
//   the one that wrote it: the Gauss-Seidel order of ksvd.py:28-43 is kept exactly (the sums are taken in a different --
//   fixed -- order than the four-launch path, so the two agree to rounding, not bit for bit).
//   The host reads row_ptr and the shared-row counts back (two short synchronisations per sweep): unused atoms are skipped, and
//   every launch gets its atoms' list bounds as kernel arguments instead of opening with a chain of dependent scalar loads.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float row16_max(float x) {  // max over the 16 lanes of a DPP row, in every lane
    x = fmaxf(x, dpp_f<0xB1>(x));
    x = fmaxf(x, dpp_f<0x4E>(x));
    x = fmaxf(x, dpp_f<0x124>(x));
    x = fmaxf(x, dpp_f<0x128>(x));
    return x;
}

// pn[a] = previous used atom of a (-1: none), pn[K + a] = next used atom (K: none)
__global__ void exact_neighbours_kernel(int K, const int32_t* __restrict__ row_ptr, int32_t* __restrict__ pn) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= K) return;
    pn[a] = prev_used_atom(row_ptr, a);
    pn[K + a] = next_used_atom(row_ptr, a, K);
}

// One 16-lane row per entry (lane = slot of the entry's signal): eflag[e], and plink[e] = coefficient position of the previous
// used atom in the same signal (-1: not there).  Membership as the CSR index defines it (coef != 0).  Four entries per row
// and pass, their loads issued together (the pass is a chain of three dependent gathers).
__global__ __launch_bounds__(256) void exact_flag_kernel(int K, int k, const int32_t* __restrict__ row_ptr,
                                                         const int32_t* __restrict__ entry, const int32_t* __restrict__ idx,
                                                         const float* __restrict__ coef, const int32_t* __restrict__ pn,
                                                         uint8_t* __restrict__ eflag, int32_t* __restrict__ plink) {
    constexpr int U = 4;
    const int q = threadIdx.x & 15;
    const int total = row_ptr[K];
    const int nrows = gridDim.x * 16;
    for (int e0 = blockIdx.x * 16 + (threadIdx.x >> 4); e0 < total; e0 += U * nrows) {
        int ss[U], aj[U], a[U], p[U], nx[U];
        float cj[U];
#pragma unroll
        for (int u = 0; u < U; ++u) ss[u] = entry[min(e0 + u * nrows, total - 1)];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t at = (int64_t)(ss[u] / k) * k + min(q, k - 1);
            aj[u] = idx[at];
            cj[u] = coef[at];
            a[u] = idx[ss[u]];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            p[u] = pn[a[u]];
            nx[u] = pn[K + a[u]];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + u * nrows;
            const bool live = q < k && cj[u] != 0.f;
            const float ps = row16_max((live && aj[u] == p[u]) ? (float)(q + 1) : 0.f);  // p = -1 never matches a live slot
            const float nsl = row16_max((live && aj[u] == nx[u]) ? 1.f : 0.f);           // nx = K neither
            if (q == 0 && e < total) {
                eflag[e] = (uint8_t)((ps > 0.f ? 1 : 0) | (nsl > 0.f ? 2 : 0));
                plink[e] = (ps > 0.f) ? (ss[u] / k) * k + (int)ps - 1 : -1;
            }
        }
    }
}

// One workgroup per atom: stable compaction of its (a, p) shared entries, in list (= signal) order, behind row_ptr[a] of shrec;
// their number in nsh[a].  A record = {signal, position of p's coefficient, x_a, x_p}: both coefficients still hold their
// sweep-start values when K1(a) uses them (x_a changes in apply(a), x_p -- for exactly these rows -- in K1(a) itself), so the
// shared part reads ONE record per row and then the residual row: two dependent loads instead of four.
__global__ __launch_bounds__(256) void exact_compact_kernel(int k, const int32_t* __restrict__ row_ptr,
                                                            const int32_t* __restrict__ entry, const int32_t* __restrict__ plink,
                                                            const float* __restrict__ coef, int4* __restrict__ shrec,
                                                            int32_t* __restrict__ nsh) {
    __shared__ int s_w[2][4];
    const int a = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int beg = row_ptr[a], end = row_ptr[a + 1];
    int filled = 0;  // shared entries found so far (uniform)
    int par = 0;
    for (int b0 = beg; b0 < end; b0 += 256, par ^= 1) {
        const int e = b0 + tid;
        const int pl = (e < end) ? plink[e] : -1;
        const int ss = entry[min(e, end - 1)];
        const unsigned long long bal = __ballot(pl >= 0);
        if (lane == 0) s_w[par][wid] = __popcll(bal);
        __syncthreads();  // (two count rows: the next round's writes cannot overtake this round's reads)
        int off = filled;
        for (int w = 0; w < wid; ++w) off += s_w[par][w];
        if (pl >= 0)  // ~1 % of the entries
            shrec[beg + off + __popcll(bal & ((1ull << lane) - 1ull))] =
                make_int4(ss / k, pl, __builtin_bit_cast(int, coef[ss]), __builtin_bit_cast(int, coef[pl]));
        filled += s_w[par][0] + s_w[par][1] + s_w[par][2] + s_w[par][3];
    }
    if (tid == 0) nsh[a] = filled;
}

__device__ __forceinline__ int xl_shared_chunk(int ns) {  // shared rows per workgroup of K1's shared part
    // spread evenly over the XL_SH workgroups (multiples of 8 rows = 4 MFMA steps): with whole rounds of 64 the first
    // workgroup carried 64 of ~100 rows -- 1.1 us of fp32 matrix products on the critical role of the launch
    const int c = (ns + XL_SH - 1) / XL_SH;
    return max(8, ((c + 7) >> 3) << 3);
}

// apply of atom p (list [beg, end)) on its entries without flag bit 1 (FB = 1: n <= 64)
__device__ __forceinline__ void exact_apply_unshared(int p, int beg, int end, int bx, int nblk, float* __restrict__ R, int64_t ldr,
                                                     int n, int k, const int32_t* __restrict__ entry,
                                                     const uint8_t* __restrict__ eflag, float* __restrict__ coef,
                                                     const float* __restrict__ D, int ldd, const float* __restrict__ Dnext) {
    const int team = threadIdx.x >> 4, q = threadIdx.x & 15;
    const int gteam = bx * 16 + team, nteams = nblk * 16;
    if (beg + bx * 16 >= end) return;
    const int f = 4 * q;
    float4 dold = make_float4(0.f, 0.f, 0.f, 0.f), u = dold;
    if (f < n) {
        dold = *reinterpret_cast<const float4*>(D + (int64_t)p * ldd + f);
        u = *reinterpret_cast<const float4*>(Dnext + (int64_t)p * ldd + f);
    }
    for (int e = beg + gteam; e < end; e += nteams) {
        const int ss = entry[e];
        const int fl = eflag[e];
        if (fl & 2) continue;  // uniform per team: K1's shared part owns this row
        const int64_t sig = ss / k;
        const float xo = coef[ss];
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        if (f < n) r = *reinterpret_cast<const float4*>(R + sig * ldr + f);
        float4 rk;
        rk.x = fmaf(dold.x, xo, r.x);
        rk.y = fmaf(dold.y, xo, r.y);
        rk.z = fmaf(dold.z, xo, r.z);
        rk.w = fmaf(dold.w, xo, r.w);
        float dot = fmaf(rk.x, u.x, 0.f);
        dot = fmaf(rk.y, u.y, dot);
        dot = fmaf(rk.z, u.z, dot);
        dot = fmaf(rk.w, u.w, dot);
        const float xn = row16_sum(dot);
        if (f < n) {
            float4 o;
            o.x = fmaf(-u.x, xn, rk.x);
            o.y = fmaf(-u.y, xn, rk.y);
            o.z = fmaf(-u.z, xn, rk.z);
            o.w = fmaf(-u.w, xn, rk.w);
            *reinterpret_cast<float4*>(R + sig * ldr + f) = o;
        }
        if (q == 0) coef[ss] = xn;
    }
}

// the (atom, p) shared rows: p's pending update in place, then their Gram partial (gram64_part's tiling).  Every wave reads the
// records of its own rows (wave-uniform addresses) and then the rows: no staging pass, no workgroup barrier before the loads;
// the sixteen rows of a wave go through one branch-free block, so their sixteen wave reductions interleave.
__device__ __forceinline__ void exact_shared_part(int atom, int p, int beg, int sb, float* __restrict__ R, int64_t ldr, int n,
                                                  const int4* __restrict__ shrec, int ns, float* __restrict__ coef,
                                                  const float* __restrict__ D, int ldd, const float* __restrict__ Dnext,
                                                  float* __restrict__ spart) {
    __shared__ float s_a[64][65];
    const int chunk = xl_shared_chunk(ns);
    const int j0 = sb * chunk, total = min(chunk, ns - j0);
    if (total <= 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ti = wid >> 1, tj = wid & 1;
    using f16v = __attribute__((ext_vector_type(16))) float;
    f16v acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const float d = (lane < n) ? D[(int64_t)atom * ldd + lane] : 0.f;
    const float dprev = (lane < n) ? D[(int64_t)p * ldd + lane] : 0.f;
    const float uprev = (lane < n) ? Dnext[(int64_t)p * ldd + lane] : 0.f;
    const int lf = min(lane, n - 1);
    const int4* rec = shrec + beg + j0;
    for (int r0 = 0; r0 < total; r0 += 64) {
        int4 rc[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) rc[q] = rec[min(r0 + wid + 4 * q, total - 1)];
        float cur[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) cur[q] = R[(int64_t)rc[q].x * ldr + lf];
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the previous round's MFMAs have read s_a
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const bool on = (r0 + wid + 4 * q < total) && lane < n;  // (row valid: wave-uniform)
            const float xa = __builtin_bit_cast(float, rc[q].z), xp = __builtin_bit_cast(float, rc[q].w);
            const float rk = on ? fmaf(dprev, xp, cur[q]) : 0.f;
            const float xn = wave_sum_f(rk * uprev);
            const float rn = fmaf(-uprev, xn, rk);
            if (on) R[(int64_t)rc[q].x * ldr + lane] = rn;
            if (on && lane == 0) coef[rc[q].y] = xn;
            s_a[wid + 4 * q][lane] = on ? fmaf(d, xa, rn) : 0.f;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const int kmax = min(64, (total - r0 + 7) & ~7);  // the rows past `total` of this round are zero: skip their products
        for (int i0 = 0; i0 < kmax; i0 += 8) {
#pragma unroll
            for (int i = 0; i < 8; i += 2) {
                const int kk = i0 + i + (lane >> 5);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(s_a[kk][32 * ti + (lane & 31)], s_a[kk][32 * tj + (lane & 31)], acc,
                                                           0, 0, 0);
            }
        }
    }
    float* out = spart + (int64_t)sb * 4096;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r >> 2) * 8 + (lane >> 5) * 4 + (r & 3), col = lane & 31;
        out[(32 * ti + row) * 64 + 32 * tj + col] = acc[r];
    }
}

// K1(atom; p = the used atom before it, -1: none): blocks [0, XK1_RED) sum atom's `parts` Gram partials, the next XL_SH take the
// nsa (atom, p) shared rows, the rest apply p.  atom < 0 closes the sweep: only the apply of p (the last used atom).
__global__ __launch_bounds__(256) void exact_k1_kernel(int atom, int abeg, int parts, int nsa, int p, int pbeg, int pend,
                                                       float* __restrict__ R, int64_t ldr, int n, int k,
                                                       const int32_t* __restrict__ entry, const uint8_t* __restrict__ eflag,
                                                       const int4* __restrict__ shrec,
                                                       float* __restrict__ coef, const float* __restrict__ D, int ldd,
                                                       const float* __restrict__ Dnext, const float* __restrict__ part,
                                                       double* __restrict__ Csum, float* __restrict__ spart) {
    const int bx = blockIdx.x;
#ifdef LYS_EXACT_K1_STAMPS
    struct K1Stamp {  // in-kernel duration of the first workgroup of each role (tools/exact_stamps.py, K1_STAMPS=1)
        unsigned long long t0;
        int slot;
        __device__ ~K1Stamp() {
            if (slot >= 0 && threadIdx.x == 0) g_exact_stamp[slot] = wall_clock64() - t0;
        }
    } k1s;
    k1s.t0 = wall_clock64();
    k1s.slot = atom < 0 ? -1 : bx == 0 ? 5 : bx == XK1_RED ? 6 : bx == XK1_RED + XL_SH ? 7 : -1;
#endif
    if (bx < XK1_RED) {
        if (atom < 0) return;
        // 64 matrix elements per workgroup, the partials split over its four waves (each has all its <= 64 loads in flight at
        // once), the four sums added in wave order: 16 workgroups of 256 elements pulled 106 KB each through one CU (3.4 us)
        __shared__ double s_r[4][64];
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        const int e = bx * 64 + lane;
        const int per = (parts + 3) / 4, p0 = w * per, p1 = min(parts, p0 + per);
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        float v[64];
#pragma unroll
        for (int u = 0; u < 64; ++u) v[u] = part[(int64_t)min(p0 + u, parts - 1) * 4096 + e];
#pragma unroll
        for (int u = 0; u < 64; ++u) acc[u & 3] += (p0 + u < p1) ? (double)v[u] : 0.0;
        s_r[w][lane] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        __syncthreads();
        if (w == 0) Csum[e] = (s_r[0][lane] + s_r[1][lane]) + (s_r[2][lane] + s_r[3][lane]);
        return;
    }
    if (p < 0) return;
    if (bx < XK1_RED + XL_SH) {
        if (atom >= 0) exact_shared_part(atom, p, abeg, bx - XK1_RED, R, ldr, n, shrec, nsa, coef, D, ldd, Dnext, spart);
        return;
    }
    exact_apply_unshared(p, pbeg, pend, bx - XK1_RED - XL_SH, (int)gridDim.x - XK1_RED - XL_SH, R, ldr, n, k, entry, eflag, coef, D, ldd,
                         Dnext);
}

// K2(atom, -1: none; nx = the used atom after it, -1: none): block 0 solves atom's eigenproblem, the other blocks form the Gram
// partials of nx over the rows that atom does not touch.
__global__ __launch_bounds__(256) void exact_k2_kernel(int atom, int nsa, int nx, int nbeg, int nm, const float* __restrict__ R,
                                                       int64_t ldr, int n, int k, const int32_t* __restrict__ entry,
                                                       const uint8_t* __restrict__ eflag,
                                                       const float* __restrict__ coef, const float* __restrict__ D, int ldd,
                                                       float* __restrict__ Dnext, float* __restrict__ part,
                                                       const double* __restrict__ Csum, const float* __restrict__ spart) {
    if (blockIdx.x == 0) {
        if (atom < 0) return;
        const int ns = (nsa + xl_shared_chunk(nsa) - 1) / xl_shared_chunk(nsa);  // shared partials (nsa = 0 for the first used atom)
        eig64_body(atom, n, nullptr, reinterpret_cast<const float*>(Csum), -1, spart, ns, D, ldd, Dnext);
        return;
    }
    if (nx < 0) return;
    gram64_part(nx, nbeg, nm, (int)blockIdx.x - 1, (int)gridDim.x - 1, R, ldr, n, k, entry, eflag, coef, D, ldd, part);
}

// x_i = rk_i . u (= sigma v_i), R_i = rk_i - u x_i with u = D_next[atom] (ksvd.py:36-40)
template <int FB>
__global__ __launch_bounds__(256) void ksvd_exact_apply_kernel(int atom, float* __restrict__ R, int64_t ldr, int n, int k,
                                                               const int32_t* __restrict__ row_ptr,
                                                               const int32_t* __restrict__ entry,
                                                               float* __restrict__ coef, const float* __restrict__ D,
                                                               int ldd, const float* __restrict__ Dnext) {
    const int beg = row_ptr[atom], end = row_ptr[atom + 1];
    const int team = threadIdx.x >> 4, q = threadIdx.x & 15;
    const int gteam = blockIdx.x * 16 + team, nteams = gridDim.x * 16;
    if (beg + blockIdx.x * 16 >= end) return;
    float4 dold[FB], u[FB];
#pragma unroll
    for (int b = 0; b < FB; ++b) {
        const int f = 64 * b + 4 * q;
        dold[b] = u[b] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (f < n) {
            dold[b] = *reinterpret_cast<const float4*>(D + (int64_t)atom * ldd + f);
            u[b] = *reinterpret_cast<const float4*>(Dnext + (int64_t)atom * ldd + f);
        }
    }
    for (int e = beg + gteam; e < end; e += nteams) {
        const int ss = entry[e];
        const int64_t sig = ss / k;
        const float xo = coef[ss];
        float4 rk[FB];
        float dot = 0.f;
#pragma unroll
        for (int b = 0; b < FB; ++b) {
            const int f = 64 * b + 4 * q;
            float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f < n) r = *reinterpret_cast<const float4*>(R + sig * ldr + f);
            rk[b].x = fmaf(dold[b].x, xo, r.x);
            rk[b].y = fmaf(dold[b].y, xo, r.y);
            rk[b].z = fmaf(dold[b].z, xo, r.z);
            rk[b].w = fmaf(dold[b].w, xo, r.w);
            dot = fmaf(rk[b].x, u[b].x, dot);
            dot = fmaf(rk[b].y, u[b].y, dot);
            dot = fmaf(rk[b].z, u[b].z, dot);
            dot = fmaf(rk[b].w, u[b].w, dot);
        }
        const float xn = row16_sum(dot);
#pragma unroll
        for (int b = 0; b < FB; ++b) {
            const int f = 64 * b + 4 * q;
            if (f < n) {
                float4 o;
                o.x = fmaf(-u[b].x, xn, rk[b].x);
                o.y = fmaf(-u[b].y, xn, rk[b].y);
                o.z = fmaf(-u[b].z, xn, rk[b].z);
                o.w = fmaf(-u[b].w, xn, rk[b].w);
                *reinterpret_cast<float4*>(R + sig * ldr + f) = o;
            }
        }
        if (q == 0) coef[ss] = xn;
    }
}

// ---- n > 256 ("tall": many features, few signals per atom -- the LC-KSVD shape, lc_ksvd.py:165 stacks [X; sqrt(a) Q;
// sqrt(b) H] to n_features + K + n_classes rows).  The rank-1 SVD of Rk (n x m, m = |omega| <= 256) goes through the
// m x m Gram matrix of its COLUMNS: M = Rk' Rk, leading eigenvector v (same Lanczos kernel), u = +-Rk v / ||Rk v|| with
// u . d_old >= 0, x_i = rk_i . u.  Workspace: [s2, u_raw . d_old | pad(6) | M (256^2) | v (256 floats) | u_raw (n floats)].
constexpr int TALL_MAX = 256;
constexpr int TALL_FCH = 256;  // features per workgroup of the Gram kernel

__global__ __launch_bounds__(256) void ksvd_gram_t_kernel(int atom, const float* __restrict__ R, int64_t ldr, int n, int k,
                                                          const int32_t* __restrict__ row_ptr,
                                                          const int32_t* __restrict__ entry,
                                                          const float* __restrict__ coef, const float* __restrict__ D,
                                                          int ldd, double* __restrict__ M) {
    __shared__ __attribute__((aligned(16))) float As[32][68];  // [feature of the tile][signal of the block]
    __shared__ __attribute__((aligned(16))) float Bs[32][68];
    const int beg = row_ptr[atom], m = row_ptr[atom + 1] - beg;
    if (m <= 0) return;
    const int mb = (m + 63) >> 6;
    int bi = 0, bj = blockIdx.y;  // upper-triangular 64x64 blocks of M, enumerated for the largest m (4 x 4)
    while (bj >= 4 - bi) {
        bj -= 4 - bi;
        ++bi;
    }
    bj += bi;
    if (bj >= mb) return;
    const int f0 = blockIdx.x * TALL_FCH, f1 = min(n, f0 + TALL_FCH);
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    const int lsig = tid >> 2, lf = (tid & 3) * 8;  // loader: signal lsig of the block, 8 consecutive features
    const int ia = bi * 64 + lsig, ib = bj * 64 + lsig;
    int64_t sa = -1, sb = -1;
    float xa = 0.f, xb = 0.f;
    if (ia < m) {
        const int ss = entry[beg + ia];
        sa = ss / k;
        xa = coef[ss];
    }
    if (ib < m) {
        const int ss = entry[beg + ib];
        sb = ss / k;
        xb = coef[ss];
    }
    float acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] = 0.f;
    for (int ft = f0; ft < f1; ft += 32) {
        float va[8], vb[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int f = ft + lf + c;
            const float d = (f < f1) ? D[(int64_t)atom * ldd + f] : 0.f;
            va[c] = (f < f1 && sa >= 0) ? fmaf(d, xa, R[sa * ldr + f]) : 0.f;
            vb[c] = (f < f1 && sb >= 0) ? fmaf(d, xb, R[sb * ldr + f]) : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            As[lf + c][lsig] = va[c];
            Bs[lf + c][lsig] = vb[c];
        }
        __syncthreads();
#pragma unroll 8
        for (int fl = 0; fl < 32; ++fl) {
            const float4 a4 = *reinterpret_cast<const float4*>(&As[fl][4 * ty]);
            const float4 b4 = *reinterpret_cast<const float4*>(&Bs[fl][4 * tx]);
            const float av[4] = {a4.x, a4.y, a4.z, a4.w}, bv[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[a][c] = fmaf(av[a], bv[c], acc[a][c]);
        }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int r = bi * 64 + 4 * ty + a, cc = bj * 64 + 4 * tx + c;
            if (r < m && cc < m) {
                atomicAdd(M + (int64_t)r * m + cc, (double)acc[a][c]);
                if (bi != bj) atomicAdd(M + (int64_t)cc * m + r, (double)acc[a][c]);
            }
        }
}

// u_raw = Rk v (one thread per feature), s2 += ||u_raw||^2
__global__ __launch_bounds__(256) void ksvd_tall_u_kernel(int atom, const float* __restrict__ R, int64_t ldr, int n, int k,
                                                          const int32_t* __restrict__ row_ptr,
                                                          const int32_t* __restrict__ entry,
                                                          const float* __restrict__ coef, const float* __restrict__ D,
                                                          int ldd, const float* __restrict__ v, float* __restrict__ uraw,
                                                          double* __restrict__ s2) {
    __shared__ int64_t sg[TALL_MAX];
    __shared__ float xs[TALL_MAX], vs[TALL_MAX];
    __shared__ double pr[256], red[16];
    const int beg = row_ptr[atom], m = row_ptr[atom + 1] - beg;
    if (m <= 0) return;
    const int tid = threadIdx.x;
    if (tid < m) {
        const int ss = entry[beg + tid];
        sg[tid] = (int64_t)(ss / k) * ldr;
        xs[tid] = coef[ss];
        vs[tid] = v[tid];
    }
    __syncthreads();
    const int f = blockIdx.x * 256 + tid;
    float u0 = 0.f, u1 = 0.f;
    if (f < n) {
        const float d = D[(int64_t)atom * ldd + f];
        int i = 0;
        for (; i + 1 < m; i += 2) {
            u0 = fmaf(vs[i], fmaf(d, xs[i], R[sg[i] + f]), u0);
            u1 = fmaf(vs[i + 1], fmaf(d, xs[i + 1], R[sg[i + 1] + f]), u1);
        }
        if (i < m) u0 = fmaf(vs[i], fmaf(d, xs[i], R[sg[i] + f]), u0);
        u0 += u1;
        uraw[f] = u0;
        u1 = u0 * d;
    }
    const double tot = block_sum_d((double)u0 * (double)u0, pr, red);
    const double sd = block_sum_d((f < n) ? (double)u1 : 0.0, pr, red);
    if (tid == 0) {
        atomicAdd(s2, tot);
        atomicAdd(s2 + 1, sd);  // u_raw . d_old: fixes the sign like the short path (u . d_old >= 0)
    }
}

// one workgroup per signal of omega: x_i = rk_i . u, R_i = rk_i - u x_i with u = u_raw / sqrt(s2); workgroup 0 also
// writes the new atom
__global__ __launch_bounds__(256) void ksvd_tall_apply_kernel(int atom, float* __restrict__ R, int64_t ldr, int n, int k,
                                                              const int32_t* __restrict__ row_ptr,
                                                              const int32_t* __restrict__ entry, float* __restrict__ coef,
                                                              const float* __restrict__ D, int ldd,
                                                              const float* __restrict__ uraw,
                                                              const double* __restrict__ s2, float* __restrict__ Dnext) {
    __shared__ double pr[256], red[16];
    const int beg = row_ptr[atom], m = row_ptr[atom + 1] - beg;
    const int i = blockIdx.x;
    if (i >= m) return;
    const int tid = threadIdx.x;
    const int ss = entry[beg + i];
    float* Ri = R + (int64_t)(ss / k) * ldr;
    const float xo = coef[ss];
    const double s2v = s2[0];
    const float inv = (s2v > 0.0) ? (float)((s2[1] < 0.0 ? -1.0 : 1.0) / sqrt(s2v)) : 0.f;
    const float* dold = D + (int64_t)atom * ldd;
    double dot = 0.0;
    for (int f = tid; f < n; f += 256) dot += (double)fmaf(dold[f], xo, Ri[f]) * (double)uraw[f];
    const float xn = (float)(block_sum_d(dot, pr, red) * (double)inv);
    const float g = xn * inv;
    for (int f = tid; f < n; f += 256) Ri[f] = fmaf(-uraw[f], g, fmaf(dold[f], xo, Ri[f]));
    if (tid == 0) coef[ss] = xn;
    if (i == 0)
        for (int f = tid; f < n; f += 256) Dnext[(int64_t)atom * ldd + f] = (s2v > 0.0) ? uraw[f] * inv : dold[f];
}

// ---- n > 256 AND an atom used by more than TALL_MAX signals: neither Gram matrix (n x n, |omega| x |omega|) is small.
// Matrix-free power iteration on C = Rk Rk' from u = d_old: per iteration ONE pass over the atom's restricted residual --
// a workgroup takes MF_SPW signals, forms v_i = rk_i . u (its slice of Rk' u) and adds sum_i v_i rk_i into the next
// iterate (rows still L2-warm) -- plus a one-workgroup kernel for the norm, the sign (u . d_old) and sin^2 of the angle
// between successive iterates, which the host polls every few iterations.  Slow next to the Gram paths (an iteration per
// pass, fp32 atomics into the iterate) but it closes the shape the exact update used to refuse.
constexpr int MF_SPW = 16;

__global__ __launch_bounds__(256) void ksvd_mf_init_kernel(int atom, int n, const float* __restrict__ D, int ldd,
                                                           float* __restrict__ u, double* __restrict__ s2) {
    __shared__ double pr[256], red[16];
    double acc = 0.0;
    for (int f = threadIdx.x; f < n; f += 256) {
        const float d = D[(int64_t)atom * ldd + f];
        u[f] = d;
        acc += (double)d * (double)d;
    }
    const double tot = block_sum_d(acc, pr, red);
    if (threadIdx.x == 0) {
        s2[0] = tot;  // ||u||^2
        s2[1] = tot;  // u . d_old
        s2[2] = 1.0;  // sin^2 of the angle to the previous iterate: "not converged"
    }
}

__global__ __launch_bounds__(256) void ksvd_mf_iter_kernel(int atom, const float* __restrict__ R, int64_t ldr, int n, int k,
                                                           const int32_t* __restrict__ row_ptr,
                                                           const int32_t* __restrict__ entry,
                                                           const float* __restrict__ coef, const float* __restrict__ D,
                                                           int ldd, const float* __restrict__ uold,
                                                           const double* __restrict__ s2old, float* __restrict__ unew) {
    __shared__ double pr[256], red[16];
    __shared__ int64_t sg[MF_SPW];
    __shared__ float xs[MF_SPW], vs[MF_SPW];
    const int beg = row_ptr[atom], m = row_ptr[atom + 1] - beg;
    const int i0 = blockIdx.x * MF_SPW;
    if (i0 >= m) return;
    const int cnt = (m - i0 < MF_SPW) ? m - i0 : MF_SPW;
    const int tid = threadIdx.x;
    if (tid < cnt) {
        const int ss = entry[beg + i0 + tid];
        sg[tid] = (int64_t)(ss / k) * ldr;
        xs[tid] = coef[ss];
    }
    __syncthreads();
    const float* dold = D + (int64_t)atom * ldd;
    const double nrm2 = s2old[0];
    const double inv = (nrm2 > 0.0) ? 1.0 / sqrt(nrm2) : 0.0;
    for (int s = 0; s < cnt; ++s) {  // v_s = rk_s . u / ||u||   (rk = R_i + d_old x_i, ksvd.py:30-31)
        const float* Ri = R + sg[s];
        const float xo = xs[s];
        double dot = 0.0;
        for (int f = tid; f < n; f += 256) dot += (double)fmaf(dold[f], xo, Ri[f]) * (double)uold[f];
        const double tot = block_sum_d(dot, pr, red);
        if (tid == 0) vs[s] = (float)(tot * inv);
    }
    __syncthreads();
    for (int f = tid; f < n; f += 256) {  // u_new += sum_s v_s rk_s
        const float d = dold[f];
        float acc = 0.f;
        for (int s = 0; s < cnt; ++s) acc = fmaf(vs[s], fmaf(d, xs[s], R[sg[s] + f]), acc);
        atomicAdd(unew + f, acc);
    }
}

__global__ __launch_bounds__(256) void ksvd_mf_norm_kernel(int atom, int n, const float* __restrict__ D, int ldd,
                                                           const float* __restrict__ unew, const float* __restrict__ uold,
                                                           const double* __restrict__ s2old, double* __restrict__ s2new) {
    __shared__ double pr[256], red[16];
    double a = 0.0, b = 0.0, c = 0.0;
    for (int f = threadIdx.x; f < n; f += 256) {
        const double un = (double)unew[f];
        a += un * un;
        b += un * (double)D[(int64_t)atom * ldd + f];
        c += un * (double)uold[f];
    }
    a = block_sum_d(a, pr, red);
    b = block_sum_d(b, pr, red);
    c = block_sum_d(c, pr, red);
    if (threadIdx.x == 0) {
        const double den = a * s2old[0];
        s2new[0] = a;
        s2new[1] = b;
        s2new[2] = (den > 0.0) ? fmax(0.0, 1.0 - (c * c) / den) : 0.0;
    }
}

// doubles of nn_ksvd's per-atom state behind the exact update's work area: [4 scalars | s (256) | working d (256 floats)]
constexpr size_t NN_STATE_DOUBLES = 4 + 256 + 128;
static size_t exact_base_doubles(int n) {
    if (n <= 64) return (size_t)G64_MAX_PARTS * 4096 / 2 + 4096;  // fp32 partial Gram matrices of ksvd_gram64_kernel + their fp64 sum
    return (size_t)n * n;
}
// bytes of the pipelined sweep's link area for an index of nnz entries over K atoms (see ksvd_exact_sweep)
size_t ksvd_exact_link_bytes(int K, int64_t nnz) {
    return (size_t)XL_SH * 4096 * sizeof(float) +
           ((((size_t)K + 3) & ~(size_t)3) + (((size_t)2 * K + 3) & ~(size_t)3)) * sizeof(int32_t) +
           (size_t)nnz * (sizeof(int4) + sizeof(int32_t)) + (((size_t)nnz + 15) & ~(size_t)15);
}
size_t ksvd_exact_work_doubles(int n) {
    if (n <= 256) return exact_base_doubles(n) + NN_STATE_DOUBLES;
    // s2 (8) | M | v | u_raw (n floats) | second iterate of the matrix-free path (n floats) | its second s2 (8)
    return 8 + (size_t)TALL_MAX * TALL_MAX + TALL_MAX / 2 + ((size_t)n + 1) / 2 + 8 + ((size_t)n + 1) / 2 + 8;
}

static int ksvd_exact_sweep_tall(float* R, int64_t ldr, int n, int K, int k, const int32_t* row_ptr, const int32_t* entry,
                                 float* coef, double* work, float* D, float* Dnext, int64_t max_support,
                                 hipStream_t stream) {
    if (max_support <= 0) return LYS_OK;
    const int ldd = padded_features(n);
    static bool attr_set[64] = {};
    int dev = 0;
    LYS_CHECK_HIP(hipGetDevice(&dev));
    const size_t eig_lds = ((size_t)(EIG_M + 1) * TALL_MAX + 64 * 64) * sizeof(double);
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        LYS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ksvd_eig_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)eig_lds));
        attr_set[dev] = true;
    }
    double* s2 = work;
    double* M = work + 8;
    float* v = reinterpret_cast<float*>(M + (size_t)TALL_MAX * TALL_MAX);
    float* uraw = v + TALL_MAX;
    float* ub = uraw + 2 * (((size_t)n + 1) / 2) + 16;  // second iterate (after 8 spare doubles behind u_raw)
    double* s2b = reinterpret_cast<double*>(ub + 2 * (((size_t)n + 1) / 2));
    // atoms used by more than TALL_MAX signals take the matrix-free path: the host needs every atom's support size
    std::vector<int32_t> rp;
    if (max_support > TALL_MAX) {
        rp.resize((size_t)K + 1);
        LYS_CHECK_HIP(hipMemcpyAsync(rp.data(), row_ptr, ((size_t)K + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        LYS_CHECK_HIP(hipStreamSynchronize(stream));
    }
    const int64_t small_support = (max_support < TALL_MAX) ? max_support : TALL_MAX;
    const int mb = (int)((small_support + 63) / 64);
    const unsigned fchunks = (unsigned)((n + TALL_FCH - 1) / TALL_FCH);
    for (int a = 0; a < K; ++a) {
        const int64_t ma = rp.empty() ? small_support : (int64_t)(rp[a + 1] - rp[a]);
        if (ma <= 0 && !rp.empty()) continue;  // unused atom: keeps its column (ksvd_commit), nothing to apply
        if (ma > TALL_MAX) {
            // ---- matrix-free power iteration (see ksvd_mf_iter_kernel)
            float *uo = uraw, *un = ub;
            double *so = s2, *sn = s2b;
            hipLaunchKernelGGL(ksvd_mf_init_kernel, dim3(1), dim3(256), 0, stream, a, n, D, ldd, uo, so);
            const unsigned grid = (unsigned)((ma + MF_SPW - 1) / MF_SPW);
            constexpr int MAX_IT = 400, POLL = 4;
            for (int it = 0; it < MAX_IT; ++it) {
                LYS_CHECK_HIP(hipMemsetAsync(un, 0, (size_t)n * sizeof(float), stream));
                hipLaunchKernelGGL(ksvd_mf_iter_kernel, dim3(grid), dim3(256), 0, stream, a, R, ldr, n, k, row_ptr, entry, coef,
                                   D, ldd, uo, so, un);
                hipLaunchKernelGGL(ksvd_mf_norm_kernel, dim3(1), dim3(256), 0, stream, a, n, D, ldd, un, uo, so, sn);
                std::swap(uo, un);
                std::swap(so, sn);
                if ((it + 1) % POLL == 0) {
                    double h[3];
                    LYS_CHECK_HIP(hipMemcpyAsync(h, so, sizeof(h), hipMemcpyDeviceToHost, stream));
                    LYS_CHECK_HIP(hipStreamSynchronize(stream));
                    // successive iterates within 1e-6 rad (sin^2 < 1e-12), or a zero restricted residual
                    if (!(h[2] > 1e-12) || !(h[0] > 0.0)) break;
                }
            }
            hipLaunchKernelGGL(ksvd_tall_apply_kernel, dim3((unsigned)ma), dim3(256), 0, stream, a, R, ldr, n, k, row_ptr, entry,
                               coef, D, ldd, uo, so, Dnext);
            LYS_LAUNCH_CHECK();
            continue;
        }
        LYS_CHECK_HIP(hipMemsetAsync(work, 0, (8 + (size_t)small_support * small_support) * sizeof(double), stream));
        hipLaunchKernelGGL(ksvd_gram_t_kernel, dim3(fchunks, mb == 1 ? 1 : 10), dim3(256), 0, stream, a, R, ldr, n, k, row_ptr,
                           entry, coef, D, ldd, M);
        hipLaunchKernelGGL(ksvd_eig_kernel, dim3(1), dim3(256), eig_lds, stream, a, n, row_ptr, M, D, ldd, Dnext, 0, entry,
                           coef, v);
        hipLaunchKernelGGL(ksvd_tall_u_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a, R, ldr, n, k,
                           row_ptr, entry, coef, D, ldd, v, uraw, s2);
        hipLaunchKernelGGL(ksvd_tall_apply_kernel, dim3((unsigned)small_support), dim3(256), 0, stream, a, R, ldr, n, k, row_ptr,
                           entry, coef, D, ldd, uraw, s2, Dnext);
        LYS_LAUNCH_CHECK();
    }
    return ksvd_commit(n, K, row_ptr, Dnext, D, stream);
}

// ---------------------------------------------------------------------------------------------
// Non-negative K-SVD atom update (lyssa/dict_learning/ksvd.py:46-95, `nn_ksvd`).  Per atom, after the same rank-1 solve as
// the exact update (u in Dnext[atom], sign u . d_old >= 0; the reference's randomized_svd leaves the sign to chance):
//     d = max(u, 0), x = max(Rk'u, 0)                                  (:73-77; skip the atom if d'd or x'x <= eps, :79-82)
//     n_cycles times:  d = max(Rk x / x'x, 0),  x = max(Rk'd / d'd, 0)  (:84-88)
//     d /= ||d||, x *= ||d||;  D[:,k] = d, X[k,omega] = x, R[:,omega] = Rk - d x'   (:90-95)
// Rk = R[:,omega] + d_old x_old is never materialised: every pass rebuilds its rows from R, D[atom] and the stored
// coefficients, which stay untouched until the commit.  State of the current atom in `st` (doubles):
//   [0] x'x accumulator  [1] d'd  [2] skip flag  [3] x'x of the finished pass  [4 .. 4+n) sum_i x_i rk_i
// x iterates live in xbuf[e - row_ptr[atom]] (float, max_support entries).
// ---------------------------------------------------------------------------------------------
constexpr int NN_ST = 4;

// x_i = max(scale * rk_i . v, 0) with v = vsrc (first: u = Dnext[atom], scale 1; later: the working d, scale 1 / d'd)
template <int FB>
__global__ __launch_bounds__(256) void nn_x_kernel(int atom, int first, const float* __restrict__ R, int64_t ldr, int n, int k,
                                                   const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ entry,
                                                   const float* __restrict__ coef, const float* __restrict__ D, int ldd,
                                                   const float* __restrict__ vsrc /* n floats: u or the working d */,
                                                   float* __restrict__ xbuf, double* __restrict__ st) {
    __shared__ double s_sq[16];
    const int beg = row_ptr[atom], end = row_ptr[atom + 1];
    const int team = threadIdx.x >> 4, q = threadIdx.x & 15;
    const int gteam = blockIdx.x * 16 + team, nteams = gridDim.x * 16;
    if (beg + blockIdx.x * 16 >= end) return;
    if (!first && st[2] != 0.0) return;
    const float scale = first ? 1.f : (float)(1.0 / st[1]);
    float4 dold[FB], v[FB];
#pragma unroll
    for (int b = 0; b < FB; ++b) {
        const int f = 64 * b + 4 * q;
        dold[b] = v[b] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (f < n) {
            dold[b] = *reinterpret_cast<const float4*>(D + (int64_t)atom * ldd + f);
            v[b] = *reinterpret_cast<const float4*>(vsrc + f);
        }
    }
    double sq = 0.0;
    for (int e = beg + gteam; e < end; e += nteams) {
        const int ss = entry[e];
        const int64_t sig = ss / k;
        const float xo = coef[ss];
        float dot = 0.f;
#pragma unroll
        for (int b = 0; b < FB; ++b) {
            const int f = 64 * b + 4 * q;
            float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f < n) r = *reinterpret_cast<const float4*>(R + sig * ldr + f);
            dot = fmaf(fmaf(dold[b].x, xo, r.x), v[b].x, dot);
            dot = fmaf(fmaf(dold[b].y, xo, r.y), v[b].y, dot);
            dot = fmaf(fmaf(dold[b].z, xo, r.z), v[b].z, dot);
            dot = fmaf(fmaf(dold[b].w, xo, r.w), v[b].w, dot);
        }
        const float x = fmaxf(row16_sum(dot) * scale, 0.f);
        if (q == 0) xbuf[e - beg] = x;
        sq += (double)x * (double)x;
    }
    if (q == 0) s_sq[team] = sq;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int t = 0; t < 16; ++t) tot += s_sq[t];
        atomicAdd(st, tot);
    }
}

// st[4 + f] += sum_i x_i rk_i[f]
template <int FB>
__global__ __launch_bounds__(256) void nn_dacc_kernel(int atom, const float* __restrict__ R, int64_t ldr, int n, int k,
                                                      const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ entry,
                                                      const float* __restrict__ coef, const float* __restrict__ D, int ldd,
                                                      const float* __restrict__ xbuf, double* __restrict__ st) {
    __shared__ float s_acc[16][FB * 64];
    const int beg = row_ptr[atom], end = row_ptr[atom + 1];
    const int team = threadIdx.x >> 4, q = threadIdx.x & 15;
    const int gteam = blockIdx.x * 16 + team, nteams = gridDim.x * 16;
    if (beg + blockIdx.x * 16 >= end) return;
    if (st[2] != 0.0) return;
    float4 dold[FB], acc[FB];
#pragma unroll
    for (int b = 0; b < FB; ++b) {
        const int f = 64 * b + 4 * q;
        dold[b] = acc[b] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (f < n) dold[b] = *reinterpret_cast<const float4*>(D + (int64_t)atom * ldd + f);
    }
    for (int e = beg + gteam; e < end; e += nteams) {
        const int ss = entry[e];
        const int64_t sig = ss / k;
        const float xo = coef[ss];
        const float x = xbuf[e - beg];
#pragma unroll
        for (int b = 0; b < FB; ++b) {
            const int f = 64 * b + 4 * q;
            if (f < n) {
                const float4 r = *reinterpret_cast<const float4*>(R + sig * ldr + f);
                acc[b].x = fmaf(fmaf(dold[b].x, xo, r.x), x, acc[b].x);
                acc[b].y = fmaf(fmaf(dold[b].y, xo, r.y), x, acc[b].y);
                acc[b].z = fmaf(fmaf(dold[b].z, xo, r.z), x, acc[b].z);
                acc[b].w = fmaf(fmaf(dold[b].w, xo, r.w), x, acc[b].w);
            }
        }
    }
#pragma unroll
    for (int b = 0; b < FB; ++b) {
        s_acc[team][64 * b + 4 * q + 0] = acc[b].x;
        s_acc[team][64 * b + 4 * q + 1] = acc[b].y;
        s_acc[team][64 * b + 4 * q + 2] = acc[b].z;
        s_acc[team][64 * b + 4 * q + 3] = acc[b].w;
    }
    __syncthreads();
    for (int f = threadIdx.x; f < n; f += 256) {
        double tot = 0.0;
#pragma unroll
        for (int t = 0; t < 16; ++t) tot += (double)s_acc[t][f];
        atomicAdd(st + NN_ST + f, tot);
    }
}

// the serial scalar logic between the passes, one workgroup.  phase 0: after the first x pass (d = max(u, 0), the skip
// test of :79-82); phase 1: after an accumulate pass (d = max(s / x'x, 0)).  Both leave d (unnormalised) in `dwork`,
// d'd in st[1], and reset the accumulators.
__global__ __launch_bounds__(256) void nn_prep_kernel(int atom, int phase, int n, const int32_t* __restrict__ row_ptr,
                                                      const float* __restrict__ D, int ldd, float* __restrict__ Dnext,
                                                      float* __restrict__ dwork, double* __restrict__ st) {
    __shared__ double s_red[256];
    if (row_ptr[atom] >= row_ptr[atom + 1]) return;
    if (phase == 1 && st[2] != 0.0) return;
    const double xtx = st[0];
    double part = 0.0;
    for (int f = threadIdx.x; f < ldd; f += 256) {
        float d = 0.f;
        if (f < n) {
            d = (phase == 0) ? Dnext[(int64_t)atom * ldd + f] : (float)(st[NN_ST + f] / xtx);
            d = (d < 0.f) ? 0.f : d;   // NaN (x'x = 0 inside the loop: the reference divides by zero too) stays NaN
        }
        dwork[f] = d;
        part += (double)d * (double)d;
    }
    s_red[threadIdx.x] = part;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s_red[threadIdx.x] += s_red[threadIdx.x + o];
        __syncthreads();
    }
    const double dtd = s_red[0];
    const double eps = 2.220446049250313e-16;  // np.finfo('float').eps
    const bool skip = (phase == 0) && (dtd <= eps || xtx <= eps);
    __syncthreads();
    if (skip)  // the atom, its coefficients and the residual stay as they were (`continue`, :82)
        for (int f = threadIdx.x; f < ldd; f += 256) Dnext[(int64_t)atom * ldd + f] = D[(int64_t)atom * ldd + f];
    for (int f = threadIdx.x; f < n; f += 256) st[NN_ST + f] = 0.0;
    if (threadIdx.x == 0) {
        // the x'x of the latest x pass stays in st[0] until the NEXT accumulate pass has been turned into d (phase 1 reads
        // and clears it; the x pass that follows accumulates afresh)
        st[0] = (phase == 0 && !skip) ? xtx : 0.0;
        st[1] = dtd;
        st[3] = xtx;
        if (phase == 0) st[2] = skip ? 1.0 : 0.0;
    }
}

// d /= ||d||, x *= ||d||, coefficients and residual rows written (:90-95); the new atom goes to Dnext[atom].
template <int FB>
__global__ __launch_bounds__(256) void nn_commit_kernel(int atom, float* __restrict__ R, int64_t ldr, int n, int k,
                                                        const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ entry,
                                                        float* __restrict__ coef, const float* __restrict__ D, int ldd,
                                                        const float* __restrict__ dwork, float* __restrict__ Dnext,
                                                        const float* __restrict__ xbuf, double* __restrict__ st,
                                                        const int32_t* __restrict__ used_ptr) {
    const int beg = row_ptr[atom], end = row_ptr[atom + 1];
    const int team = threadIdx.x >> 4, q = threadIdx.x & 15;
    const int gteam = blockIdx.x * 16 + team, nteams = gridDim.x * 16;
    // used_ptr: non-empty for an atom used on ANY shard (== row_ptr on one GPU): workgroup 0 publishes the new atom also when
    // this shard holds none of its non-zeros
    if (used_ptr[atom] >= used_ptr[atom + 1] || st[2] != 0.0) return;
    if (blockIdx.x != 0 && beg + blockIdx.x * 16 >= end) return;
    const double nrm = sqrt(st[1]);
    if (blockIdx.x == 0 && threadIdx.x == 0) st[0] = 0.0;  // the last x pass's x'x is never consumed: clean for the next atom
    const float fn = (float)nrm, rn = (float)(1.0 / nrm);
    float4 dold[FB], d[FB];
#pragma unroll
    for (int b = 0; b < FB; ++b) {
        const int f = 64 * b + 4 * q;
        dold[b] = d[b] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (f < n) {
            dold[b] = *reinterpret_cast<const float4*>(D + (int64_t)atom * ldd + f);
            d[b] = *reinterpret_cast<const float4*>(dwork + f);
            d[b].x *= rn; d[b].y *= rn; d[b].z *= rn; d[b].w *= rn;
            if (blockIdx.x == 0 && team == 0) *reinterpret_cast<float4*>(Dnext + (int64_t)atom * ldd + f) = d[b];
        }
    }
    for (int e = beg + gteam; e < end; e += nteams) {
        const int ss = entry[e];
        const int64_t sig = ss / k;
        const float xo = coef[ss];
        const float xn = xbuf[e - beg] * fn;
#pragma unroll
        for (int b = 0; b < FB; ++b) {
            const int f = 64 * b + 4 * q;
            if (f < n) {
                const float4 r = *reinterpret_cast<const float4*>(R + sig * ldr + f);
                float4 o;
                o.x = fmaf(-d[b].x, xn, fmaf(dold[b].x, xo, r.x));
                o.y = fmaf(-d[b].y, xn, fmaf(dold[b].y, xo, r.y));
                o.z = fmaf(-d[b].z, xn, fmaf(dold[b].z, xo, r.z));
                o.w = fmaf(-d[b].w, xn, fmaf(dold[b].w, xo, r.w));
                *reinterpret_cast<float4*>(R + sig * ldr + f) = o;
            }
        }
        if (q == 0) coef[ss] = xn;
    }
}

template <int FB>
static int nn_atom_passes(int a, int nn_cycles, float* R, int64_t ldr, int n, int k, const int32_t* row_ptr,
                          const int32_t* entry, float* coef, const float* D, int ldd, float* Dnext, float* xbuf, double* st,
                          hipStream_t stream) {
    float* dwork = reinterpret_cast<float*>(st + NN_ST + 256);  // ldd <= 256 floats
    hipLaunchKernelGGL(nn_x_kernel<FB>, dim3(KSVD_BLOCKS), dim3(256), 0, stream, a, 1, R, ldr, n, k, row_ptr, entry, coef, D, ldd,
                       Dnext + (int64_t)a * ldd, xbuf, st);
    hipLaunchKernelGGL(nn_prep_kernel, dim3(1), dim3(256), 0, stream, a, 0, n, row_ptr, D, ldd, Dnext, dwork, st);
    for (int j = 0; j < nn_cycles; ++j) {
        hipLaunchKernelGGL(nn_dacc_kernel<FB>, dim3(KSVD_BLOCKS), dim3(256), 0, stream, a, R, ldr, n, k, row_ptr, entry, coef, D,
                           ldd, xbuf, st);
        hipLaunchKernelGGL(nn_prep_kernel, dim3(1), dim3(256), 0, stream, a, 1, n, row_ptr, D, ldd, Dnext, dwork, st);
        hipLaunchKernelGGL(nn_x_kernel<FB>, dim3(KSVD_BLOCKS), dim3(256), 0, stream, a, 0, R, ldr, n, k, row_ptr, entry, coef, D,
                           ldd, dwork, xbuf, st);
    }
    hipLaunchKernelGGL(nn_commit_kernel<FB>, dim3(KSVD_BLOCKS), dim3(256), 0, stream, a, R, ldr, n, k, row_ptr, entry, coef, D,
                       ldd, dwork, Dnext, xbuf, st, row_ptr);
    LYS_LAUNCH_CHECK();
    return LYS_OK;
}

// The same passes one at a time, for signal SHARDS (dist.nn_ksvd_cycle_sharded): between them the caller all-reduces
// st[0] (x'x of the pass just run) or st[4 .. 4+n) (sum x_i rk_i).  used_ptr: see nn_commit_kernel.
template <int FB>
static int nn_atom_phase(int phase, int a, float* R, int64_t ldr, int n, int k, const int32_t* row_ptr, const int32_t* used_ptr,
                         const int32_t* entry, float* coef, const float* D, int ldd, float* Dnext, float* xbuf, double* st,
                         hipStream_t stream) {
    float* dwork = reinterpret_cast<float*>(st + NN_ST + 256);
    switch (phase) {
        case 0:  // first x pass (u in Dnext[a])
            hipLaunchKernelGGL(nn_x_kernel<FB>, dim3(KSVD_BLOCKS), dim3(256), 0, stream, a, 1, R, ldr, n, k, row_ptr, entry, coef,
                               D, ldd, Dnext + (int64_t)a * ldd, xbuf, st);
            break;
        case 1:  // d = max(u, 0), skip test on the (reduced) x'x
            hipLaunchKernelGGL(nn_prep_kernel, dim3(1), dim3(256), 0, stream, a, 0, n, used_ptr, D, ldd, Dnext, dwork, st);
            break;
        case 2:
            hipLaunchKernelGGL(nn_dacc_kernel<FB>, dim3(KSVD_BLOCKS), dim3(256), 0, stream, a, R, ldr, n, k, row_ptr, entry, coef,
                               D, ldd, xbuf, st);
            break;
        case 3:  // d from the (reduced) sums, then the next x pass
            hipLaunchKernelGGL(nn_prep_kernel, dim3(1), dim3(256), 0, stream, a, 1, n, used_ptr, D, ldd, Dnext, dwork, st);
            hipLaunchKernelGGL(nn_x_kernel<FB>, dim3(KSVD_BLOCKS), dim3(256), 0, stream, a, 0, R, ldr, n, k, row_ptr, entry, coef,
                               D, ldd, dwork, xbuf, st);
            break;
        default:
            hipLaunchKernelGGL(nn_commit_kernel<FB>, dim3(KSVD_BLOCKS), dim3(256), 0, stream, a, R, ldr, n, k, row_ptr, entry,
                               coef, D, ldd, dwork, Dnext, xbuf, st, used_ptr);
    }
    LYS_LAUNCH_CHECK();
    return LYS_OK;
}

// One exact cycle on one GPU.  max_support: upper bound of |omega_a| over the atoms (sizes the Gram grid).
// nn_cycles >= 0: the non-negative variant (nn_ksvd, ksvd.py:46-95) with that many alternating projections per atom;
// xbuf: max_support floats (nn only).
int ksvd_exact_sweep(float* R, int64_t ldr, int n, int K, int k, const int32_t* row_ptr, const int32_t* entry,
                     float* coef, double* work, float* D, float* Dnext, int64_t max_support, hipStream_t stream,
                     int nn_cycles, float* xbuf, const int32_t* idx, void* link, int64_t link_nnz) {
    if (K <= 0) return LYS_OK;
    if (n > 256) {
        if (nn_cycles >= 0) {
            set_error("nn_ksvd: n = %d > 256 is outside the non-negative update", n);
            return LYS_ENOSUP;
        }
        return ksvd_exact_sweep_tall(R, ldr, n, K, k, row_ptr, entry, coef, work, D, Dnext, max_support, stream);
    }
    double* nn_st = work + exact_base_doubles(n);
    if (nn_cycles >= 0) LYS_CHECK_HIP(hipMemsetAsync(nn_st, 0, NN_STATE_DOUBLES * sizeof(double), stream));
    const int ldd = padded_features(n);
    const int fb = fb_of(n);
    static bool attr_set[64] = {};
    int dev = 0;
    LYS_CHECK_HIP(hipGetDevice(&dev));
    const int c_in_lds = (n <= 64) ? 1 : 0;
    const size_t eig_lds = ((size_t)(EIG_M + 1) * n + (c_in_lds ? (size_t)n * n : 0)) * sizeof(double);
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        LYS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ksvd_eig_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize,
                                          ((EIG_M + 1) * 256 + 64 * 64) * (int)sizeof(double)));
        attr_set[dev] = true;
    }
    const int nb = (n + 63) / 64;
    const unsigned gx = (unsigned)std::max<int64_t>(1, (max_support + GRAM_SPB - 1) / GRAM_SPB);
    // n <= 64: MFMA Gram kernel with per-workgroup partial sums (no memset, no atomics); slices of <= 320 signals
    constexpr int slice = 128;  // signals per Gram workgroup (measured at configs[1] with the 16-deep reduce kernel: 96 / 128 / 160 / 192 / 256 -> 37.9 / 36.6 / 37.6 / 37.8 / 39.4 ms per sweep)
    int parts = (int)std::min<int64_t>(G64_MAX_PARTS, std::max<int64_t>(1, (max_support + slice - 1) / slice));
    if ((max_support + parts - 1) / parts + 63 > G64_ROWS) parts = 0;  // an atom used by > 45k signals: atomics path
    if (n > 64) parts = 0;
    // idx given (the codes' atom indices), n <= 64, k <= 16, plain exact update: the pipelined sweep (exact_k1_kernel /
    // exact_k2_kernel: two dependent launches per atom instead of four)
    // (The single-wave solver runs EIG_PRE = 4 power steps before the Lanczos recurrence and tests the Ritz pair from step
    // EIG_MMIN = 2 on -- configs[1], same box: 0 / 1 / 2 / 3 pre-steps -> 4.7 / 4.0 / 3.0 / 3.0 Lanczos steps per solve, sweep
    // 24.0 / 24.8 / 22.6-23.3 / 23.8 ms; after the K1 rework 2 / test from 3 -> 21.3 ms, 3 / 2 -> 21.3, 4 / 2 -> 20.8.  Rounds 3-4
    // read both from the environment; they are constants of the kernel now.  The four-launch form below stays for the callers
    // the pipelined sweep does not cover: nn_ksvd, the sharded update, k > 16, n > 64.)
    if (idx && link && parts > 0 && k <= 16 && nn_cycles < 0) {
        double* Csum = work + (size_t)G64_MAX_PARTS * 4096 / 2;
        float* gpart = reinterpret_cast<float*>(work);
        // link area: [shared partials XL_SH x 64 x 64 floats | nsh K ints | prev / next used atom 2K ints | shrec nnz int4 |
        //             plink nnz ints | eflag nnz bytes]
        float* spart = reinterpret_cast<float*>(link);
        int32_t* nsh = reinterpret_cast<int32_t*>(spart + (size_t)XL_SH * 4096);
        int32_t* pn = nsh + (((size_t)K + 3) & ~(size_t)3);
        int4* shrec = reinterpret_cast<int4*>(pn + (((size_t)2 * K + 3) & ~(size_t)3));
        int32_t* plink = reinterpret_cast<int32_t*>(shrec + link_nnz);
        uint8_t* eflag = reinterpret_cast<uint8_t*>(plink + link_nnz);
        std::vector<int32_t> rp((size_t)K + 1), hns((size_t)K);
        // the index size first (its own short synchronisation): the link kernels below write row_ptr[K] entries of the link area
        LYS_CHECK_HIP(hipMemcpyAsync(rp.data(), row_ptr, rp.size() * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        LYS_CHECK_HIP(hipStreamSynchronize(stream));
        if (rp[K] > link_nnz) {
            set_error("ksvd_exact_sweep_idx: nnz_total = %lld < row_ptr[K] = %d", (long long)link_nnz, rp[K]);
            return LYS_EINVAL;
        }
        hipLaunchKernelGGL(exact_neighbours_kernel, dim3((unsigned)(K + 255) / 256), dim3(256), 0, stream, K, row_ptr, pn);
        hipLaunchKernelGGL(exact_flag_kernel, dim3(2048), dim3(256), 0, stream, K, k, row_ptr, entry, idx, coef, pn, eflag, plink);
        hipLaunchKernelGGL(exact_compact_kernel, dim3((unsigned)K), dim3(256), 0, stream, k, row_ptr, entry, plink, coef, shrec, nsh);
        LYS_LAUNCH_CHECK();
        // the second read-back: with the lists' bounds, the shared-row counts become kernel ARGUMENTS of the 2 L launches below
        LYS_CHECK_HIP(hipMemcpyAsync(hns.data(), nsh, hns.size() * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        LYS_CHECK_HIP(hipStreamSynchronize(stream));
        std::vector<int> used;
        for (int a = 0; a < K; ++a)
            if (rp[a + 1] > rp[a]) used.push_back(a);
        auto parts_of = [&](int a) {
            const int64_t m = rp[a + 1] - rp[a];
            return (int)std::min<int64_t>(G64_MAX_PARTS, std::max<int64_t>(1, (m + slice - 1) / slice));
        };
        const int L = (int)used.size();
        if (L > 0)
            hipLaunchKernelGGL(exact_k2_kernel, dim3(1u + (unsigned)parts_of(used[0])), dim3(256), 0, stream, -1, 0, used[0],
                               rp[used[0]], rp[used[0] + 1] - rp[used[0]], R, ldr, n, k, entry, eflag, coef, D, ldd, Dnext, gpart,
                               Csum, spart);
        for (int t = 0; t <= L && L > 0; ++t) {
            const int a = (t < L) ? used[t] : -1, p = (t > 0) ? used[t - 1] : -1;
            const int pbeg = (p >= 0) ? rp[p] : 0, pend = (p >= 0) ? rp[p + 1] : 0;
            const unsigned ab = (unsigned)std::min<int64_t>(XK1_APPLY_BLOCKS, ((int64_t)(pend - pbeg) + 15) / 16);
            const int nsa = (a >= 0 && p >= 0) ? hns[a] : 0;
            hipLaunchKernelGGL(exact_k1_kernel, dim3((unsigned)XK1_RED + XL_SH + ab), dim3(256), 0, stream, a, (a >= 0) ? rp[a] : 0,
                               (a >= 0) ? parts_of(a) : 0, nsa, p, pbeg, pend, R, ldr, n, k, entry, eflag, shrec, coef, D, ldd,
                               Dnext, gpart, Csum, spart);
            if (a < 0) break;
            const int nx = (t + 1 < L) ? used[t + 1] : -1;
            hipLaunchKernelGGL(exact_k2_kernel, dim3(1u + (unsigned)(nx >= 0 ? parts_of(nx) : 0)), dim3(256), 0, stream, a, nsa, nx,
                               (nx >= 0) ? rp[nx] : 0, (nx >= 0) ? rp[nx + 1] - rp[nx] : 0, R, ldr, n, k, entry, eflag, coef, D,
                               ldd, Dnext, gpart, Csum, spart);
        }
        LYS_LAUNCH_CHECK();
        return ksvd_commit(n, K, row_ptr, Dnext, D, stream);
    }
    for (int a = 0; a < K; ++a) {
        if (parts > 0) {
            hipLaunchKernelGGL(ksvd_gram64_kernel, dim3((unsigned)parts), dim3(256), 0, stream, a, R, ldr, n, k, row_ptr, entry,
                               coef, D, ldd, reinterpret_cast<float*>(work));
        } else {
            LYS_CHECK_HIP(hipMemsetAsync(work, 0, (size_t)n * n * sizeof(double), stream));
            hipLaunchKernelGGL(ksvd_gram_kernel, dim3(gx, nb * (nb + 1) / 2), dim3(256), 0, stream, a, R, ldr, n, k, row_ptr,
                               entry, coef, D, ldd, work);
        }
        if (parts > 0) {
            double* Csum = work + (size_t)G64_MAX_PARTS * 4096 / 2;
            hipLaunchKernelGGL(ksvd_gram64_reduce_kernel, dim3(16), dim3(256), 0, stream, a, row_ptr,
                               reinterpret_cast<const float*>(work), parts, Csum);
            hipLaunchKernelGGL(ksvd_eig64_kernel, dim3(1), dim3(256), 0, stream, a, n, row_ptr,
                               reinterpret_cast<const float*>(Csum), -1, D, ldd, Dnext);
        }
        else
            hipLaunchKernelGGL(ksvd_eig_kernel, dim3(1), dim3(256), eig_lds, stream, a, n, row_ptr, work, D, ldd, Dnext,
                               c_in_lds);
        if (nn_cycles >= 0) {
            int rc;
            switch (fb) {
                case 1: rc = nn_atom_passes<1>(a, nn_cycles, R, ldr, n, k, row_ptr, entry, coef, D, ldd, Dnext, xbuf, nn_st, stream); break;
                case 2: rc = nn_atom_passes<2>(a, nn_cycles, R, ldr, n, k, row_ptr, entry, coef, D, ldd, Dnext, xbuf, nn_st, stream); break;
                default: rc = nn_atom_passes<4>(a, nn_cycles, R, ldr, n, k, row_ptr, entry, coef, D, ldd, Dnext, xbuf, nn_st, stream); break;
            }
            if (rc) return rc;
            continue;
        }
        switch (fb) {
            case 1: hipLaunchKernelGGL(ksvd_exact_apply_kernel<1>, dim3(KSVD_BLOCKS), dim3(256), 0, stream, a, R, ldr, n, k, row_ptr, entry, coef, D, ldd, Dnext); break;
            case 2: hipLaunchKernelGGL(ksvd_exact_apply_kernel<2>, dim3(KSVD_BLOCKS), dim3(256), 0, stream, a, R, ldr, n, k, row_ptr, entry, coef, D, ldd, Dnext); break;
            default: hipLaunchKernelGGL(ksvd_exact_apply_kernel<4>, dim3(KSVD_BLOCKS), dim3(256), 0, stream, a, R, ldr, n, k, row_ptr, entry, coef, D, ldd, Dnext); break;
        }
        LYS_LAUNCH_CHECK();
    }
    return ksvd_commit(n, K, row_ptr, Dnext, D, stream);
}

// Per-atom halves of the exact update for signal shards (n <= 256): `ksvd_exact_gram` adds this shard's C = Rk Rk' into
// an fp64 n x n buffer (zeroed here) -- the caller all-reduces it over the ranks --, `ksvd_exact_update` runs the
// eigen-solve on the reduced matrix (replicated: every rank gets the same u) and applies it to the local rows.
int ksvd_exact_gram(int atom, const float* R, int64_t ldr, int n, int k, const int32_t* row_ptr, const int32_t* entry,
                    const float* coef, const float* D, double* C, int64_t max_support, hipStream_t stream) {
    if (n > 256) {
        set_error("sharded exact ksvd needs n <= 256 (n = %d)", n);
        return LYS_ENOSUP;
    }
    const int ldd = padded_features(n), nb = (n + 63) / 64;
    const unsigned gx = (unsigned)std::max<int64_t>(1, (max_support + GRAM_SPB - 1) / GRAM_SPB);
    LYS_CHECK_HIP(hipMemsetAsync(C, 0, (size_t)n * n * sizeof(double), stream));
    hipLaunchKernelGGL(ksvd_gram_kernel, dim3(gx, nb * (nb + 1) / 2), dim3(256), 0, stream, atom, R, ldr, n, k, row_ptr, entry,
                       coef, D, ldd, C);
    LYS_LAUNCH_CHECK();
    return LYS_OK;
}

// `used_ptr`: a row_ptr-like array that is non-empty for atoms used on ANY rank (the eigen-solve and the atom write
// must run on every rank, also on one whose shard does not use the atom); `row_ptr` is the local index.
int ksvd_exact_update(int atom, float* R, int64_t ldr, int n, int k, const int32_t* row_ptr, const int32_t* used_ptr,
                      const int32_t* entry, float* coef, const double* C, const float* D, float* Dnext,
                      hipStream_t stream) {
    const int fb = fb_of(n);
    if (!fb) {
        set_error("sharded exact ksvd needs n <= 256 (n = %d)", n);
        return LYS_ENOSUP;
    }
    const int ldd = padded_features(n);
    static bool attr_set[64] = {};
    int dev = 0;
    LYS_CHECK_HIP(hipGetDevice(&dev));
    const int c_in_lds = (n <= 64) ? 1 : 0;
    const size_t eig_lds = ((size_t)(EIG_M + 1) * n + (c_in_lds ? (size_t)n * n : 0)) * sizeof(double);
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        LYS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ksvd_eig_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize,
                                          ((EIG_M + 1) * 256 + 64 * 64) * (int)sizeof(double)));
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL(ksvd_eig_kernel, dim3(1), dim3(256), eig_lds, stream, atom, n, used_ptr, C, D, ldd, Dnext, c_in_lds);
    switch (fb) {
        case 1: hipLaunchKernelGGL(ksvd_exact_apply_kernel<1>, dim3(KSVD_BLOCKS), dim3(256), 0, stream, atom, R, ldr, n, k, row_ptr, entry, coef, D, ldd, Dnext); break;
        case 2: hipLaunchKernelGGL(ksvd_exact_apply_kernel<2>, dim3(KSVD_BLOCKS), dim3(256), 0, stream, atom, R, ldr, n, k, row_ptr, entry, coef, D, ldd, Dnext); break;
        default: hipLaunchKernelGGL(ksvd_exact_apply_kernel<4>, dim3(KSVD_BLOCKS), dim3(256), 0, stream, atom, R, ldr, n, k, row_ptr, entry, coef, D, ldd, Dnext); break;
    }
    LYS_LAUNCH_CHECK();
    return LYS_OK;
}

// ---- n > 256 on signal SHARDS (round 4): the n x n Gram matrix is no statistic one can all-reduce (24 635^2 doubles for
// LC-KSVD's stack) and the column Gram matrix couples the shards; the matrix-free power iteration of ksvd_exact_sweep_tall
// needs ONE n-vector from the other shards per iteration: u' = sum over shards of sum_i (rk_i . u) rk_i.  Phases of one atom
// (dist.ksvd_exact_cycle_sharded_mf):
//   0  u = d_old, s2 = (||u||^2, u . d_old, 1)                                            (replicated)
//   1  un = this shard's sum_i (rk_i . u / ||u||) rk_i      -> the caller all-reduces the n floats of un
//   2  s2' = (||un||^2, un . d_old, sin^2 of the angle to u); u = un, s2 = s2'            (replicated; the caller reads s2[2])
//   3  x_i = rk_i . u / ||u||, R_i = rk_i - u x_i / ||u|| on the local rows, the new atom on every shard
// Work layout = ksvd_exact_sweep_tall's: [s2 (8 doubles) | M | v | u | 8 spare doubles | un | s2'].
struct MfLayout {
    double* s2;
    float* u;
    float* un;
    double* s2n;
};
static MfLayout mf_layout(double* work, int n) {
    MfLayout l;
    l.s2 = work;
    double* M = work + 8;
    float* v = reinterpret_cast<float*>(M + (size_t)TALL_MAX * TALL_MAX);
    l.u = v + TALL_MAX;
    l.un = l.u + 2 * (((size_t)n + 1) / 2) + 16;
    l.s2n = reinterpret_cast<double*>(l.un + 2 * (((size_t)n + 1) / 2));
    return l;
}
void ksvd_exact_mf_offsets(int n, int64_t* out4) {  // byte offsets of u, un, s2, s2' in the work area
    const uintptr_t base = (uintptr_t)1 << 30;
    MfLayout l = mf_layout(reinterpret_cast<double*>(base), n);
    out4[0] = (int64_t)(reinterpret_cast<uintptr_t>(l.u) - base);
    out4[1] = (int64_t)(reinterpret_cast<uintptr_t>(l.un) - base);
    out4[2] = 0;
    out4[3] = (int64_t)(reinterpret_cast<uintptr_t>(l.s2n) - base);
}

__global__ __launch_bounds__(256) void ksvd_mf_atom_kernel(int atom, int n, const float* __restrict__ D, int ldd,
                                                           const float* __restrict__ uraw, const double* __restrict__ s2,
                                                           float* __restrict__ Dnext) {
    const double s2v = s2[0];
    const float inv = (s2v > 0.0) ? (float)((s2[1] < 0.0 ? -1.0 : 1.0) / sqrt(s2v)) : 0.f;  // ksvd_tall_apply_kernel's
    for (int f = blockIdx.x * 256 + threadIdx.x; f < n; f += gridDim.x * 256)
        Dnext[(int64_t)atom * ldd + f] = (s2v > 0.0) ? uraw[f] * inv : D[(int64_t)atom * ldd + f];
}

int ksvd_exact_mf_phase(int phase, int atom, float* R, int64_t ldr, int n, int k, const int32_t* row_ptr, const int32_t* entry,
                        float* coef, double* work, const float* D, float* Dnext, int64_t local_support, hipStream_t stream) {
    const int ldd = padded_features(n);
    MfLayout l = mf_layout(work, n);
    switch (phase) {
        case 0:
            hipLaunchKernelGGL(ksvd_mf_init_kernel, dim3(1), dim3(256), 0, stream, atom, n, D, ldd, l.u, l.s2);
            break;
        case 1:
            LYS_CHECK_HIP(hipMemsetAsync(l.un, 0, (size_t)n * sizeof(float), stream));
            if (local_support > 0)
                hipLaunchKernelGGL(ksvd_mf_iter_kernel, dim3((unsigned)((local_support + MF_SPW - 1) / MF_SPW)), dim3(256), 0, stream,
                                   atom, R, ldr, n, k, row_ptr, entry, coef, D, ldd, l.u, l.s2, l.un);
            break;
        case 2:
            hipLaunchKernelGGL(ksvd_mf_norm_kernel, dim3(1), dim3(256), 0, stream, atom, n, D, ldd, l.un, l.u, l.s2, l.s2n);
            LYS_CHECK_HIP(hipMemcpyAsync(l.u, l.un, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, stream));
            LYS_CHECK_HIP(hipMemcpyAsync(l.s2, l.s2n, 3 * sizeof(double), hipMemcpyDeviceToDevice, stream));
            break;
        default:
            if (local_support > 0)
                hipLaunchKernelGGL(ksvd_tall_apply_kernel, dim3((unsigned)local_support), dim3(256), 0, stream, atom, R, ldr, n, k,
                                   row_ptr, entry, coef, D, ldd, l.u, l.s2, Dnext);
            hipLaunchKernelGGL(ksvd_mf_atom_kernel, dim3((unsigned)std::min(64, (n + 255) / 256)), dim3(256), 0, stream, atom, n, D,
                               ldd, l.u, l.s2, Dnext);
    }
    LYS_LAUNCH_CHECK();
    return LYS_OK;
}

// nn_ksvd on signal shards, one phase of one atom (see nn_atom_phase): phase -1 zeroes the state (once per cycle), 0 runs the
// replicated eigen-solve on the reduced Gram matrix C (u -> Dnext[atom]) and the first x pass, 1..4 the remaining passes.
size_t nn_ksvd_state_offset_doubles(int n) { return exact_base_doubles(n); }
int nn_ksvd_phase(int phase, int atom, float* R, int64_t ldr, int n, int k, const int32_t* row_ptr, const int32_t* used_ptr,
                  const int32_t* entry, float* coef, const double* C, double* work, float* xbuf, const float* D, float* Dnext,
                  hipStream_t stream) {
    const int fb = fb_of(n);
    if (!fb) {
        set_error("sharded nn_ksvd needs n <= 256 (n = %d)", n);
        return LYS_ENOSUP;
    }
    double* st = work + exact_base_doubles(n);
    if (phase < 0) {
        LYS_CHECK_HIP(hipMemsetAsync(st, 0, NN_STATE_DOUBLES * sizeof(double), stream));
        return LYS_OK;
    }
    const int ldd = padded_features(n);
    if (phase == 0) {
        static bool attr_set[64] = {};
        int dev = 0;
        LYS_CHECK_HIP(hipGetDevice(&dev));
        const int c_in_lds = (n <= 64) ? 1 : 0;
        const size_t eig_lds = ((size_t)(EIG_M + 1) * n + (c_in_lds ? (size_t)n * n : 0)) * sizeof(double);
        if (dev >= 0 && dev < 64 && !attr_set[dev]) {
            LYS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ksvd_eig_kernel),
                                              hipFuncAttributeMaxDynamicSharedMemorySize,
                                              ((EIG_M + 1) * 256 + 64 * 64) * (int)sizeof(double)));
            attr_set[dev] = true;
        }
        hipLaunchKernelGGL(ksvd_eig_kernel, dim3(1), dim3(256), eig_lds, stream, atom, n, used_ptr, C, D, ldd, Dnext, c_in_lds);
    }
    switch (fb) {
        case 1: return nn_atom_phase<1>(phase, atom, R, ldr, n, k, row_ptr, used_ptr, entry, coef, D, ldd, Dnext, xbuf, st, stream);
        case 2: return nn_atom_phase<2>(phase, atom, R, ldr, n, k, row_ptr, used_ptr, entry, coef, D, ldd, Dnext, xbuf, st, stream);
        default: return nn_atom_phase<4>(phase, atom, R, ldr, n, k, row_ptr, used_ptr, entry, coef, D, ldd, Dnext, xbuf, st, stream);
    }
}


int ksvd_sweep_fused(float* R, int64_t ldr, int n, int K, int k, const int32_t* row_ptr, const int32_t* entry,
                     const int32_t* idx, float* coef, double* sbuf, float* D, float* Dnext, hipStream_t stream) {
    LYS_CHECK_HIP(hipMemsetAsync(sbuf, 0, (size_t)K * (n + 1) * sizeof(double), stream));
    // (handing every launch its segment bounds by value -- one small D2H copy per cycle -- was measured: 11.59 against 11.55 ms
    // per sweep, the row_ptr load is not on the critical path; removed in round 5)
    const int32_t* rp_host = nullptr;
    for (int a = 0; a <= K; ++a) {
        const int rc = ksvd_fused_step(a, K, R, ldr, n, k, row_ptr, entry, idx, coef, sbuf, D, Dnext, stream, rp_host);
        if (rc) return rc;
    }
    return ksvd_commit(n, K, row_ptr, Dnext, D, stream);
}

static int ksvd_sweep_eager(float* R, int64_t ldr, int n, int K, int k, const int32_t* row_ptr, const int32_t* entry,
                            float* coef, double* sbuf, float* D, float* Dnext, hipStream_t stream) {
    LYS_CHECK_HIP(hipMemsetAsync(sbuf, 0, (size_t)K * (n + 1) * sizeof(double), stream));
    for (int a = 0; a < K; ++a) {
        int rc = ksvd_atom_accumulate(a, R, ldr, n, k, row_ptr, entry, coef, sbuf, stream);
        if (rc) return rc;
        rc = ksvd_atom_apply(a, R, ldr, n, k, row_ptr, entry, coef, sbuf, D, Dnext, stream);
        if (rc) return rc;
    }
    return ksvd_commit(n, K, row_ptr, Dnext, D, stream);
}

int ksvd_sweep(float* R, int64_t ldr, int n, int K, int k, const int32_t* row_ptr, const int32_t* entry, float* coef,
               double* sbuf, float* D, float* Dnext, hipStream_t stream) {
    // (replaying the 2 K launches as one hipGraph measured 17.8 against 16.4 ms eager at config 2 -- the atom kernels' own
    // latency chains dominate, not the launches -- and was removed in round 5)
    return ksvd_sweep_eager(R, ldr, n, K, k, row_ptr, entry, coef, sbuf, D, Dnext, stream);
}


}  // namespace lys
