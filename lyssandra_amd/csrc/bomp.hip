// Batch-OMP greedy stage for gfx950 -- restates lyssa/sparse_coding.py:302-367 (`batch_omp`).
//
// One 64-lane wavefront per signal.  The K correlations a[] live in registers (R = Kp/64 per lane), the
// selected-atom state is kept in the *progressive* (orthogonalised) form so that no triangular solve and
// no re-read of earlier Gram rows is needed inside the loop:
//
//     step j:  kk   = argmax |a|                      (first maximum wins, like np.argmax :322)
//              stop if kk was selected before          (:323-325)
//              w_i  = p_i[kk]              (i<j)       == L^-1 G[Dx,kk]            (:341 / :331)
//              vs   = 1 - sum w_i^2 ; stop if vs<eps   unit Gram diagonal is HARD-CODED (:333-335,343-346)
//              rho  = sqrt(vs)                         new Cholesky row is [w, rho]  (:337-338,348-349)
//              t_j  = a[kk] / rho                      == (L^-1 a0[Dx])_j           (:353)
//              p_j  = (G[kk,:] - sum_i w_i p_i) / rho  == column j of G[:,Dx] L^-T
//              a   -= t_j p_j                          == a0 - G[:,Dx] z             (:359)
//     end:     z    = L^-T t                           second triangular solve       (:354)
//
// Algebraically identical to the reference for ANY dictionary (also non-normalised ones: the reference's
// L is the Cholesky factor of G[Dx,Dx] with its diagonal replaced by 1, and so is ours), but only j FMAs per
// correlation per step and one 4-KB Gram row read per step.  Register budget at K=1024, k=10:
// (k+1)*16 = 176 VGPRs of vectors -> 2 waves/SIMD.
//
// Element r = c*V + e of lane l holds atom c*64*V + l*V + e  (V = min(R,4): one coalesced dwordx4 per chunk).
#include "common.h"

namespace lys {

int bomp_wave2_launch(int Kp, const float* alpha0, const float* G, int64_t N, int k, int32_t* idx, float* coef,
                      int32_t* nnz, hipStream_t stream, int unit_diag);  // bomp_wave.hip

template <int R>
struct Lay {
    static constexpr int V = (R >= 4) ? 4 : R;
    static constexpr int C = R / V;
    static constexpr int Kp = 64 * R;
    __device__ static __forceinline__ int atom(int r, int lane) { return (r / V) * (64 * V) + lane * V + (r % V); }
    __device__ static __forceinline__ int lane_of(int atom) { return (atom / V) & 63; }
    __device__ static __forceinline__ int reg_of(int atom) { return (atom / (64 * V)) * V + (atom % V); }
};

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// STREAM: the row is read exactly once (alpha0) -> non-temporal load, so it does not evict Gram rows from L2
template <int R, bool STREAM = false>
__device__ __forceinline__ void load_row(const float* __restrict__ row, int lane, float (&v)[R]) {
    using L = Lay<R>;
    if constexpr (L::V == 4) {
#pragma unroll
        for (int c = 0; c < L::C; ++c) {
            const f32x4* ptr = reinterpret_cast<const f32x4*>(row) + (c * 64 + lane);
            const f32x4 t = STREAM ? __builtin_nontemporal_load(ptr) : *ptr;
            v[4 * c + 0] = t.x;
            v[4 * c + 1] = t.y;
            v[4 * c + 2] = t.z;
            v[4 * c + 3] = t.w;
        }
    } else if constexpr (L::V == 2) {
        const f32x2* ptr = reinterpret_cast<const f32x2*>(row) + lane;
        const f32x2 t = STREAM ? __builtin_nontemporal_load(ptr) : *ptr;
        v[0] = t.x;
        v[1] = t.y;
    } else {
        v[0] = STREAM ? __builtin_nontemporal_load(row + lane) : row[lane];
    }
}

// Uniform extraction  w[i] = p[i][rr](lane L)  for NLDS <= i < J, with rr only known at run time (wave-uniform), as
// an if-chain over rr: every register index static, one v_readlane per vector.  Used for k > 10 (many vectors per
// step); for k <= 10 the VGPR index mode (below) is faster.
template <int R, int KMAX, int NLDS, int J, int RR, class PV>
__device__ __forceinline__ void extract_case(const PV (&p)[(KMAX - 1 - NLDS) > 0 ? (KMAX - 1 - NLDS) : 1], int rr, int L,
                                             float (&w)[KMAX]) {
    if constexpr (RR < R) {
        if (rr == RR) {
#pragma unroll
            for (int i = NLDS; i < J; ++i) w[i] = readlane_f(p[i - NLDS][RR], L);
        } else {
            extract_case<R, KMAX, NLDS, J, RR + 1, PV>(p, rr, L, w);
        }
    }
}

// registers 4c..4c+3 of lane Lo, c only known at run time (wave-uniform): static if-chain keeps every register index static
template <int R, int C>
__device__ __forceinline__ void argmax_group_case(const float (&a)[R], int csel, int Lo, unsigned mbits, int& rsel,
                                                  unsigned& vsel) {
    if constexpr (4 * C < R) {
        if (csel == C) {
#pragma unroll
            for (int q = 3; q >= 0; --q) {
                const unsigned sv = (unsigned)__builtin_amdgcn_readlane(__builtin_bit_cast(int, a[4 * C + q]), Lo);
                const bool hit = (sv & 0x7fffffffu) == mbits;
                rsel = hit ? 4 * C + q : rsel;
                vsel = hit ? sv : vsel;
            }
        } else {
            argmax_group_case<R, C + 1>(a, csel, Lo, mbits, rsel, vsel);
        }
    }
}

template <int R>
__device__ __forceinline__ bool wave_argmax(const float (&a)[R], int lane, int& kk, float& akk, int& Lown, int& rown,
                                            float& mabs) {
    using L = Lay<R>;
    constexpr bool GROUPED = (R >= 8) && (R % 4 == 0);  // resolve the owner's register in two rounds of 4 readlanes
    constexpr int NG = GROUPED ? R / 4 : 1;
    float m4[NG];
    float best;
    if constexpr (GROUPED) {
        // max |.| of each 4-register group: v_max3_f32 + v_max_f32 with |.| source modifiers (written as asm: the
        // compiler canonicalises every fabsf() operand of fmaxf with an extra v_max |x|,|x|)
#pragma unroll
        for (int c = 0; c < NG; ++c) {
            float t3;
            asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(t3) : "v"(a[4 * c]), "v"(a[4 * c + 1]), "v"(a[4 * c + 2]));
            asm("v_max_f32_e64 %0, %1, |%2|" : "=v"(m4[c]) : "v"(t3), "v"(a[4 * c + 3]));
        }
        if constexpr (NG == 4) {
            best = fmaxf(fmaxf(fmaxf(m4[0], m4[1]), m4[2]), m4[3]);  // v_max3 + v_max (operands are already canonical)
        } else {
            best = m4[0];
#pragma unroll
            for (int c = 1; c < NG; ++c) best = fmaxf(best, m4[c]);
        }
    } else {
        best = fabsf(a[0]);
#pragma unroll
        for (int r = 1; r < R; ++r) best = fmaxf(best, fabsf(a[r]));
    }
    const float m = wave_max_f(best);
    mabs = m;
    const unsigned long long bal = __ballot(best == m);
    if (bal == 0ull) return false;  // NaN correlations: nothing sensible to select
    const unsigned mbits = __builtin_bit_cast(unsigned, m);
    if (__popcll(bal) == 1) {
        // common case: a single lane owns the maximum -> resolve its register with scalar compares
        const int Lo = __builtin_ctzll(bal);
        int rsel = 0;
        unsigned vsel = 0;
        if constexpr (GROUPED) {
            int csel = 0;
#pragma unroll
            for (int c = NG - 1; c >= 0; --c) {
                const unsigned sv = (unsigned)__builtin_amdgcn_readlane(__builtin_bit_cast(int, m4[c]), Lo);
                csel = (sv == mbits) ? c : csel;  // first (lowest) group that holds the maximum
            }
            argmax_group_case<R, 0>(a, csel, Lo, mbits, rsel, vsel);
        } else {
#pragma unroll
            for (int r = R - 1; r >= 0; --r) {
                const unsigned sv = (unsigned)__builtin_amdgcn_readlane(__builtin_bit_cast(int, a[r]), Lo);
                const bool hit = (sv & 0x7fffffffu) == mbits;
                rsel = hit ? r : rsel;
                vsel = hit ? sv : vsel;
            }
        }
        Lown = Lo;
        rown = rsel;
        kk = L::atom(rsel, Lo);
        akk = __builtin_bit_cast(float, vsel);
        return true;
    }
    // ties across lanes (duplicate atoms, all-zero signal): lowest atom index wins, exactly like np.argmax
    int cand = 0x7fffffff;
#pragma unroll
    for (int r = R - 1; r >= 0; --r) {
        const bool hit = fabsf(a[r]) == m;
        cand = hit ? L::atom(r, lane) : cand;
    }
    kk = wave_min_i(cand);
    Lown = L::lane_of(kk);
    rown = L::reg_of(kk);
    float v = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) v = (r == rown) ? a[r] : v;
    akk = readlane_f(v, Lown);
    return true;
}

// NLDS of the k-1 orthogonalised vectors (the oldest ones, p_0..p_{NLDS-1}) live in LDS instead of VGPRs: each
// costs 4 KB of LDS per wave and 4 ds_read_b128 per step, and frees 16 VGPRs -- at K=1024, k=10 three of them
// bring the kernel under 168 VGPRs = 3 waves/SIMD, which is what hides the Gram-row fetch latency.
template <int R, int KMAX, int NLDS>
struct OmpState {
    float a[R];                                                     // current correlations
    // p_i for i >= NLDS (the last selection needs none).  A true vector type: element rr of a vector can be read with
    // a run-time (wave-uniform) rr through the VGPR index mode (s_set_gpr_idx_on), which a float[R] array cannot
    typedef float pvec_t __attribute__((ext_vector_type(R)));
    pvec_t p[(KMAX - 1 - NLDS) > 0 ? (KMAX - 1 - NLDS) : 1];
    float Lrow[KMAX];                                               // Lrow[j] lane i (<j) = L[j][i]
    float tv;                                                       // lane j = t_j
    float rinv;                                                     // lane j = 1/rho_j
    float m0;                                                       // NOISE_REL * max |alpha0| (noise floor)
    int dxv;                                                        // lane j = Dx[j]
    int nsel;
};

// Once the largest correlation has dropped below NOISE_REL * max|alpha0| the residual is fp32 rounding noise of the
// alpha0 GEMM: the float64 reference goes on selecting atoms there with coefficients of order 1e-16 (SURVEY
// appendix A, "exactly representable signal"); in fp32 such picks carry 1e-7-sized coefficients that an
// ill-conditioned support amplifies into the significant ones.  The engine stops instead (dense results agree with
// the reference to 1e-5; nnz is smaller).  Never reached on signals with a real residual.
constexpr float NOISE_REL = 4e-6f;

// Steps J..KMAX-1 as a compile-time recursion (a loop with early exits around the convergent cross-lane
// operations is not unrolled by the compiler, which would push p[][] to scratch).
// VAR != 0 are timing ablations (wrong results on purpose; reachable only through lys_debug_bomp_variant):
//   1: Gram row index forced to kk & 7 (cache-hot rows)   2: no orthogonalisation FMAs   3: IEEE sqrt + divide
//   4: Gram rows restricted to kk & 255 (1 MB, L2-resident)   5: kk & 63 (256 KB)   6/7/8: 3 MB / 2 MB / 3.5 MB
// The Gram-row variants (1, 4..8) produce garbage correlations, so they also switch the data-dependent exits off and
// keep the values bounded: every signal runs all k steps, like a real Gaussian signal does (otherwise the wrong rows
// make the next argmax re-select the same atom and the kernel "speeds up" by stopping after two steps).
// FAST: k == KMAX and the unit Gram diagonal are known at compile time (the 'bomp' headline launch): the per-step
// `J >= k`, `J + 1 < k` and `unit_diag` tests and their scalar branches disappear.
template <int R, int KMAX, int NLDS, int J, int VAR = 0, bool FAST = false>
__device__ __forceinline__ void omp_steps(OmpState<R, KMAX, NLDS>& s, const float* __restrict__ G, int k, int lane,
                                          f32x4* __restrict__ lds /* [NLDS][R/4][64] of this wave */, int unit_diag_rt) {
    using L = Lay<R>;
    const bool unit_diag = FAST ? true : (unit_diag_rt != 0);
    constexpr bool FAKE_G = (VAR == 1) || (VAR >= 4 && VAR <= 8);
    if constexpr (J < KMAX) {
        if (!FAST && J >= k) return;
        int kk, Lown, rown;
        float akk, mabs;
        if (!wave_argmax<R>(s.a, lane, kk, akk, Lown, rown, mabs)) return;
        // noise floor: both sides are non-negative wave-uniform floats, so the comparison runs on their bit patterns in
        // the scalar unit (s.m0 holds the bits of NOISE_REL * max|alpha0|, computed once)
        if constexpr (J == 0) {
            s.m0 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, NOISE_REL * mabs)));
        } else {
            if (!FAKE_G && __builtin_bit_cast(unsigned, mabs) < __builtin_bit_cast(unsigned, s.m0)) return;
        }
        // re-selection => stop (sparse_coding.py:323-325); lanes >= J still hold dxv = -1, which never equals kk
        if (!FAKE_G && __ballot(s.dxv == kk) != 0ull) return;
        // The vector update is only needed if another selection follows: the reference's last
        // `a = a0 - G[:,Dx] z` (:359) is never read.  (J + 1 < k is wave-uniform; for J == KMAX-1 it is
        // statically false, so p[KMAX-1] never exists.)
        const bool more = (J + 1 < KMAX) && (FAST || J + 1 < k);
        // Gram row of the new atom (G is symmetric: row kk == column kk), issued before the scalar work
        float g[R];
        if (!more) {
            // g is never read on this path: give the registers a "definition" so that the compiler does not zero them
#pragma unroll
            for (int r = 0; r < R; ++r) asm volatile("" : "=v"(g[r]));
        }
        if (more)
            load_row<R>(G + (int64_t)(VAR == 1 ? (kk & 7) : VAR == 4 ? (kk & 255) : VAR == 5 ? (kk & 63)
                                      : VAR == 6 ? (kk % 768) : VAR == 7 ? (kk & 511) : VAR == 8 ? (kk % 896) : kk) * L::Kp,
                        lane, g);

        float w[KMAX];
        if constexpr (NLDS > 0) {
            // element kk of an LDS-resident vector: one broadcast read
            const float* lf = reinterpret_cast<const float*>(lds);
#pragma unroll
            for (int i = 0; i < (J < NLDS ? J : NLDS); ++i) {
                const float v = lf[((i * L::C + (rown >> 2)) * 64 + Lown) * 4 + (rown & 3)];
                w[i] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
            }
        }
        if constexpr (J > NLDS) {
            if constexpr (KMAX <= 10) {
                // w[i] = p_i[rown](lane Lown): indexed v_mov + v_readlane per vector, no branch tree (0.970 -> 0.945 ms
                // at K=1024, k=10 against the static if-chain; at k=20 the if-chain wins: one VALU op per vector)
                const int rr = __builtin_amdgcn_readfirstlane(rown);
#pragma unroll
                for (int i = NLDS; i < J; ++i) w[i] = readlane_f(s.p[i - NLDS][rr], Lown);
            } else {
                extract_case<R, KMAX, NLDS, J, 0>(s.p, rown, Lown, w);
            }
        }
        // Cholesky pivot: batch_omp hard-codes a unit Gram diagonal (:333-349); 'omp' (`_omp`, :44-52) inverts the
        // true G[Dx,Dx], i.e. uses G[kk][kk] (wave-uniform scalar load, only taken on the 'omp' path)
        const float gkk = unit_diag ? 1.f : G[(int64_t)kk * L::Kp + kk];
        float vs = gkk;
#pragma unroll
        for (int i = 0; i < J; ++i) vs = fmaf(-w[i], w[i], vs);
        if constexpr (FAKE_G) vs = fmaxf(fabsf(vs), 0.25f) < 4.f ? fmaxf(fabsf(vs), 0.25f) : 4.f;
        if ((J > 0 || !unit_diag) && vs < EPS32_F * gkk) return;  // reference: vs < eps (:335,345) / singular G (:48-51)
        // 1/sqrt(vs): hardware rsq (1 ulp) + one Newton step instead of an IEEE sqrt and an IEEE divide
        float inv;
        if constexpr (VAR == 3) {
            inv = 1.f / sqrtf(vs);
        } else {
            // hardware rsq: 1 ulp, i.e. the rounding level of every other fp32 operation of the step (a Newton step
            // on top was measured to change no support and no coefficient beyond 2e-7; it cost 4 VALU ops per step)
            inv = __builtin_amdgcn_rsqf(vs);
        }
        const float t = FAKE_G ? 1e-3f * akk * inv : akk * inv;

        if constexpr (J + 1 < KMAX) {
            if (more) {
                if constexpr (NLDS > 0) {
#pragma unroll
                    for (int c = 0; c < L::C; ++c) {
                        f32x4 acc = {g[4 * c], g[4 * c + 1], g[4 * c + 2], g[4 * c + 3]};
#pragma unroll
                        for (int i = 0; i < (VAR == 2 ? 0 : J); ++i) {
                            f32x4 pv;
                            if (i < NLDS) {
                                pv = lds[(i * L::C + c) * 64 + lane];
                            } else {
                                pv = f32x4{s.p[i - NLDS][4 * c], s.p[i - NLDS][4 * c + 1], s.p[i - NLDS][4 * c + 2],
                                           s.p[i - NLDS][4 * c + 3]};
                            }
                            acc.x = fmaf(-w[i], pv.x, acc.x);
                            acc.y = fmaf(-w[i], pv.y, acc.y);
                            acc.z = fmaf(-w[i], pv.z, acc.z);
                            acc.w = fmaf(-w[i], pv.w, acc.w);
                        }
                        acc *= inv;
                        if constexpr (J < NLDS) {
                            lds[(J * L::C + c) * 64 + lane] = acc;
                        } else {
                            s.p[J - NLDS][4 * c] = acc.x;
                            s.p[J - NLDS][4 * c + 1] = acc.y;
                            s.p[J - NLDS][4 * c + 2] = acc.z;
                            s.p[J - NLDS][4 * c + 3] = acc.w;
                        }
                        s.a[4 * c] = fmaf(-t, acc.x, s.a[4 * c]);
                        s.a[4 * c + 1] = fmaf(-t, acc.y, s.a[4 * c + 1]);
                        s.a[4 * c + 2] = fmaf(-t, acc.z, s.a[4 * c + 2]);
                        s.a[4 * c + 3] = fmaf(-t, acc.w, s.a[4 * c + 3]);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        float acc = g[r];
#pragma unroll
                        for (int i = 0; i < (VAR == 2 ? 0 : J); ++i) acc = fmaf(-w[i], s.p[i][r], acc);
                        acc *= inv;
                        s.p[J][r] = acc;
                        s.a[r] = fmaf(-t, acc, s.a[r]);
                    }
                }
            }
        }
        float lr = 0.f;
#pragma unroll
        for (int i = 0; i < J; ++i) lr = writelane_sgpr(lr, w[i], i);  // w[i] came from readlane: already scalar
        s.Lrow[J] = lr;
        s.tv = writelane_f(t, J, s.tv, lane);
        s.rinv = writelane_f(inv, J, s.rinv, lane);
        s.dxv = writelane_i(kk, J, s.dxv, lane);
        s.nsel = J + 1;
        omp_steps<R, KMAX, NLDS, J + 1, VAR, FAST>(s, G, k, lane, lds, unit_diag_rt);
    }
}

// BW = waves per workgroup.  (A persistent variant that prefetches the next signal's alpha0 row into 16 more VGPRs
// was tried and rejected: the loop-carried registers push the 3-waves/SIMD build into scratch, 4.5x slower.)
template <int R, int KMAX, int WAVES_PER_SIMD, int NLDS = 0, int VAR = 0, int BW = 4, bool FAST = false>
__global__ __launch_bounds__(64 * BW, WAVES_PER_SIMD) void bomp_wave_kernel(const float* __restrict__ alpha0,
                                                                           const float* __restrict__ G, int64_t N,
                                                                           int k, int32_t* __restrict__ idx_out,
                                                                           float* __restrict__ coef_out,
                                                                           int32_t* __restrict__ nnz_out,
                                                                           int unit_diag) {
    using L = Lay<R>;
    static_assert(NLDS == 0 || L::V == 4, "LDS-resident vectors need the dwordx4 layout");
    __shared__ f32x4 s_p[NLDS > 0 ? BW * NLDS * L::C * 64 : 1];
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    const int64_t sig = (int64_t)blockIdx.x * BW + wid;
    if (sig >= N) return;

    OmpState<R, KMAX, NLDS> s;
    load_row<R, true>(alpha0 + (VAR == 9 ? (sig & 63) : sig) * L::Kp, lane, s.a);  // VAR 9: alpha0 rows L2-hot (ablation)
#pragma unroll
    for (int j = 0; j < KMAX; ++j) s.Lrow[j] = 0.f;
    s.tv = 0.f;
    s.rinv = 0.f;
    s.m0 = 0.f;
    s.dxv = -1;
    s.nsel = 0;
    omp_steps<R, KMAX, NLDS, 0, VAR, FAST>(s, G, k, lane, s_p + wid * (NLDS * L::C * 64), unit_diag);
    const int nsel = s.nsel;

    // z = L^-T t  (second triangular solve, sparse_coding.py:354), column-oriented over lanes
    // lane i of zv ends up as t_i - sum_{j>i} L[j][i] z_j (row j only touches lanes i < j), so z = zv * rinv at the end
    float zv = s.tv;
#pragma unroll
    for (int i = KMAX - 1; i >= 1; --i) {
        if (i < nsel) {
            const float zi = readlane_f(zv * s.rinv, i);
            zv = fmaf(-zi, s.Lrow[i], zv);
        }
    }
    const float zout = zv * s.rinv;
    if (lane < k) {
        idx_out[sig * k + lane] = (lane < nsel) ? s.dxv : -1;
        coef_out[sig * k + lane] = (lane < nsel) ? zout : 0.f;
    }
    if (lane == 0) nnz_out[sig] = nsel;
}

// ------------------------------------------------------------------------------------------------
// 'thresh' encoder (lyssa/sparse_coding.py:416-425): keep the k largest SIGNED correlations of
// every signal, coefficient = the correlation itself.  One wave per signal, alpha0 row in registers, k rounds of
// (wave max over an order-preserving integer key, owner lane masks its element).  Output slots are in descending
// order of the correlation (the reference's `argsort()[::-1][:k]`).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned ordered_key(float x) {  // monotone map float -> unsigned (no NaN handling)
    const unsigned b = __builtin_bit_cast(unsigned, x);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key_to_float(unsigned kx) {
    const unsigned b = (kx & 0x80000000u) ? (kx & 0x7fffffffu) : ~kx;
    return __builtin_bit_cast(float, b);
}

template <int R>
__global__ __launch_bounds__(256) void thresh_wave_kernel(const float* __restrict__ alpha0, int64_t N, int K, int k,
                                                           int32_t* __restrict__ idx_out, float* __restrict__ coef_out,
                                                           int32_t* __restrict__ nnz_out) {
    using L = Lay<R>;
    const int lane = threadIdx.x & 63;
    const int64_t sig = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (sig >= N) return;
    float a[R];
    load_row<R, true>(alpha0 + sig * L::Kp, lane, a);
    unsigned key[R];
#pragma unroll
    for (int r = 0; r < R; ++r) key[r] = (L::atom(r, lane) < K) ? ordered_key(a[r]) : 0u;  // padded atoms never win
    for (int j = 0; j < k; ++j) {
        unsigned best = key[0];
#pragma unroll
        for (int r = 1; r < R; ++r) best = max(best, key[r]);
        unsigned u = best;
        u = max(u, dpp_u0<0xB1>(u));
        u = max(u, dpp_u0<0x4E>(u));
        u = max(u, dpp_u0<0x124>(u));
        u = max(u, dpp_u0<0x128>(u));
        u = max(u, dpp_u0<0x142>(u));
        u = max(u, dpp_u0<0x143>(u));
        const unsigned m = (unsigned)__builtin_amdgcn_readlane((int)u, 63);
        // lowest atom index among equal keys
        int cand = 0x7fffffff;
#pragma unroll
        for (int r = R - 1; r >= 0; --r) cand = (key[r] == m) ? L::atom(r, lane) : cand;
        const int kk = wave_min_i(cand);
#pragma unroll
        for (int r = 0; r < R; ++r) key[r] = (L::atom(r, lane) == kk) ? 0u : key[r];
        if (lane == 0) {
            idx_out[sig * k + j] = kk;
            coef_out[sig * k + j] = key_to_float(m);
        }
    }
    if (lane == 0) nnz_out[sig] = k;
}

// K > 1024: one workgroup of 256 threads per signal, thread t keeps the keys of atoms t, t+256, .. in registers (R =
// Kp/256 <= 32, i.e. K <= 8192); k rounds of a workgroup-wide maximum over the packed (ordered key, ~atom) pair, so
// that equal correlations resolve to the lowest atom index like `argsort()[::-1][:k]` never does deterministically
// but the single-wave kernel above does.  (sparse_coding.py:416-425 has no limit on K.)
template <int R>
__global__ __launch_bounds__(256) void thresh_block_kernel(const float* __restrict__ alpha0, int Kp, int64_t N, int K,
                                                            int k, int32_t* __restrict__ idx_out,
                                                            float* __restrict__ coef_out, int32_t* __restrict__ nnz_out) {
    __shared__ unsigned long long s_best[4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int64_t sig = blockIdx.x;
    unsigned key[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int a = tid + 256 * r;
        key[r] = (a < K) ? ordered_key(alpha0[sig * Kp + a]) : 0u;  // padded atoms never win
    }
    for (int j = 0; j < k; ++j) {
        unsigned long long best = 0ull;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const unsigned long long pk = ((unsigned long long)key[r] << 32) | (unsigned)(0x7fffffff - (tid + 256 * r));
            best = (pk > best) ? pk : best;
        }
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned long long o = __shfl_xor(best, off);
            best = (o > best) ? o : best;
        }
        if (lane == 0) s_best[wid] = best;
        __syncthreads();
        best = s_best[0];
#pragma unroll
        for (int w = 1; w < 4; ++w) best = (s_best[w] > best) ? s_best[w] : best;
        const int kk = 0x7fffffff - (int)(unsigned)(best & 0xffffffffull);
        const unsigned m = (unsigned)(best >> 32);
#pragma unroll
        for (int r = 0; r < R; ++r) key[r] = (tid + 256 * r == kk) ? 0u : key[r];
        if (tid == 0) {
            idx_out[sig * k + j] = kk;
            coef_out[sig * k + j] = key_to_float(m);
        }
        __syncthreads();
    }
    if (tid == 0) nnz_out[sig] = k;
}

int thresh_from_alpha0(const float* alpha0, int K, int Kp, int k, int64_t N, int32_t* idx, float* coef, int32_t* nnz,
                       hipStream_t stream) {
    if (N <= 0) return LYS_OK;
    if (k < 1 || k > K) {
        set_error("thresh: n_nonzero_coefs must be in [1, K], got %d", k);
        return LYS_EINVAL;
    }
    if (Kp > 1024) {
        const dim3 g((unsigned)N), b(256);
        if (Kp <= 2048) hipLaunchKernelGGL(thresh_block_kernel<8>, g, b, 0, stream, alpha0, Kp, N, K, k, idx, coef, nnz);
        else if (Kp <= 4096) hipLaunchKernelGGL(thresh_block_kernel<16>, g, b, 0, stream, alpha0, Kp, N, K, k, idx, coef, nnz);
        else if (Kp <= 8192) hipLaunchKernelGGL(thresh_block_kernel<32>, g, b, 0, stream, alpha0, Kp, N, K, k, idx, coef, nnz);
        else {
            set_error("thresh: K = %d > 8192 is not implemented", K);
            return LYS_ENOSUP;
        }
        LYS_LAUNCH_CHECK();
        return LYS_OK;
    }
    const dim3 grid((unsigned)((N + 3) / 4)), block(256);
    switch (Kp / 64) {
        case 1: hipLaunchKernelGGL(thresh_wave_kernel<1>, grid, block, 0, stream, alpha0, N, K, k, idx, coef, nnz); break;
        case 2: hipLaunchKernelGGL(thresh_wave_kernel<2>, grid, block, 0, stream, alpha0, N, K, k, idx, coef, nnz); break;
        case 4: hipLaunchKernelGGL(thresh_wave_kernel<4>, grid, block, 0, stream, alpha0, N, K, k, idx, coef, nnz); break;
        case 8: hipLaunchKernelGGL(thresh_wave_kernel<8>, grid, block, 0, stream, alpha0, N, K, k, idx, coef, nnz); break;
        case 16: hipLaunchKernelGGL(thresh_wave_kernel<16>, grid, block, 0, stream, alpha0, N, K, k, idx, coef, nnz); break;
        default: set_error("thresh: unsupported padded K %d", Kp); return LYS_ENOSUP;
    }
    LYS_LAUNCH_CHECK();
    return LYS_OK;
}

// ------------------------------------------------------------------------------------------------
// Workgroup-per-signal register kernel for K > 1024: T threads (T/64 waves) share one signal, thread t holds the
// R = Kp/T correlations  c*4T + 4t + e  (dwordx4-coalesced rows) and the matching slices of the k-1
// orthogonalised vectors in VGPRs.  Same arithmetic as bomp_wave_kernel; the argmax and the w_i = p_i[kk]
// extraction go through LDS with two barriers per step.  K = 4096, k = 20 (config 3): 19*8 = 152 vector VGPRs,
// one workgroup per CU.
// ------------------------------------------------------------------------------------------------
template <int R, int KMAX>
struct BlkState {
    typedef float vec_t __attribute__((ext_vector_type(R)));  // true vectors: element r (run-time) through the VGPR index mode
    vec_t a;
    vec_t p[KMAX - 1];
    int dxv;  // lane i (of every wave): Dx[i], -1 while unset
#ifdef LYS_BLK_STAMPS
    unsigned ph[8];
#endif
};

// Element r (wave-uniform, run-time) of a register vector.  R = 16: the VGPR index mode.  R = 8: LLVM expands the dynamic index
// into compares and selects (8 v_cndmask per read); R = 4: a dynamic index would go through scratch, so a select chain.
// Round 5 packed the vectors into 16-wide ones so that every read is one indexed v_mov (as bomp_wave2.h does, where it made
// the K = 512 kernel 1.5x faster): HERE it measured slower -- 13.97 against 12.85 ms per 131072 signals at K = 4096, k = 20,
// 10.27 against 9.55 ms at K = 2048 -- every read pays its own s_set_gpr_idx_on / off pair (201 pairs in the k = 20 kernel;
// the compiler does not merge them), and a mode switch costs more than the eight selects it replaces.  Not adopted.
template <int R, class V>
__device__ __forceinline__ float blk_elem(const V& v, int r) {
    if constexpr (R == 4) {
        float x = v[3];
        x = (r == 2) ? v[2] : x;
        x = (r == 1) ? v[1] : x;
        x = (r == 0) ? v[0] : x;
        return x;
    } else {
        return v[r];
    }
}

// Development instrumentation of bomp_block_kernel (build with -DLYS_BLK_STAMPS, e.g. lyssandra_amd.build.build(True,
// extra_flags=["-DLYS_BLK_STAMPS"]); tools/blk_timeline.py reads it): with LYS_ABL=<Kp> in the environment every workgroup
// runs all k steps (exits disabled, pivots forced to 1) and accumulates 100 MHz wall-clock ticks per phase of a step in
// registers; start / end / hardware id of every workgroup go to g_blk_tl.  The product build carries none of it.
#ifdef LYS_BLK_STAMPS
__device__ unsigned long long g_blk_ph[8];
__device__ unsigned long long g_blk_tl[4 * 16384];
#define BLK_ABL ((unit_diag >> 8) != 0)
#define BLK_ST(i) do { if (unit_diag >> 8) { const unsigned long long t_ = wall_clock64(); s.ph[i] += (unsigned)(t_ - tl); tl = t_; } } while (0)
#else
#define BLK_ABL false
#define BLK_ST(i) do { } while (0)
#endif
template <int R, int KMAX, int J, int T>
__device__ __forceinline__ void blk_steps(BlkState<R, KMAX>& s, const float* __restrict__ G, int Kp, int k, int tid,
                                          float* s_max, int* s_idx, float* s_w, float* s_akk, float* s_L, float* s_t,
                                          float* s_rinv, int* s_dx, int* s_nsel, float& m0, int unit_diag) {
    if constexpr (J < KMAX) {
        if (J >= k) return;
        const int lane = tid & 63, wid = tid >> 6;
#ifdef LYS_BLK_STAMPS
        unsigned long long tl = wall_clock64();
#endif
        // ---- block-wide argmax |a|, lowest index wins
        float best = fabsf(s.a[0]);
#pragma unroll
        for (int r = 1; r < R; ++r) best = fmaxf(best, fabsf(s.a[r]));
        const float mw = wave_max_f(best);
        int cand = 0x7fffffff;
#pragma unroll
        for (int r = R - 1; r >= 0; --r) cand = (fabsf(s.a[r]) == mw) ? ((r >> 2) * (4 * T) + tid * 4 + (r & 3)) : cand;
        const int cw = wave_min_i(cand);
        float* smx = s_max + (J & 1) * (T / 64);
        int* six = s_idx + (J & 1) * (T / 64);
        if (lane == 0) {
            smx[wid] = mw;
            six[wid] = cw;
        }
        BLK_ST(0);
        __syncthreads();
        BLK_ST(1);
        // the T/64 per-wave candidates, one per lane (replicated over the wave), reduced like the per-wave search: maximum,
        // then the lowest atom among its holders.  (A scalar loop over the candidates compiles into T/64 serialised
        // [ds_read -> s_waitcnt -> v_cmp -> v_readfirstlane -> branch] round trips: 0.47 of the 1.6 us of a step at k = 20.)
        float m;
        int kk;
        if constexpr (T / 64 >= 4) {
            const float vq = smx[lane & (T / 64 - 1)];
            const int cq = six[lane & (T / 64 - 1)];
            m = wave_max_f(vq);
            kk = wave_min_i((vq == m) ? cq : 0x7fffffff);
        } else {  // two waves: the scalar compare is cheaper than two wave reductions
            m = smx[0];
            kk = six[0];
#pragma unroll
            for (int q = 1; q < T / 64; ++q) {
                const float v = smx[q];
                const int c = six[q];
                const bool better = (v > m) || (v == m && c < kk);
                m = better ? v : m;
                kk = better ? c : kk;
            }
        }
        const bool abl = BLK_ABL;
        if (!abl && (!(m == m) || kk == 0x7fffffff)) return;  // NaN correlations
        if (abl) kk &= (Kp - 1);
        if constexpr (J == 0) {
            m0 = m;
        } else {
            if (!abl && m < NOISE_REL * m0) return;
        }
        // re-selection => stop (sparse_coding.py:323-325).  Lane i of every wave holds Dx[i]: one compare + ballot (a loop of
        // J dependent LDS reads with an exit each was the longest phase of a step: 0.47 of 1.6 us at k = 20)
        if (!abl && __ballot(s.dxv == kk) != 0ull) return;
        s.dxv = (lane == J) ? kk : s.dxv;
        BLK_ST(2);
        const bool more = (J + 1 < KMAX) && (J + 1 < k);
        float g[R];
        if (!more) {
#pragma unroll
            for (int r = 0; r < R; ++r) asm volatile("" : "=v"(g[r]));  // never read: no zero-fill
        }
        if (more) {
#pragma unroll
            for (int c = 0; c < R / 4; ++c) {
                const f32x4 t4 = *(reinterpret_cast<const f32x4*>(G + (int64_t)kk * Kp) + c * T + tid);
                g[4 * c] = t4.x;
                g[4 * c + 1] = t4.y;
                g[4 * c + 2] = t4.z;
                g[4 * c + 3] = t4.w;
            }
        }
        // ---- owner publishes a[kk] and w_i = p_i[kk]
        // the owner's WAVE reads element r of every vector (r is wave-uniform: one indexed v_mov per vector), the owner lane
        // stores them (before: a branch tree over the register slots on one thread while 511 waited at the barrier)
        {
            const int owner = (kk % (4 * T)) >> 2, rown = (kk / (4 * T)) * 4 + (kk & 3);
            if (wid == (owner >> 6)) {  // wave-uniform
                const float va = blk_elem<R>(s.a, rown);
                float tmp[J > 0 ? J : 1];
#pragma unroll
                for (int i = 0; i < J; ++i) tmp[i] = blk_elem<R>(s.p[i], rown);
                if (tid == owner) {
                    *s_akk = va;
#pragma unroll
                    for (int i = 0; i < J; ++i) s_w[i] = tmp[i];
                }
            }
        }
        BLK_ST(3);
        __syncthreads();
        BLK_ST(4);
        float w[KMAX];
        const float gkk = unit_diag ? 1.f : G[(int64_t)kk * Kp + kk];
        float vs = gkk;
#pragma unroll
        for (int i = 0; i < J; ++i) {
            w[i] = s_w[i];
            vs = fmaf(-w[i], w[i], vs);
        }
        if (!abl && (J > 0 || !unit_diag) && vs < EPS32_F * gkk) return;
        if (abl) vs = 1.f + 1e-30f * vs;
        const float inv = __builtin_amdgcn_rsqf(vs);  // 1 ulp, like the single-wave kernel
        const float t = abl ? 1e-3f * inv : (*s_akk) * inv;
        BLK_ST(5);
        if constexpr (J + 1 < KMAX) {
            if (more) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    float acc = g[r];
#pragma unroll
                    for (int i = 0; i < J; ++i) acc = fmaf(-w[i], s.p[i][r], acc);
                    acc *= inv;
                    s.p[J][r] = acc;
                    s.a[r] = fmaf(-t, acc, s.a[r]);
                }
            }
        }
        if (tid == 0) {
#pragma unroll
            for (int i = 0; i < J; ++i) s_L[J * KMAX + i] = w[i];
            s_t[J] = t;
            s_rinv[J] = inv;
            s_dx[J] = kk;
            *s_nsel = J + 1;
        }
#ifdef LYS_BLK_STAMPS
        { float dep = s.a[0]; asm volatile("" : "+v"(dep)); s.a[0] = dep; }
#endif
        BLK_ST(6);
        // s_dx[J] is read by every thread in the next step only after that step's first barrier
        blk_steps<R, KMAX, J + 1, T>(s, G, Kp, k, tid, s_max, s_idx, s_w, s_akk, s_L, s_t, s_rinv, s_dx, s_nsel, m0,
                                  unit_diag);
    }
}

template <int R, int KMAX, int W, int T = 512>
__global__ __launch_bounds__(T, W) void bomp_block_kernel(const float* __restrict__ alpha0,
                                                            const float* __restrict__ G, int64_t N, int k,
                                                            int32_t* __restrict__ idx_out,
                                                            float* __restrict__ coef_out,
                                                            int32_t* __restrict__ nnz_out, int unit_diag) {
    constexpr int Kp = T * R;
    __shared__ float s_max[16];
    __shared__ int s_idx[16];
    __shared__ float s_w[KMAX];
    __shared__ float s_akk;
    __shared__ float s_L[KMAX * KMAX];
    __shared__ float s_t[KMAX], s_rinv[KMAX];
    __shared__ int s_dx[KMAX];
    __shared__ int s_nsel;
    const int tid = threadIdx.x;
    const int64_t sig = blockIdx.x;
#ifdef LYS_BLK_STAMPS
    const unsigned long long t_wg0 = wall_clock64();
#endif
    if (tid == 0) s_nsel = 0;
    if (tid < KMAX) s_dx[tid] = -1;
    BlkState<R, KMAX> s;
#ifdef LYS_BLK_STAMPS
    for (int i = 0; i < 8; ++i) s.ph[i] = 0;
#endif
    s.dxv = -1;
#pragma unroll
    for (int c = 0; c < R / 4; ++c) {
        const f32x4 t4 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(alpha0 + sig * Kp) + c * T + tid);
        s.a[4 * c] = t4.x;
        s.a[4 * c + 1] = t4.y;
        s.a[4 * c + 2] = t4.z;
        s.a[4 * c + 3] = t4.w;
    }
    __syncthreads();
    float m0 = 0.f;
#ifdef LYS_BLK_STAMPS
    const unsigned long long t_wg1 = wall_clock64();
#endif
    blk_steps<R, KMAX, 0, T>(s, G, Kp, k, tid, s_max, s_idx, s_w, &s_akk, s_L, s_t, s_rinv, s_dx, &s_nsel, m0, unit_diag);
    __syncthreads();
#ifdef LYS_BLK_STAMPS
    const unsigned long long t_wg2 = wall_clock64();
#endif
    const int nsel = s_nsel;
    if (tid < 64) {
        // z = L^-T t (sparse_coding.py:354) on wave 0, column by column: lane q holds t_q, every solved z_i is broadcast and
        // subtracted from the lanes below it (row i of L is contiguous in LDS).  The former single-thread double loop --
        // k^2/2 dependent LDS round trips -- took a quarter of the kernel at k = 20.
        float tq = (tid < nsel) ? s_t[tid] : 0.f;
        const float rq = (tid < nsel) ? s_rinv[tid] : 0.f;
        float zq = 0.f;
        for (int i = nsel - 1; i >= 0; --i) {
            const float zi = readlane_f(tq * rq, i);
            zq = (tid == i) ? zi : zq;
            const float Liq = (tid < i) ? s_L[i * KMAX + tid] : 0.f;
            tq = fmaf(-Liq, zi, tq);
        }
        if (tid == 0) nnz_out[sig] = nsel;
        if (tid < k) {
            idx_out[sig * k + tid] = (tid < nsel) ? s_dx[tid] : -1;
            coef_out[sig * k + tid] = (tid < nsel) ? zq : 0.f;
        }
    }
#ifdef LYS_BLK_STAMPS
    if ((unit_diag >> 8) && tid == 0) {
        const unsigned long long t_wg3 = wall_clock64();
        for (int i = 0; i < 7; ++i) atomicAdd(&g_blk_ph[i], (unsigned long long)s.ph[i]);
        atomicAdd(&g_blk_ph[7], t_wg3 - t_wg0);
        if (blockIdx.x < 16384) {
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            g_blk_tl[4 * blockIdx.x] = t_wg0;
            g_blk_tl[4 * blockIdx.x + 1] = t_wg3;
            g_blk_tl[4 * blockIdx.x + 2] = ((unsigned long long)xcc << 32) | hw;
            g_blk_tl[4 * blockIdx.x + 3] = t_wg2 - t_wg1;
        }
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// Generic kernel: any Kp (multiple of 64) and any k <= 64.  One 256-thread workgroup per signal, the
// correlations and the p-vectors live in a per-workgroup global scratch slab (L2-resident), L in LDS.
// Same arithmetic, used for K > 1024 (config 3: K=4096, k=20) and for k beyond the register kernels.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bomp_generic_kernel(const float* __restrict__ alpha0,
                                                            const float* __restrict__ G, int Kp, int64_t N, int k,
                                                            float* __restrict__ scratch,  // [grid][(k+1)*Kp]
                                                            int32_t* __restrict__ idx_out,
                                                            float* __restrict__ coef_out,
                                                            int32_t* __restrict__ nnz_out, int unit_diag,
                                                            const float* __restrict__ xnorm2 = nullptr,
                                                            float tol2 = 0.f) {
    // xnorm2 != nullptr: error-constrained mode (`_omp` with tol and no n_nonzero_coefs, sparse_coding.py:27-31): select
    // while ||r|| >= tol, with ||r||^2 = ||x||^2 - sum_j t_j^2 (t = the forward-substituted coefficients; exact for the
    // least-squares fit on the support, fp32 cancellation limits it to tol >~ 1e-3 ||x||)
    __shared__ float s_val[4];
    __shared__ int s_idx[4];
    __shared__ float s_res2;
    __shared__ float s_w[64];
    __shared__ float s_L[64 * 64];
    __shared__ float s_t[64], s_rinv[64], s_z[64];
    __shared__ int s_dx[64];
    __shared__ float s_akk, s_m0;
    __shared__ int s_kk, s_stop;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    float* a = scratch + (int64_t)blockIdx.x * (int64_t)(k + 1) * Kp;
    float* P = a + Kp;  // P[i][Kp]

    for (int64_t sig = blockIdx.x; sig < N; sig += gridDim.x) {
        for (int x = tid; x < Kp; x += 256) a[x] = alpha0[sig * Kp + x];
        if (tid == 0) {
            s_stop = 0;
            s_res2 = xnorm2 ? xnorm2[sig] : 0.f;
        }
        __syncthreads();
        int nsel = 0;
        for (int j = 0; j < k; ++j) {
            // ---- argmax |a| with lowest-index tie-break
            float best = -1.f;
            int bi = 0x7fffffff;
            for (int x = tid; x < Kp; x += 256) {
                const float v = fabsf(a[x]);
                if (v > best) {
                    best = v;
                    bi = x;
                }
            }
            const float m = wave_max_f(best);
            const int ci = wave_min_i((best == m) ? bi : 0x7fffffff);
            if (lane == 0) {
                s_val[wid] = m;
                s_idx[wid] = ci;
            }
            __syncthreads();
            if (tid == 0) {
                float mm = s_val[0];
                int kk = s_idx[0];
                for (int q = 1; q < 4; ++q) {
                    if (s_val[q] > mm || (s_val[q] == mm && s_idx[q] < kk)) {
                        mm = s_val[q];
                        kk = s_idx[q];
                    }
                }
                bool stop = !(mm == mm) || kk == 0x7fffffff;
                if (xnorm2 && !(s_res2 >= tol2)) stop = true;  // ||r|| < tol (also before the first selection)
                if (j == 0) s_m0 = mm;
                else if (mm < NOISE_REL * s_m0) stop = true;  // fp32 noise floor, see NOISE_REL
                for (int i = 0; i < j && !stop; ++i) stop = (s_dx[i] == kk);
                s_kk = kk;
                s_stop = stop ? 1 : 0;
                if (!stop) s_akk = a[kk];
            }
            __syncthreads();
            if (s_stop) break;
            const int kk = s_kk;
            // ---- w_i = p_i[kk]
            if (tid < j) s_w[tid] = P[(int64_t)tid * Kp + kk];
            __syncthreads();
            if (tid == 0) {
                const float gkk = unit_diag ? 1.f : G[(int64_t)kk * Kp + kk];
                float vs = gkk;
                for (int i = 0; i < j; ++i) vs = fmaf(-s_w[i], s_w[i], vs);
                if ((j > 0 || !unit_diag) && vs < EPS32_F * gkk) {
                    s_stop = 1;
                } else {
                    const float rho = sqrtf(vs);
                    const float inv = 1.f / rho;
                    s_rinv[j] = inv;
                    s_t[j] = s_akk * inv;
                    s_res2 = fmaf(-s_t[j], s_t[j], s_res2);
                    s_dx[j] = kk;
                    for (int i = 0; i < j; ++i) s_L[j * 64 + i] = s_w[i];
                }
            }
            __syncthreads();
            if (s_stop) break;
            const float inv = s_rinv[j], t = s_t[j];
            float* pj = P + (int64_t)j * Kp;
            const float* grow = G + (int64_t)kk * Kp;
            for (int x = tid; x < Kp; x += 256) {
                float acc = grow[x];
                for (int i = 0; i < j; ++i) acc = fmaf(-s_w[i], P[(int64_t)i * Kp + x], acc);
                acc *= inv;
                pj[x] = acc;
                a[x] = fmaf(-t, acc, a[x]);
            }
            nsel = j + 1;
            __syncthreads();
        }
        __syncthreads();
        if (tid == 0) {
            for (int i = nsel - 1; i >= 0; --i) {
                float zi = s_t[i];
                for (int mI = i + 1; mI < nsel; ++mI) zi = fmaf(-s_L[mI * 64 + i], s_z[mI], zi);
                s_z[i] = zi * s_rinv[i];
            }
            nnz_out[sig] = nsel;
        }
        __syncthreads();
        if (tid < k) {
            idx_out[sig * k + tid] = (tid < nsel) ? s_dx[tid] : -1;
            coef_out[sig * k + tid] = (tid < nsel) ? s_z[tid] : 0.f;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
template <int R, int KMAX>
static int launch_wave(const float* alpha0, const float* G, int64_t N, int k, int32_t* idx, float* coef, int32_t* nnz,
                       hipStream_t stream, int unit_diag) {
    constexpr int regs = (KMAX + 2) * R + KMAX + 40;
    constexpr int W = (regs > 256) ? 1 : (regs > 168) ? 2 : (regs > 128) ? 3 : (regs > 96) ? 4 : 5;
    const int64_t blocks = (N + 3) / 4;
    if (blocks > 0x7fffffffLL) {
        set_error("bomp: too many signals per launch (%lld)", (long long)N);
        return LYS_ENOSUP;
    }
    hipLaunchKernelGGL((bomp_wave_kernel<R, KMAX, W>), dim3((unsigned)blocks), dim3(256), 0, stream, alpha0, G, N, k, idx,
                       coef, nnz, unit_diag);
    LYS_LAUNCH_CHECK();
    return LYS_OK;
}

template <int R>
static int dispatch_k(const float* alpha0, const float* G, int64_t N, int k, int32_t* idx, float* coef, int32_t* nnz,
                      hipStream_t stream, int unit_diag) {
    // (R >= 4, k <= 10) is served by the second-generation kernel (bomp_wave.hip) before this dispatch is reached
    // (round 5: also R = 8, k <= 5 -- 0.72 against 0.91 ms per 2^20 signals since its vectors are read through the index mode)
    if constexpr (R < 4) {
        if (k <= 5) return launch_wave<R, 5>(alpha0, G, N, k, idx, coef, nnz, stream, unit_diag);
    }
    if constexpr (R < 4) {
        if (k <= 10) return launch_wave<R, 10>(alpha0, G, N, k, idx, coef, nnz, stream, unit_diag);
    }
    if constexpr (R <= 8) {
        if (k <= 20) return launch_wave<R, 20>(alpha0, G, N, k, idx, coef, nnz, stream, unit_diag);
    }
    if constexpr (R <= 4) {
        if (k <= 32) return launch_wave<R, 32>(alpha0, G, N, k, idx, coef, nnz, stream, unit_diag);
    }
    return 1;  // not covered by a register kernel
}

// A/B reference for tools/omp_ab2.py: variant 0 = the FIRST-generation headline launch (round 1-2 product kernel: 2
// vectors in LDS, 3 waves/SIMD, compile-time k and unit diagonal).  The round-1 timing ablations (cache-hot Gram rows,
// no orthogonalisation FMAs, ...; results in profiles/r01_ablations_and_shapes.txt) were removed with their 15
// instantiations in round 3.
// out[0 .. 8): per-phase tick sums over all workgroups since the library was loaded; out[8 ..): 4 words per workgroup of
// the last launch (start, end, XCC_ID << 32 | HW_ID, ticks inside the steps).  LYS_ENOSUP unless built with LYS_BLK_STAMPS.
int bomp_debug_timeline(unsigned long long* out) {
#ifdef LYS_BLK_STAMPS
    LYS_CHECK_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_blk_ph), sizeof(unsigned long long) * 8));
    LYS_CHECK_HIP(hipMemcpyFromSymbol(out + 8, HIP_SYMBOL(g_blk_tl), sizeof(unsigned long long) * 4 * 16384));
    return LYS_OK;
#else
    (void)out;
    set_error("bomp_debug_timeline: library built without -DLYS_BLK_STAMPS");
    return LYS_ENOSUP;
#endif
}

int bomp_debug_variant(const float* alpha0, const float* G, int64_t N, int k, int32_t* idx, float* coef, int32_t* nnz,
                       int variant, int lds_bytes, hipStream_t stream) {
    const dim3 grid((unsigned)((N + 3) / 4)), block(256);
    if (variant != 0 || k != 10) {
        set_error("debug variant %d (k = %d): only variant 0 with k = 10 is built", variant, k);
        return LYS_EINVAL;
    }
    hipLaunchKernelGGL((bomp_wave_kernel<16, 10, 3, 2, 0, 4, true>), grid, block, lds_bytes, stream, alpha0, G, N, k, idx,
                       coef, nnz, 1);
    LYS_LAUNCH_CHECK();
    return LYS_OK;
}

template <int R, int KMAX, int W, int T = 512>
static int launch_block(const float* alpha0, const float* G, int64_t N, int k, int32_t* idx, float* coef, int32_t* nnz,
                        hipStream_t stream, int unit_diag) {
    if (N > 0x7fffffffLL) {
        set_error("bomp: too many signals per launch (%lld)", (long long)N);
        return LYS_ENOSUP;
    }
#ifdef LYS_BLK_STAMPS
    if (getenv("LYS_ABL")) unit_diag |= atoi(getenv("LYS_ABL")) << 8;
#endif
    hipLaunchKernelGGL((bomp_block_kernel<R, KMAX, W, T>), dim3((unsigned)N), dim3(T), 0, stream, alpha0, G, N, k, idx,
                       coef, nnz, unit_diag);
    LYS_LAUNCH_CHECK();
    return LYS_OK;
}

static bool bomp_has_block_kernel(int Kp, int k) {
    if (Kp == 2048 || Kp == 4096) return k <= 20;
    if (Kp == 8192) return k <= 10;
    return false;
}

// true when (K,k) is served by a register-resident kernel (no global scratch needed)
bool bomp_has_wave_kernel(int Kp, int k) {
    if (Kp > 1024) return bomp_has_block_kernel(Kp, k);
    const int R = Kp / 64;
    if (k <= 10) return true;
    if (k <= 20) return R <= 8;
    if (k <= 32) return R <= 4;
    return false;
}

size_t bomp_generic_scratch_bytes(int Kp, int k) {
    const int grid = num_cus() * 4;
    return (size_t)grid * (size_t)(k + 1) * (size_t)Kp * sizeof(float);
}

int bomp_from_alpha0(const float* alpha0, const float* G, int Kp, int k, int64_t N, int32_t* idx, float* coef,
                     int32_t* nnz, float* generic_scratch, hipStream_t stream, int unit_diag) {
    if (N <= 0) return LYS_OK;
    if (k < 1 || k > 64) {
        set_error("bomp: n_nonzero_coefs must be in [1,64], got %d", k);
        return LYS_ENOSUP;
    }
    int rc = 1;
    if (Kp > 1024 && bomp_has_block_kernel(Kp, k)) {
        // k <= 10: 16 correlations per thread (2 / 4 / 8 waves per signal), like the single-wave kernel (K = 2048:
        // 41 -> 62 M patches/s against the 512-thread split); k <= 20 needs the thinner split to hold 19 vectors.
        // (Letting the winning wave resolve the argmax with readlanes and publish w itself was tried: slower, the other
        // waves wait at the barrier for that serial chain.)
        if (Kp == 2048) return (k <= 10) ? launch_block<16, 10, 2, 128>(alpha0, G, N, k, idx, coef, nnz, stream, unit_diag)
                                         : launch_block<4, 20, 3>(alpha0, G, N, k, idx, coef, nnz, stream, unit_diag);
        if (Kp == 4096) return (k <= 10) ? launch_block<16, 10, 2, 256>(alpha0, G, N, k, idx, coef, nnz, stream, unit_diag)
                                         : launch_block<8, 20, 2>(alpha0, G, N, k, idx, coef, nnz, stream, unit_diag);
        return launch_block<16, 10, 2>(alpha0, G, N, k, idx, coef, nnz, stream, unit_diag);
    }
    if (Kp <= 1024 && Kp >= 256 && k <= 10) {
        rc = bomp_wave2_launch(Kp, alpha0, G, N, k, idx, coef, nnz, stream, unit_diag);
        if (rc != 1) return rc;
    }
    if (bomp_has_wave_kernel(Kp, k)) {
        switch (Kp / 64) {
            case 1: rc = dispatch_k<1>(alpha0, G, N, k, idx, coef, nnz, stream, unit_diag); break;
            case 2: rc = dispatch_k<2>(alpha0, G, N, k, idx, coef, nnz, stream, unit_diag); break;
            case 4: rc = dispatch_k<4>(alpha0, G, N, k, idx, coef, nnz, stream, unit_diag); break;
            case 8: rc = dispatch_k<8>(alpha0, G, N, k, idx, coef, nnz, stream, unit_diag); break;
            case 16: rc = dispatch_k<16>(alpha0, G, N, k, idx, coef, nnz, stream, unit_diag); break;
            default: rc = 1;
        }
    }
    if (rc != 1) return rc;
    if (generic_scratch == nullptr) {
        set_error("bomp: generic kernel needs scratch (K=%d, k=%d)", Kp, k);
        return LYS_EWORKSPACE;
    }
    const int grid = (int)((N < (int64_t)num_cus() * 4) ? N : (int64_t)num_cus() * 4);
    hipLaunchKernelGGL(bomp_generic_kernel, dim3(grid), dim3(256), 0, stream, alpha0, G, Kp, N, k, generic_scratch,
                       idx, coef, nnz, unit_diag, (const float*)nullptr, 0.f);
    LYS_LAUNCH_CHECK();
    return LYS_OK;
}

// ||x||^2 of every signal-major row (one wave per row)
__global__ __launch_bounds__(256) void row_norm2_kernel(const float* __restrict__ X, int64_t ldx, int n, int64_t N,
                                                        float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t s = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= N) return;
    float ss = 0.f;
    for (int f = lane; f < n; f += 64) ss = fmaf(X[s * ldx + f], X[s * ldx + f], ss);
    ss = wave_sum_f(ss);
    if (lane == 0) out[s] = ss;
}

// error-constrained 'omp' on one alpha0 tile: generic kernel, true Gram diagonal, stop when ||r|| < tol
int omp_tol_from_alpha0(const float* alpha0, const float* G, int Kp, int kcap, int64_t N, const float* X, int64_t ldx,
                        int n, float tol, float* xnorm2, int32_t* idx, float* coef, int32_t* nnz, float* generic_scratch,
                        hipStream_t stream) {
    if (N <= 0) return LYS_OK;
    if (kcap < 1 || kcap > 64) {
        set_error("omp(tol): at most 64 atoms per signal, got kcap = %d", kcap);
        return LYS_ENOSUP;
    }
    hipLaunchKernelGGL(row_norm2_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, stream, X, ldx, n, N, xnorm2);
    LYS_LAUNCH_CHECK();
    const int grid = (int)((N < (int64_t)num_cus() * 4) ? N : (int64_t)num_cus() * 4);
    hipLaunchKernelGGL(bomp_generic_kernel, dim3(grid), dim3(256), 0, stream, alpha0, G, Kp, N, kcap, generic_scratch,
                       idx, coef, nnz, 0, (const float*)xnorm2, tol * tol);
    LYS_LAUNCH_CHECK();
    return LYS_OK;
}

}  // namespace lys
