// Second-generation single-wave Batch-OMP kernel for gfx950 (restates lyssa/sparse_coding.py:302-367, `batch_omp`).
//
// Same progressive (orthogonalised) algebra as the first kernel (header of bomp.hip); what changed is the SHAPE of the
// per-step instruction chain, because the first kernel turned out to be bound by the LENGTH of one wave's serial chain
// (argmax -> owner lookup -> extraction -> pivot -> Gram row -> FMAs: ~1800 cycles per step, of which the FMAs are ~250),
// not by VALU throughput -- removing 80 % of its FMAs bought 10 %, a third wave per SIMD 22 %:
//
//   * of the k-1 orthogonalised vectors the NLDS oldest live in LDS, the rest in VGPRs, and (NV = 1) the LAST one is
//     never stored: its only later use is the single element p_{k-2}[kk] in the last selection, which is
//         (G[kk_{k-2}][kk] - sum_l L[k-2][l] w_l) / rho_{k-2}
//     -- one scalar load and k-2 scalar FMAs from values the step already has.  16 VGPRs less.
//     (A general form that carries several virtual vectors as coefficients over re-read Gram rows was built and
//     measured in round 3: correct, but it only pays with a fourth wave per SIMD, which LDS capacity rules out.)
//   * ONE data-dependent exit per step instead of three: the noise-floor, re-selection and pivot tests are folded into
//     a flag that is tested after the vector update (the update of a stopping signal is wasted and never read), so the
//     whole step is one basic block and the Gram-row load is issued as soon as kk is known, before the extraction chain.
//   * ties between lanes resolve through per-group lane masks (v_cmp -> s_ff1), exact lowest-index semantics without the
//     separate slow path; L (strictly lower triangle, k(k-1)/2 entries) is packed into the lanes of one VGPR.
#pragma once
#include "common.h"

namespace lys {
namespace w2 {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr float NOISE_REL = 4e-6f;  // see bomp.hip

// Element r = c*4 + e of lane l holds atom c*256 + l*4 + e (one coalesced dwordx4 per chunk c).
template <int R>
struct Lay {
    static_assert(R % 4 == 0, "dwordx4 layout");
    static constexpr int C = R / 4;
    static constexpr int Kp = 64 * R;
};

__device__ __forceinline__ float sgpr_f(float x) {  // force a wave-uniform value into an SGPR
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x)));
}

constexpr int tri(int j) { return j * (j - 1) / 2; }

template <int R, int KMAX, int NLDS, int NV>
struct State {
    static constexpr int NS = KMAX - 1 - NV;  // stored vectors p_0 .. p_{NS-1}
    static constexpr int NREG = NS - NLDS;    // of them in VGPRs
    static_assert(NS >= NLDS && NS >= 0, "more LDS vectors than stored vectors");
    // per-wave LDS scalar area: L[j][i] at j*KMAX+i, t_j at SC_T+j, 1/rho_j at SC_I+j (floats)
    static constexpr int SC_T = KMAX * KMAX, SC_I = SC_T + KMAX, SC_FLOATS = SC_I + KMAX;
    // The register-resident vectors -- logical vector 0 = the current correlations a, 1 + i = p_{NLDS+i} -- are PACKED into
    // 16-wide register vectors, 16 / R of them each (round 5).  Why: element `rown` of every vector is read with a run-time
    // (wave-uniform) index in every step, and LLVM serves a dynamic index into a vector of <= 8 floats with a chain of
    // compares and selects (8 v_cndmask + 8 s_cmp per read; the K = 512 kernel spent 961 of its 1169 VALU instructions
    // outside the FMAs, more than the K = 1024 kernel in total), into a 16-wide one with the VGPR index mode (one v_mov).
    static constexpr int PACK = 16 / R;
    static constexpr int NLOG = 1 + (NREG > 0 ? NREG : 0);
    static constexpr int N16 = (NLOG + PACK - 1) / PACK;
    typedef float v16_t __attribute__((ext_vector_type(16)));
    v16_t vec[N16];
    __device__ __forceinline__ float get(int j, int r) const { return vec[j / PACK][(j % PACK) * R + r]; }
    __device__ __forceinline__ void set(int j, int r, float x) { vec[j / PACK][(j % PACK) * R + r] = x; }
    // element r (run-time, wave-uniform) of logical vector j.  R = 4: the caller makes r opaque -- knowing r < 4 the compiler
    // served the vectors from scratch (384 B per lane at k = 10) instead of the index mode; against the three-select chain
    // the index mode measured 0.61 -> 0.59 ms (K = 256, k = 5) and 1.33 -> 1.19 ms (k = 10) per 2^20 signals.
    __device__ __forceinline__ float dyn(int j, int r) const {
        return vec[j / PACK][(j % PACK) * R + r];
    }
    int dxv;                       // lane j = Dx[j]
    unsigned m0;                   // bits of NOISE_REL * max|alpha0|
    unsigned laneoff;              // LDS byte address of this lane's dwordx4 slot of (vector 0, chunk 0)
    int nsel;
    static_assert(NV == 0 || NV == 1, "at most the last vector is virtual");
    // NV = 1: what the last selection needs of the unstored vector p_{KMAX-2}
    int kkv;
    float invv;
    float wv[KMAX];
    // STAMP builds only: s_memtime at the last stamp, cycles per phase (argmax+reduce | owner lookup | extraction+pivot |
    // vector update+commit)
    unsigned long long tlast;
    unsigned cyc[4];
};

// Diagnostic timestamps (STAMP builds): s_memtime ordered against the computation through a register dependency.
__device__ __forceinline__ unsigned long long stamp_s(unsigned& dep) {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "+s"(dep));
    return t;
}
__device__ __forceinline__ unsigned long long stamp_v(float& dep) {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "+v"(dep));
    return t;
}
#define W2_STAMP_S(ph, dep)                                            \
    if constexpr (STAMP) {                                             \
        const unsigned long long t_ = stamp_s(dep);                    \
        s.cyc[ph] += (unsigned)(t_ - s.tlast);                         \
        s.tlast = t_;                                                  \
    }
#define W2_STAMP_V(ph, dep)                                            \
    if constexpr (STAMP) {                                             \
        const unsigned long long t_ = stamp_v(dep);                    \
        s.cyc[ph] += (unsigned)(t_ - s.tlast);                         \
        s.tlast = t_;                                                  \
    }

// Element r (wave-uniform, run-time) of a register vector.  R >= 8: the VGPR index mode (one indexed v_mov).  R = 4: LLVM
// serves a dynamic index into a 4-element vector from SCRATCH (160-256 B per lane; K = 256 ran 2-3x slower than the first
// kernel), so a three-select chain instead.
template <int R, class V>
__device__ __forceinline__ float elem_dyn(const V& v, int r) {
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

// argmax |a| over the wave, first (lowest atom index) maximum wins like np.argmax (sparse_coding.py:322).
//
// Every VALU instruction of a wave64 costs the SIMD ~4 cycles whatever it does (PMC: SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU
// = 1.05 quad-cycles), so a v_cndmask is as expensive as a v_pk_fma_f32 that does 128 FMAs; the SALU is a separate pipe
// with headroom, but a chain of VALU -> SGPR -> SALU round trips is the slowest thing a wave can do (the first kernel's
// readlane / s_cmp / branch-tree lookup: 45 instructions, 700 cycles per step).  Hence: the VALU produces LANE MASKS
// (one v_cmp each), the SALU only tests bits of them:
//   E_c   = lanes whose group c (registers 4c..4c+3) holds the lane's own maximum      (independent of the wave maximum:
//           issued beside the DPP reduction)
//   bal   = lanes that hold the wave maximum;  first c with E_c & bal != 0 -> chunk, s_ff1 -> owner lane
//   F_e   = lanes whose element e of the owner's group equals the lane maximum (group read through the index mode)
// Atom order is (chunk c, lane, element e), so "lowest chunk, then lowest lane, then lowest element" is exactly
// np.argmax's first maximum, also when several lanes tie (duplicate atoms, zero signals): no separate slow path.
template <int R, bool STAMP = false, class AV>
__device__ __forceinline__ bool wave_argmax(const AV& a, int& kk, int& Lown, int& rown, unsigned& mbits_out,
                                            unsigned long long* tmid = nullptr) {
    constexpr int NG = R / 4;
    float m4[NG];
#pragma unroll
    for (int c = 0; c < NG; ++c) {
        float t3;
        asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(t3) : "v"(a[4 * c]), "v"(a[4 * c + 1]), "v"(a[4 * c + 2]));
        asm("v_max_f32_e64 %0, %1, |%2|" : "=v"(m4[c]) : "v"(t3), "v"(a[4 * c + 3]));
    }
    float best = m4[0];
    if constexpr (NG == 4) {
        float t3;
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(t3) : "v"(m4[0]), "v"(m4[1]), "v"(m4[2]));
        asm("v_max_f32_e32 %0, %1, %2" : "=v"(best) : "v"(t3), "v"(m4[3]));
    } else if constexpr (NG == 2) {
        asm("v_max_f32_e32 %0, %1, %2" : "=v"(best) : "v"(m4[0]), "v"(m4[1]));
    }
    unsigned long long E[NG > 1 ? NG - 1 : 1];
#pragma unroll
    for (int c = 0; c < NG - 1; ++c) E[c] = __ballot(m4[c] == best);
    const float m = wave_max_f(best);
    unsigned mbits = __builtin_bit_cast(unsigned, m);
    if constexpr (STAMP) *tmid = stamp_s(mbits);
    mbits_out = mbits;
    const unsigned long long bal = __ballot(best == m);
    if (bal == 0ull) return false;  // NaN correlations: nothing sensible to select
    // the first chunk that holds the maximum in any lane, then its lowest lane
    unsigned long long T = bal;
    int csel = NG - 1;
#pragma unroll
    for (int c = NG - 2; c >= 0; --c) {
        const unsigned long long tc = E[c] & bal;
        const bool ne = tc != 0ull;
        T = ne ? tc : T;
        csel = ne ? c : csel;
    }
    const int Lo = __builtin_ctzll(T);
    int q = 3;
    // the owner's group, read with a run-time (wave-uniform) register index.  (Round 5 tried the lane-local form -- every lane
    // selects ITS OWN first group holding its maximum with the E masks, 9 v_cndmask beside the DPP reduction, no indexed read
    // and one index-mode region per step instead of two: K = 1024, k = 10 2.93 against 2.93 ms; not kept.)
    float x[3];
#pragma unroll
    for (int e = 0; e < 3; ++e) x[e] = a[csel * 4 + e];
#pragma unroll
    for (int e = 2; e >= 0; --e) {
        const unsigned long long fe = __ballot(fabsf(x[e]) == best);
        const bool hit = ((fe >> Lo) & 1ull) != 0ull;  // wave-uniform
        q = hit ? e : q;
    }
    // (the signed value a[kk] is read by the caller, in the same index-mode region as the elements of the stored vectors)
    Lown = Lo;
    rown = csel * 4 + q;
    kk = csel * 256 + Lo * 4 + q;
    return true;
}

// one Gram / alpha0 row chunk per dwordx4
template <int R, bool STREAM>
__device__ __forceinline__ void load_row4(const float* __restrict__ row, int lane, f32x4 (&v)[R / 4]) {
#pragma unroll
    for (int c = 0; c < R / 4; ++c) {
        const f32x4* ptr = reinterpret_cast<const f32x4*>(row) + (c * 64 + lane);
        v[c] = STREAM ? __builtin_nontemporal_load(ptr) : *ptr;
    }
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) f32x4 lds_f32x4;

// acc += negw * pv (four lanes of a chunk) as two packed FMAs
__device__ __forceinline__ void fma4(f32x4& acc, float negw, const f32x4& pv) {
    const f32x2 ww = {negw, negw};
    const f32x2 lo = __builtin_elementwise_fma(ww, f32x2{pv.x, pv.y}, f32x2{acc.x, acc.y});
    const f32x2 hi = __builtin_elementwise_fma(ww, f32x2{pv.z, pv.w}, f32x2{acc.z, acc.w});
    acc = f32x4{lo.x, lo.y, hi.x, hi.y};
}

// chunk c of LDS-resident vector i through an explicit byte address, so that the moment the read may be issued can be
// pinned with a fake register dependency on `addr` (the scheduler otherwise hoists all NLDS*C ds_read_b128 of a step to
// its top: 48 VGPRs of temporaries at NLDS = 3, and the Gram row gets spilled behind an s_waitcnt vmcnt(0))
template <int C>
__device__ __forceinline__ f32x4 lds_chunk(unsigned addr, int i, int c) {
    return *reinterpret_cast<lds_f32x4*>(addr + (unsigned)((i * C + c) * 1024));
}

// All of x[0..N-1] are produced before anything behind this point runs (keeps a step's indexed reads adjacent, see `steps`).
template <int N>
__device__ __forceinline__ void reads_done(float (&x)[N]) {
    static_assert(N >= 1 && N <= 10, "one operand per value");
    if constexpr (N == 1) asm volatile("" : "+v"(x[0]));
    if constexpr (N == 2) asm volatile("" : "+v"(x[0]), "+v"(x[1]));
    if constexpr (N == 3) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]));
    if constexpr (N == 4) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));
    if constexpr (N == 5) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]));
    if constexpr (N == 6) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]));
    if constexpr (N == 7)
        asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]));
    if constexpr (N == 8)
        asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
    if constexpr (N == 9)
        asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]),
                     "+v"(x[8]));
    if constexpr (N == 10)
        asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]),
                     "+v"(x[8]), "+v"(x[9]));
}

// Steps J..KMAX-1 as a compile-time recursion (see bomp.hip: loops with early exits around convergent cross-lane
// operations are not unrolled, which would push the state to scratch).
template <int R, int KMAX, int NLDS, int NV, int J, bool FAST, bool STAMP = false>
__device__ __forceinline__ void steps(State<R, KMAX, NLDS, NV>& s, const float* __restrict__ G, int k, int lane,
                                      f32x4* lds /* [NLDS][C][64] of this wave; also read through s.laneoff: no restrict */,
                                      float* __restrict__ sc /* scalar area of this wave */, int unit_diag_rt) {
    using L = Lay<R>;
    using S = State<R, KMAX, NLDS, NV>;
    constexpr int C = L::C;
    constexpr int NS = S::NS;
    const bool unit_diag = FAST ? true : (unit_diag_rt != 0);
    if constexpr (J < KMAX) {
        if (!FAST && J >= k) return;
        int kk, Lown, rown;
        unsigned mbits;
        unsigned long long tmid = 0;
        if (!wave_argmax<R, STAMP>(s.vec[0], kk, Lown, rown, mbits, &tmid)) return;
        if constexpr (R == 4) asm("" : "+s"(rown));  // see State::dyn
        if constexpr (STAMP) {
            s.cyc[0] += (unsigned)(tmid - s.tlast);
            s.tlast = tmid;
            unsigned kd = (unsigned)kk;
            W2_STAMP_S(1, kd);
            kk = (int)kd;
        }
        int stop = 0;
        // noise floor (see NOISE_REL in bomp.hip): non-negative floats compare like their bit patterns
        if constexpr (J == 0) {
            s.m0 = (unsigned)__builtin_amdgcn_readfirstlane(
                __builtin_bit_cast(int, NOISE_REL * __builtin_bit_cast(float, mbits)));
        } else {
            stop = (mbits < s.m0) ? 1 : 0;
        }
        // re-selection => stop (sparse_coding.py:323-325); lanes >= J still hold dxv = -1, which never equals kk
        stop |= (__ballot(s.dxv == kk) != 0ull) ? 1 : 0;
        // the last selection needs no vector update: the reference's last `a = a0 - G[:,Dx] z` (:359) is never read
        const bool more = (J + 1 < KMAX) && (FAST || J + 1 < k);
        constexpr bool VIRT_BEFORE = (NV == 1) && (J == KMAX - 1);  // p_{KMAX-2} exists only as scalars
        constexpr int NLJ = (J < NLDS ? J : NLDS);                  // LDS-resident vectors this step reads
        constexpr int NRJ = (J < NS ? J : NS);                      // stored vectors this step reads

        // ---- Gram row of the new atom, and the first two chunks of the LDS-resident vectors: in flight behind the
        // extraction chain
        const float* grow = G + (int64_t)kk * L::Kp;
        f32x4 g[C];
        f32x4 buf[2][NLJ > 0 ? NLJ : 1];
        if (more) load_row4<R, false>(grow, lane, g);
        // the first two chunks of the LDS-resident vectors (measured: queueing them behind the extraction reads instead
        // bought nothing)
        if (more) {
            if constexpr (NLJ > 0) {
                asm("" : "+v"(s.laneoff) : "s"(kk));  // not before kk is known (i.e. not above the argmax)
#pragma unroll
                for (int i = 0; i < NLJ; ++i) buf[0][i] = lds_chunk<C>(s.laneoff, i, 0);
                if constexpr (C > 1) {
#pragma unroll
                    for (int i = 0; i < NLJ; ++i) buf[1][i] = lds_chunk<C>(s.laneoff, i, 1);
                }
            }
        }
        float gsv = 0.f;
        if constexpr (VIRT_BEFORE) gsv = G[(int64_t)s.kkv * L::Kp + kk];  // scalar load through the constant cache
        const float gkk = unit_diag ? 1.f : grow[kk];

        // ---- a[kk], and w_i = p_i[kk]  (== L^-1 G[Dx,kk], sparse_coding.py:331,341); the row of L goes to the LDS scalar area.
        // Element `rown` of a and of every register-resident vector is read in ONE index-mode region: all reads first, one
        // barrier, then the v_readlanes (round 5).  A v_readlane between two indexed reads closes the region (its source
        // would be indexed too), and an s_set_gpr_idx_on / off pair costs the chain ~50 cycles (tools/probes/idx_mode_probe.hip:
        // 84 against 32 ticks per dependent read): the K = 1024 kernel had 2 + up to 4 regions per step.
        float w[KMAX];
        float akk;
        {
            const float* lf = reinterpret_cast<const float*>(lds);
            const int eo = ((rown >> 2) * 64 + Lown) * 4 + (rown & 3);
#pragma unroll
            for (int i = 0; i < NLJ; ++i) {
                w[i] = lf[i * C * 256 + eo];  // every lane reads the same word: a broadcast, the value stays in a VGPR
                sc[J * KMAX + i] = w[i];
            }
            constexpr int NX = 1 + (NRJ > NLDS ? NRJ - NLDS : 0);  // a, then the register vectors of this step
            float xv[NX];
            xv[0] = s.dyn(0, rown);
#pragma unroll
            for (int i = NLDS; i < NRJ; ++i) xv[1 + i - NLDS] = s.dyn(1 + i - NLDS, rown);  // lane Lown holds what we want
            reads_done<NX>(xv);
            akk = readlane_f(xv[0], Lown);
            float tmp[KMAX];
#pragma unroll
            for (int i = NLDS; i < NRJ; ++i) {
                tmp[i] = xv[1 + i - NLDS];
                w[i] = readlane_f(tmp[i], Lown);
            }
            if constexpr (NRJ > NLDS) {
                if (lane == Lown) {
#pragma unroll
                    for (int i = NLDS; i < NRJ; ++i) sc[J * KMAX + i] = tmp[i];
                }
            }
            if constexpr (VIRT_BEFORE) {
                // p_{J-1}[kk] = (G[kk_{J-1}][kk] - sum_l L[J-1][l] w_l) / rho_{J-1}
                float acc = gsv;
#pragma unroll
                for (int l = 0; l < J - 1; ++l) acc = fmaf(-s.wv[l], w[l], acc);
                w[J - 1] = acc * s.invv;
                sc[J * KMAX + J - 1] = w[J - 1];
            }
        }
        // ---- Cholesky pivot: unit Gram diagonal hard-coded by batch_omp (:333-349); 'omp' uses G[kk][kk] (:44-52)
        float vs = gkk;
#pragma unroll
        for (int i = 0; i < J; ++i) vs = fmaf(-w[i], w[i], vs);
        if (J > 0 || !unit_diag) stop |= (__ballot(vs < EPS32_F * gkk) != 0ull) ? 1 : 0;  // reference: vs < eps (:335,345)
        const float inv = __builtin_amdgcn_rsqf(vs);  // 1 ulp (see bomp.hip)
        float t = akk * inv;
        W2_STAMP_V(2, t);
        sc[S::SC_T + J] = t;
        sc[S::SC_I + J] = inv;
        s.dxv = __builtin_bit_cast(int, writelane_sgpr(__builtin_bit_cast(float, s.dxv), __builtin_bit_cast(float, kk), J));

        // ---- vector update: p_J = (G[kk,:] - sum_i w_i p_i) / rho,  a -= t p_J
        if constexpr (J + 1 < KMAX) {
            if (more) {
                // the FMA burst can wait for its issue slots; the chain (argmax -> lookup -> extraction) of another wave cannot:
                // priority 3 from the end of the update to the start of the next one (measured +1.5 %)
                __builtin_amdgcn_s_setprio(0);
                constexpr bool STORE = (J < NS);
                const float tt = STORE ? t : t * inv;  // an unstored p_J is never scaled: a -= (t / rho) * acc
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    f32x4 acc = g[c];
#pragma unroll
                    for (int i = NLDS; i < NRJ; ++i) {
                        const f32x4 pv = f32x4{s.get(1 + i - NLDS, 4 * c), s.get(1 + i - NLDS, 4 * c + 1),
                                               s.get(1 + i - NLDS, 4 * c + 2), s.get(1 + i - NLDS, 4 * c + 3)};
                        fma4(acc, -w[i], pv);
                    }
#pragma unroll
                    for (int i = 0; i < NLJ; ++i) fma4(acc, -w[i], buf[c & 1][i]);
                    if constexpr (STORE) {
                        if constexpr (!(FAST && J == 0)) acc *= inv;  // J = 0, unit diagonal: vs = 1, rsq(1) = 1 exactly
                        if constexpr (J < NLDS) {
                            lds[(J * C + c) * 64 + lane] = acc;
                        } else {
                            s.set(1 + J - NLDS, 4 * c, acc.x);
                            s.set(1 + J - NLDS, 4 * c + 1, acc.y);
                            s.set(1 + J - NLDS, 4 * c + 2, acc.z);
                            s.set(1 + J - NLDS, 4 * c + 3, acc.w);
                        }
                    }
                    f32x4 av = {s.get(0, 4 * c), s.get(0, 4 * c + 1), s.get(0, 4 * c + 2), s.get(0, 4 * c + 3)};
                    fma4(av, -tt, acc);
                    s.set(0, 4 * c, av.x);
                    s.set(0, 4 * c + 1, av.y);
                    s.set(0, 4 * c + 2, av.z);
                    s.set(0, 4 * c + 3, av.w);
                    if constexpr (NLJ > 0) {
                        if (c + 2 < C) {
                            // chunk c+2 of the LDS vectors reuses chunk c's buffer: issue once chunk c is done
                            asm("" : "+v"(s.laneoff) : "v"(av.x));
#pragma unroll
                            for (int i = 0; i < NLJ; ++i) buf[c & 1][i] = lds_chunk<C>(s.laneoff, i, c + 2);
                        }
                    }
                }
                if constexpr (!STORE) {
                    // the last selection needs p_J[kk]: keep this step's row of L, 1/rho and the atom
                    s.kkv = kk;
                    s.invv = inv;
#pragma unroll
                    for (int l = 0; l < J; ++l) s.wv[l] = w[l];
                }
            }
        }
        // keep the update above the exit: without this the optimiser sinks the whole vector update (and the Gram-row
        // load with it) below the branch, i.e. behind the extraction chain
        asm volatile("" : "+v"(s.vec[0]));
        __builtin_amdgcn_s_setprio(3);
        if constexpr (STAMP) {
            float a0 = s.vec[0][0];
            W2_STAMP_V(3, a0);
            s.vec[0][0] = a0;
        }
        if (stop) return;
        s.nsel = J + 1;
        steps<R, KMAX, NLDS, NV, J + 1, FAST, STAMP>(s, G, k, lane, lds, sc, unit_diag_rt);
    }
}

// One signal on one wave: `s.a` holds its alpha0 row; greedy steps, back-substitution, outputs.
template <int R, int KMAX, int NLDS, int NV, bool FAST, bool STAMP>
__device__ __forceinline__ void run_signal(State<R, KMAX, NLDS, NV>& s, const float* __restrict__ G, int64_t sig, int k,
                                           int lane, f32x4* lds_wave, float* __restrict__ sc,
                                           int32_t* __restrict__ idx_out, float* __restrict__ coef_out,
                                           int32_t* __restrict__ nnz_out, int unit_diag) {
    using L = Lay<R>;
    using S = State<R, KMAX, NLDS, NV>;
    constexpr int C = L::C;
    __builtin_amdgcn_s_setprio(3);
    s.m0 = 0u;
    s.dxv = -1;
    s.nsel = 0;
    s.kkv = 0;
    s.invv = 0.f;
    s.laneoff = (unsigned)(uintptr_t)((lds_f32x4*)(lds_wave + lane));  // generic -> LDS address
    unsigned long long tstart = 0;
    if constexpr (STAMP) {
        s.cyc[0] = s.cyc[1] = s.cyc[2] = s.cyc[3] = 0;
        tstart = __builtin_amdgcn_s_memtime();
        float a0 = s.vec[0][0];
        s.tlast = stamp_v(a0);  // alpha0 row has landed
        s.vec[0][0] = a0;
    }
    steps<R, KMAX, NLDS, NV, 0, FAST, STAMP>(s, G, k, lane, lds_wave, sc, unit_diag);
    const int nsel = s.nsel;

    // z = L^-T t  (second triangular solve, sparse_coding.py:354), column-oriented over lanes: lane i of zv ends up as
    // t_i - sum_{j>i} L[j][i] z_j, so z = zv * rinv.  L, t and 1/rho come back from the wave's LDS scalar area with
    // lane-indexed reads (entries of selections that never happened are garbage and masked out).
    const int li = lane < KMAX ? lane : KMAX - 1;
    const float tv = sc[S::SC_T + li];
    const float rinv = sc[S::SC_I + li];
    float rows[KMAX];
#pragma unroll
    for (int j = 1; j < KMAX; ++j) rows[j] = sc[j * KMAX + li];
    float zv = lane < nsel ? tv : 0.f;
#pragma unroll
    for (int j = KMAX - 1; j >= 1; --j) {
        if (j < nsel) {
            const float row = (lane < j) ? rows[j] : 0.f;
            const float zj = readlane_f(zv * rinv, j);
            zv = fmaf(-zj, row, zv);
        }
    }
    float zout = zv * rinv;
    if constexpr (STAMP) {
        // diagnostics instead of coefficients: slot 0 = cycles waiting for the alpha0 row, 1..4 = the four phases summed
        // over the steps, 5 = back-substitution
        const unsigned long long tend = stamp_v(zout);
        const unsigned c0 = (unsigned)(s.tlast - tstart) - (s.cyc[0] + s.cyc[1] + s.cyc[2] + s.cyc[3]);
        zout = lane == 0 ? (float)c0 : lane == 1 ? (float)s.cyc[0] : lane == 2 ? (float)s.cyc[1] : lane == 3 ? (float)s.cyc[2]
               : lane == 4 ? (float)s.cyc[3] : (float)(unsigned)(tend - s.tlast);
        if (lane < k) coef_out[sig * k + lane] = zout;
        if (lane < k) idx_out[sig * k + lane] = (lane < nsel) ? s.dxv : -1;
        if (lane == 0) nnz_out[sig] = nsel;
        return;
    }
    if (lane < k) {
        idx_out[sig * k + lane] = (lane < nsel) ? s.dxv : -1;
        coef_out[sig * k + lane] = (lane < nsel) ? zout : 0.f;
    }
    if (lane == 0) nnz_out[sig] = nsel;
}

// BW = waves per workgroup, WPS = waves per SIMD the register budget is bounded for.
// (A persistent form -- grid sized to the chip, every wave walking signals wave, wave + #waves, ... with the NEXT
// signal's alpha0 row fetched by LDS-DMA (global_load_lds_dwordx4) into a 4-KB slot of the wave -- was built and measured
// in round 3: correct, 13 % SLOWER.  While an LDS-DMA is in flight hipcc turns every counted s_waitcnt vmcnt(N) of the
// Gram-row loads into vmcnt(0), and the loop costs 4 spilled VGPRs; the ~950 cycles a fresh wave waits for its row
// (5 % of its life) stay.  Touching the row of a workgroup 384 / 768 / 1536 places ahead with one strided load per signal
// -- an L2 / Infinity-Cache prefetch -- was 8 % slower as well; the persistent loop with the next row loaded into the
// dead correlation registers behind the back-substitution (ordinary loads, no LDS-DMA): 8 spilled VGPRs, 20 % slower.
// Round 5, K <= 512: 2 / 4 / 8 signals per wave one after the other with the next row requested ahead: within 1 % of one
// signal per wave at K = 256 and 2 % slower at K = 512 -- these kernels are bound by VALU + SALU issue (K = 256, k = 5: 233
// VALU + 311 SALU instructions per signal, both pipes ~80 % busy at 8 waves per SIMD), not by wave launches.)
template <int R, int KMAX, int WPS, int NLDS, int NV, bool FAST, int BW = 4, bool STAMP = false>
__global__ __launch_bounds__(64 * BW, WPS) void bomp_wave2_kernel(const float* __restrict__ alpha0,
                                                                  const float* __restrict__ G, int64_t N, int k,
                                                                  int32_t* __restrict__ idx_out,
                                                                  float* __restrict__ coef_out,
                                                                  int32_t* __restrict__ nnz_out, int unit_diag) {
    using L = Lay<R>;
    using S = State<R, KMAX, NLDS, NV>;
    constexpr int C = L::C;
    __shared__ f32x4 s_p[NLDS > 0 ? BW * NLDS * C * 64 : 1];
    __shared__ float s_sc[BW * S::SC_FLOATS];
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    const int64_t sig = (int64_t)blockIdx.x * BW + wid;
    if (sig >= N) return;
    S s;
    f32x4 a4[C];
    load_row4<R, true>(alpha0 + sig * L::Kp, lane, a4);
#pragma unroll
    for (int c = 0; c < C; ++c) {
        s.set(0, 4 * c, a4[c].x);
        s.set(0, 4 * c + 1, a4[c].y);
        s.set(0, 4 * c + 2, a4[c].z);
        s.set(0, 4 * c + 3, a4[c].w);
    }
    run_signal<R, KMAX, NLDS, NV, FAST, STAMP>(s, G, sig, k, lane, s_p + wid * (NLDS * C * 64), s_sc + wid * S::SC_FLOATS,
                                               idx_out, coef_out, nnz_out, unit_diag);
}

}  // namespace w2
}  // namespace lys
