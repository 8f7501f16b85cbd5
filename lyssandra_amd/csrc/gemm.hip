// fp32 MFMA "NT" GEMM for gfx950:  C[M x Nc] = A[M x Kin] * B[Nc x Kin]^T   (all row-major).
//
// Used for   G      = D'D      (A = B = packed dictionary)            lyssa/sparse_coding.py:630
//            alpha0 = X D      (A = signals, B = packed dictionary)    lyssa/sparse_coding.py:631
//            DA     = A_odl D  (A = A_odl, B = D^T feature-major)      lyssa/dict_learning/online_dict_learn.py:91
//
// v_mfma_f32_32x32x2_f32: exact fp32 (bitwise a k-ordered fmaf chain), 64 FLOP/clk/SIMD.
// Block = 256 threads = 4 waves (2 x 2), block tile 128 x 128, wave tile 64 x 64 = 2 x 2 MFMA tiles
// (64 accumulator registers).  K is consumed in slabs of 32 staged through LDS with a row stride of
// 36 floats: a lane reads its operand as ONE ds_read_b128 (4 consecutive k) which feeds 4 MFMAs, and
// 36 = 4 (mod 32) makes the 16-lane ds_read_b128 groups conflict-free (MI355X_MICROARCH.md, LDS).
// The k-pairing inside an MFMA is (8q+e, 8q+4+e): any pairing is valid because the sum runs over all k.
#include <stdlib.h>

#include "common.h"

namespace lys {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GM_BM = 128, GM_BN = 128, GM_BK = 32, GM_LD = GM_BK + 4;

template <bool STREAM_C>  // STREAM_C: C is write-once/read-once (alpha0) -> non-temporal stores keep G resident in L2
__global__ __launch_bounds__(256) void gemm_nt_f32_kernel(const float* __restrict__ A, int64_t lda,
                                                           const float* __restrict__ B, int64_t ldb,
                                                           float* __restrict__ C, int64_t ldc,
                                                           int64_t M, int Nc, int Kin) {
    __shared__ __attribute__((aligned(16))) float As[GM_BM * GM_LD];
    __shared__ __attribute__((aligned(16))) float Bs[GM_BN * GM_LD];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    // column tiles are the fast grid dimension so that consecutive blocks share the A (signal) tile
    const int n_ct = (Nc + GM_BN - 1) / GM_BN;
    const int64_t bm = (int64_t)(blockIdx.x / n_ct) * GM_BM;
    const int bn = (blockIdx.x % n_ct) * GM_BN;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int lrow = tid >> 3;        // 0..31
    const int lc4 = (tid & 7) * 4;    // 0,4,..,28
    const bool a_vec = ((lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
    const bool b_vec = ((ldb & 3) == 0) && ((reinterpret_cast<uintptr_t>(B) & 15) == 0);

    for (int k0 = 0; k0 < Kin; k0 += GM_BK) {
        // ---- stage A and B slabs (coalesced: 8 threads x 16 B = one 128-B row segment)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = lrow + 32 * i;
            const int kc = k0 + lc4;
            float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = va;
            const int64_t ga = bm + r;
            if (ga < M) {
                const float* p = A + ga * lda + kc;
                if (a_vec && kc + 3 < Kin) {
                    va = *reinterpret_cast<const float4*>(p);
                } else {
                    if (kc + 0 < Kin) va.x = p[0];
                    if (kc + 1 < Kin) va.y = p[1];
                    if (kc + 2 < Kin) va.z = p[2];
                    if (kc + 3 < Kin) va.w = p[3];
                }
            }
            const int gb = bn + r;
            if (gb < Nc) {
                const float* p = B + (int64_t)gb * ldb + kc;
                if (b_vec && kc + 3 < Kin) {
                    vb = *reinterpret_cast<const float4*>(p);
                } else {
                    if (kc + 0 < Kin) vb.x = p[0];
                    if (kc + 1 < Kin) vb.y = p[1];
                    if (kc + 2 < Kin) vb.z = p[2];
                    if (kc + 3 < Kin) vb.w = p[3];
                }
            }
            *reinterpret_cast<float4*>(&As[r * GM_LD + lc4]) = va;
            *reinterpret_cast<float4*>(&Bs[r * GM_LD + lc4]) = vb;
        }
        __syncthreads();
        // ---- MFMA over the slab
        const int h = lane >> 5, l31 = lane & 31;
#pragma unroll
        for (int q = 0; q < GM_BK / 8; ++q) {
            float4 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = *reinterpret_cast<const float4*>(&As[(wm * 64 + i * 32 + l31) * GM_LD + q * 8 + h * 4]);
                b[i] = *reinterpret_cast<const float4*>(&Bs[(wn * 64 + i * 32 + l31) * GM_LD + q * 8 + h * 4]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
    }
    // ---- epilogue: C[row = (r&3) + 8*(r>>2) + 4*(lane>>5)][col = lane&31]
    const int h = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = bn + wn * 64 + j * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = bm + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < M && col < Nc) {
                    if constexpr (STREAM_C)
                        __builtin_nontemporal_store(acc[i][j][r], &C[row * ldc + col]);
                    else
                        C[row * ldc + col] = acc[i][j][r];
                }
            }
        }
}

int gemm_nt(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int Nc,
            int Kin, hipStream_t stream, bool stream_c) {
    if (M <= 0 || Nc <= 0) return LYS_OK;
    const int64_t n_rt = (M + GM_BM - 1) / GM_BM;
    const int n_ct = (Nc + GM_BN - 1) / GM_BN;
    const int64_t blocks = n_rt * n_ct;
    if (blocks > 0x7fffffffLL) {
        set_error("gemm_nt: grid too large (%lld blocks)", (long long)blocks);
        return LYS_ENOSUP;
    }
    if (stream_c)
        hipLaunchKernelGGL(gemm_nt_f32_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, stream, A, lda, B, ldb, C,
                           ldc, M, Nc, Kin);
    else
        hipLaunchKernelGGL(gemm_nt_f32_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, stream, A, lda, B, ldb, C,
                           ldc, M, Nc, Kin);
    LYS_LAUNCH_CHECK();
    return LYS_OK;
}

// ------------------------------------------------------------------------------------------------
// alpha0 = X D for patches with n <= 64 (the metric shape): signal-tile-stationary variant.
// A workgroup keeps its 128-signal tile of X in LDS for the whole kernel and walks the atom tiles of D
// (L2-resident, 256 KB) with the next tile prefetched into registers during the MFMA phase.  The MFMA is issued
// with the operands swapped (rows = atoms, columns = signals) so that a lane ends up with 4 CONSECUTIVE atoms of
// one signal per accumulator quad: the epilogue is 16 non-temporal dwordx4 stores per lane instead of 64 dword
// stores (the plain kernel is store-issue bound: 4 KB written per 131 kFLOP).
// ------------------------------------------------------------------------------------------------
constexpr int A0_LD = 68;  // 64 + 4: conflict-free ds_read_b128 (68 = 4 mod 32)

// MODE 0: direct epilogue (dword stores, 128-B row segments).
// MODE 2: software-pipelined stores -- two accumulator sets; tile t's 64 stores are issued 8 at a time between the MFMA
//         groups of tile t+1, so the store queue drains at a steady rate under the matrix pipe instead of in bursts.
// MODE 1: the epilogue goes through LDS (the atom tile's buffer is free by then) so that every store instruction writes
// two full 512-byte row segments (dwordx4 per lane) instead of two 128-byte segments (dword per lane).
template <int NJ, int MODE>  // NJ atom sub-tiles per wave: workgroup tile = 128 signals x 64*NJ atoms
__global__ __launch_bounds__(256, (NJ == 1 ? 3 : 2)) void alpha0_n64_kernel(const float* __restrict__ X, int64_t ldx,
                                                            const float* __restrict__ D, int ldd,
                                                            float* __restrict__ C, int Kp, int64_t N, int n) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                 // [128][68] signals
    float* Bs = smem + 128 * A0_LD;   // [64*NJ][68] atoms
    constexpr int BN = 64 * NJ;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wsig = wid >> 1, watom = wid & 1;
    const int64_t bm = (int64_t)blockIdx.x * 128;
    const int lrow = tid >> 4;        // 0..15
    const int lc4 = (tid & 15) * 4;   // 0..60
    const bool x_vec = ((ldx & 3) == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
    // ---- stage the signal tile once (zero padded to 64 features)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = lrow + 16 * i;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const int64_t gr = bm + r;
        if (gr < N) {
            const float* p = X + gr * ldx + lc4;
            if (x_vec && lc4 + 3 < n) {
                v = *reinterpret_cast<const float4*>(p);
            } else {
                if (lc4 + 0 < n) v.x = p[0];
                if (lc4 + 1 < n) v.y = p[1];
                if (lc4 + 2 < n) v.z = p[2];
                if (lc4 + 3 < n) v.w = p[3];
            }
        }
        *reinterpret_cast<float4*>(&As[r * A0_LD + lc4]) = v;
    }
    // ---- first atom tile into registers (D is packed: ldd is a multiple of 8 and columns >= n are zero)
    float4 pre[4 * NJ];
    auto fetch = [&](int bn) {
#pragma unroll
        for (int i = 0; i < 4 * NJ; ++i) {
            const int r = lrow + 16 * i;
            pre[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (lc4 < ldd) pre[i] = *reinterpret_cast<const float4*>(D + (int64_t)(bn + r) * ldd + lc4);
        }
    };
    fetch(0);
    const int h = lane >> 5, l31 = lane & 31;
    if constexpr (MODE == 2) {
        f32x16 accA[2][NJ], accB[2][NJ];
        // MODE 2 is launched on whole 128-signal tiles only (no row checks).  Addresses are a wave-uniform 64-bit
        // base (SALU) plus one 32-bit lane offset, so the 64 stores of a tile cost no address VGPRs.
        const int wsig_u = __builtin_amdgcn_readfirstlane(wsig), watom_u = __builtin_amdgcn_readfirstlane(watom);
        // buffer stores: descriptor = this workgroup's 128 rows of C (SGPRs), one VGPR lane offset, SALU row/column offset
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc(C + bm * Kp, 0, 128 * Kp * (int)sizeof(float), 0x00020000);
        const int lane_off = ((4 * h) * Kp + l31) * (int)sizeof(float);
        auto store_part = [&](f32x16 (&acc)[2][NJ], int bn, int part) {   // 8 parts of 4*NJ stores
            const int i = part >> 2;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int r = (part & 3) * 4 + rr;
                    const int soff = ((wsig_u * 64 + i * 32 + (r & 3) + 8 * (r >> 2)) * Kp + bn + watom_u * 32 * NJ + j * 32) *
                                     (int)sizeof(float);
                    const float val = acc[i][j][r];
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val), rsrc, lane_off, soff, 2 /* nt */);
                }
            }
        };
        auto tile = [&](f32x16 (&cur)[2][NJ], f32x16 (&prev)[2][NJ], int bn, bool have_prev) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 4 * NJ; ++i) *reinterpret_cast<float4*>(&Bs[(lrow + 16 * i) * A0_LD + lc4]) = pre[i];
            __syncthreads();
            if (bn + BN < Kp) fetch(bn + BN);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) cur[i][j][r] = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float4 a[2], b[NJ];
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    a[i] = *reinterpret_cast<const float4*>(&As[(wsig * 64 + i * 32 + l31) * A0_LD + q * 8 + h * 4]);
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    b[j] = *reinterpret_cast<const float4*>(&Bs[(watom * 32 * NJ + j * 32 + l31) * A0_LD + q * 8 + h * 4]);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        cur[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, cur[i][j], 0, 0, 0);
                        cur[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, cur[i][j], 0, 0, 0);
                        cur[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, cur[i][j], 0, 0, 0);
                        cur[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, cur[i][j], 0, 0, 0);
                    }
                if (have_prev) store_part(prev, bn - BN, q);
            }
        };
        // at the loop back edge the pending (not yet stored) tile is always in accB
        bool pending_b = false;
        for (int bn = 0; bn < Kp; bn += 2 * BN) {
            tile(accA, accB, bn, bn > 0);
            if (bn + BN < Kp) {
                tile(accB, accA, bn + BN, true);
                pending_b = true;
            } else {
                pending_b = false;
#pragma unroll
                for (int part = 0; part < 8; ++part) store_part(accA, bn, part);
            }
        }
        if (pending_b) {
#pragma unroll
            for (int part = 0; part < 8; ++part) store_part(accB, Kp - BN, part);
        }
        return;
    }
    for (int bn = 0; bn < Kp; bn += BN) {
        __syncthreads();  // previous tile's LDS reads are done (and As is visible on the first pass)
#pragma unroll
        for (int i = 0; i < 4 * NJ; ++i) *reinterpret_cast<float4*>(&Bs[(lrow + 16 * i) * A0_LD + lc4]) = pre[i];
        __syncthreads();
        if (bn + BN < Kp) fetch(bn + BN);  // prefetch behind the MFMAs
        f32x16 acc[2][NJ];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            float4 a[2], b[NJ];
#pragma unroll
            for (int i = 0; i < 2; ++i)
                a[i] = *reinterpret_cast<const float4*>(&As[(wsig * 64 + i * 32 + l31) * A0_LD + q * 8 + h * 4]);   // signals
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                b[j] = *reinterpret_cast<const float4*>(&Bs[(watom * 32 * NJ + j * 32 + l31) * A0_LD + q * 8 + h * 4]);  // atoms
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
                }
        }
        if constexpr (MODE == 1 && NJ == 2) {
            constexpr int SLD = 132;  // staging row stride (floats): 64 rows x 132 x 4 B = 33 792 B <= the Bs region
            float* Stg = Bs;
            __syncthreads();          // every wave is done reading this atom tile
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                if (wsig == half) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int r = 0; r < 16; ++r)
                                Stg[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * SLD + watom * 64 + j * 32 + l31] =
                                    acc[i][j][r];
                }
                __syncthreads();
                // 64 rows x 128 floats = 2048 float4: 8 per thread; a wave stores 2 rows x 512 B per instruction
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int v = tid + 256 * t;
                    const int row = v >> 5, c4 = (v & 31) * 4;
                    const int64_t grow = bm + half * 64 + row;
                    if (grow < N) {
                        typedef float f32x4_ __attribute__((ext_vector_type(4)));
                        const f32x4_ val = *reinterpret_cast<const f32x4_*>(&Stg[row * SLD + c4]);
                        __builtin_nontemporal_store(val, reinterpret_cast<f32x4_*>(C + grow * Kp + bn + c4));
                    }
                }
                __syncthreads();
            }
        } else {
            // ---- epilogue (rows = signals): acc[i][j][r] = C[signal = wsig*64 + i*32 + (r&3) + 8(r>>2) + 4h][atom = .. + l31]
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int col = bn + watom * 32 * NJ + j * 32 + l31;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int64_t row = bm + wsig * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        if (row < N) __builtin_nontemporal_store(acc[i][j][r], &C[row * Kp + col]);
                    }
                }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// alpha0 = X D on the bf16 matrix cores with fp32 accuracy ("bf16x3").  Every fp32 operand is split into three bf16
// planes, x = x1 + x2 + x3 (x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2): 24 mantissa bits, fp32's exponent
// range), and the product is the six plane products with i + j <= 4 -- (1,1) (1,2) (2,1) (1,3) (3,1) (2,2) -- accumulated
// in fp32 by v_mfma_f32_32x32x16_bf16; the dropped terms are below 2^-25 of the result, a bf16 x bf16 product is exact
// in fp32.  Six MFMAs at 16x the fp32 matrix rate = 2.7x fewer matrix-pipe cycles than the fp32 kernel above, which
// leaves the kernel bound by its 1 GiB of alpha0 stores alone (measured floor of the store scheme: 0.23 ms per 262 144
// signals; fp32 kernel 0.35 ms).  Same tile walk, same software-pipelined buffer stores as MODE 2 above.
// Which feature a (register slot, lane half) pair of the MFMA's K = 16 carries is irrelevant as long as A and B use the
// same assignment: slot s of half h carries feature 16 ks + 8 h + s for both.
// ------------------------------------------------------------------------------------------------
__host__ __device__ constexpr int b3_slot(int k0) { return k0 < 3 ? k0 : k0 - 1; }  // store positions 0,1,2,4,5,6 of a k-step -> 0..5
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int B3_LD = 72;  // bf16 elements per LDS row (144 B = 36 dwords = 4 mod 32: conflict-free ds_read_b128)

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
// three bf16 planes of two floats (packed lo | hi << 16 per plane)
__device__ __forceinline__ void split3(float a, float b, unsigned& p1, unsigned& p2, unsigned& p3) {
    p1 = cvt_pk_bf16(a, b);
    // a leading plane that is not finite (x = +-inf, or |x| above bf16's largest finite 3.39e38, which rounds to inf)
    // carries the whole value: its residual planes are flushed to zero instead of becoming inf - inf = NaN.  The product
    // is then +-inf (or NaN against a zero, like any inf) where the fp32 MFMA kernel would return the same for x = inf
    // and a finite 3.4e38-sized number for 3.39e38 < |x| <= FLT_MAX: saturation, documented in DESIGN.md 3.1.
    const float ha = __uint_as_float(p1 << 16), hb = __uint_as_float(p1 & 0xffff0000u);
    const float ra = (fabsf(ha) < __builtin_inff()) ? a - ha : 0.f, rb = (fabsf(hb) < __builtin_inff()) ? b - hb : 0.f;
    p2 = cvt_pk_bf16(ra, rb);
    const float sa = ra - __uint_as_float(p2 << 16), sb = rb - __uint_as_float(p2 & 0xffff0000u);
    p3 = cvt_pk_bf16(sa, sb);
}

// D_packed [Kp][ldd] fp32 -> Dsp [3][Kp][64] bf16 (features >= min(n, ldd) are zero)
__global__ __launch_bounds__(256) void split_bf16x3_kernel(const float* __restrict__ D, int ldd, int Kp, int n,
                                                           unsigned* __restrict__ Dsp) {
    const int t = blockIdx.x * 256 + threadIdx.x;  // one feature PAIR
    if (t >= Kp * 32) return;
    const int a = t >> 5, f = (t & 31) * 2;
    const float v0 = (f < n && f < ldd) ? D[(int64_t)a * ldd + f] : 0.f;
    const float v1 = (f + 1 < n && f + 1 < ldd) ? D[(int64_t)a * ldd + f + 1] : 0.f;
    unsigned p1, p2, p3;
    split3(v0, v1, p1, p2, p3);
    Dsp[t] = p1;
    Dsp[(size_t)Kp * 32 + t] = p2;
    Dsp[(size_t)2 * Kp * 32 + t] = p3;
}

// Workgroup = 4 waves = 128 signals x 64 atoms per iteration; wave (wsig, watom) owns 64 signals x 32 atoms.  The
// SIGNAL fragments are loop invariant and live in registers (2 blocks x 4 k-steps x 3 planes x 4 VGPRs = 96): only the
// atom planes go through LDS (3 x 64 x 144 B per buffer, double buffered: one barrier per iteration), which leaves room
// for two workgroups per CU -- with one, every prologue / barrier / store drain is exposed (measured 0.39 ms against
// 0.35 ms for the fp32 kernel).
// XPOSE (round 4): the stores of a tile go through a wave-private 4-KB LDS transposition and leave as dwordx4 row
// stores (8 per tile and wave instead of 32 dword stores), still spread between the MFMAs of the next tile.  The counters
// say why (tools/pmc_store.sh): with dword stores the SQ -> TA address FIFO of the alpha0 kernel is full 97 % of the
// kernel's duration (SQ_VMEM_TA_ADDR_FIFO_FULL) -- the kernel is bound by the NUMBER of store instructions the texture
// addresser takes (one per ~38 cycles and CU = 3.6 TB/s), not by DRAM.
template <bool XPOSE>
__global__ __launch_bounds__(256, 2) void alpha0_n64_bf16x3_kernel(const float* __restrict__ X, int64_t ldx,
                                                                  const unsigned* __restrict__ Dsp,
                                                                  float* __restrict__ C, int Kp, int n) {
    extern __shared__ __attribute__((aligned(16))) unsigned short smem3[];
    constexpr int BUF = 3 * 64 * B3_LD;               // bf16 elements of one atom-tile buffer
    unsigned short* Bs = smem3;                       // [2][3][64][B3_LD]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wsig = __builtin_amdgcn_readfirstlane(wid >> 1), watom = __builtin_amdgcn_readfirstlane(wid & 1);
    const int64_t bm = (int64_t)blockIdx.x * 128;
    const int h = lane >> 5, l31 = lane & 31;
    // ---- signal fragments: the fp32 tile goes through LDS once (coalesced global reads), then every lane splits its own
    // 2 blocks x 4 k-steps x 8 features into the three planes
    bf16x8 afr[2][4][3];
    {
        float* Xs = reinterpret_cast<float*>(smem3);  // [128][68] floats = 34 816 B <= the two atom buffers (55 296 B)
        const bool x_vec = ((ldx & 3) == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
        const int lrow = tid >> 4, lc4 = (tid & 15) * 4;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = lrow + 16 * i;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            const float* p = X + (bm + r) * ldx + lc4;
            if (x_vec && lc4 + 3 < n) {
                v = *reinterpret_cast<const float4*>(p);
            } else {
                if (lc4 + 0 < n) v.x = p[0];
                if (lc4 + 1 < n) v.y = p[1];
                if (lc4 + 2 < n) v.z = p[2];
                if (lc4 + 3 < n) v.w = p[3];
            }
            *reinterpret_cast<float4*>(&Xs[r * A0_LD + lc4]) = v;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const float* src = &Xs[(wsig * 64 + i * 32 + l31) * A0_LD + ks * 16 + h * 8];
                const float4 u = *reinterpret_cast<const float4*>(src), w = *reinterpret_cast<const float4*>(src + 4);
                unsigned p1[4], p2[4], p3[4];
                split3(u.x, u.y, p1[0], p2[0], p3[0]);
                split3(u.z, u.w, p1[1], p2[1], p3[1]);
                split3(w.x, w.y, p1[2], p2[2], p3[2]);
                split3(w.z, w.w, p1[3], p2[3], p3[3]);
                afr[i][ks][0] = __builtin_bit_cast(bf16x8, (u32x4){p1[0], p1[1], p1[2], p1[3]});
                afr[i][ks][1] = __builtin_bit_cast(bf16x8, (u32x4){p2[0], p2[1], p2[2], p2[3]});
                afr[i][ks][2] = __builtin_bit_cast(bf16x8, (u32x4){p3[0], p3[1], p3[2], p3[3]});
            }
        __syncthreads();  // the atom buffers may now overwrite the staging area
    }
    // ---- atom tile prefetch: 3 planes x 64 atoms x 128 B = 1536 x 16 B, 6 per thread
    u32x4 pre[6];
    auto fetch = [&](int bn) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int e = tid + 256 * i;            // 16-byte chunk: plane = e / 512, atom = (e / 8) % 64, chunk = e % 8
            const int pl = e >> 9, a = (e >> 3) & 63, ch = e & 7;
            pre[i] = *reinterpret_cast<const u32x4*>(Dsp + ((size_t)pl * Kp + bn + a) * 32 + ch * 4);
        }
    };
    auto stage_b = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int e = tid + 256 * i;
            const int pl = e >> 9, a = (e >> 3) & 63, ch = e & 7;
            *reinterpret_cast<u32x4*>(&Bs[buf * BUF + (pl * 64 + a) * B3_LD + ch * 8]) = pre[i];
        }
    };
    fetch(0);
    stage_b(0);
    f32x16 accA[2], accB[2];
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(C + bm * Kp, 0, 128 * Kp * (int)sizeof(float), 0x00020000);
    const int lane_off = ((4 * h) * Kp + l31) * (int)sizeof(float);
    auto store_part = [&](f32x16 (&acc)[2], int bn, int part) __attribute__((always_inline)) {   // 4 parts of 8 stores
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            const int i = part >> 1, r = (part & 1) * 8 + rr;
            const int soff = ((wsig * 64 + i * 32 + (r & 3) + 8 * (r >> 2)) * Kp + bn + watom * 32) * (int)sizeof(float);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[i][r]), rsrc, lane_off, soff, 2 /* nt */);
        }
    };
    // stores number first .. first + cnt - 1 of a tile's 32 (number = 16 * block + accumulator register)
    auto store_some = [&](f32x16 (&acc)[2], int bn, int first, int cnt) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if (c < cnt) {
                const int sidx = first + c, i = sidx >> 4, r = sidx & 15;
                const int soff = ((wsig * 64 + i * 32 + (r & 3) + 8 * (r >> 2)) * Kp + bn + watom * 32) * (int)sizeof(float);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[i][r]), rsrc, lane_off, soff, 2 /* nt */);
            }
        }
    };
    // ---- XPOSE epilogue: 24 slots per tile (one behind every MFMA pair).  Signal block i = slot / 12: slots 0..7 write the
    // block's 16 accumulator registers to the wave's LDS region [32 signals][32 atoms] (unpadded: the b128 lane groups of
    // the read-back cover all 64 banks), slot 8 reads them back as four float4 (lane = (signal % 8, atom quad)), slot 10
    // issues the four dwordx4 stores (8 signal rows x 128 B each)
    float* Tw = reinterpret_cast<float*>(smem3) + BUF + wid * 1024;   // behind the two atom buffers (2 BUF shorts = BUF floats)
    const int t_wr = (4 * h) * 32 + l31, t_rd = (lane >> 3) * 32 + (lane & 7) * 4;
    const int lane_off4 = ((lane >> 3) * Kp + (lane & 7) * 4) * (int)sizeof(float);
    f32x4v tq[4];
    auto epi = [&](f32x16 (&acc)[2], int bn, int slot) __attribute__((always_inline)) {
        const int i = slot / 12, sl = slot % 12;
        if (sl < 8) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int r = 2 * sl + c;
                Tw[((r & 3) + 8 * (r >> 2)) * 32 + t_wr] = acc[i][r];
            }
        } else if (sl == 8) {
#pragma unroll
            for (int p = 0; p < 4; ++p) tq[p] = *reinterpret_cast<const f32x4v*>(&Tw[t_rd + 8 * p * 32]);
        } else if (sl == 10) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int soff = ((wsig * 64 + i * 32 + 8 * p) * Kp + bn + watom * 32) * (int)sizeof(float);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, tq[p]), rsrc, lane_off4, soff, 2 /* nt */);
            }
        }
    };
    int it = 0;
    auto tile = [&](f32x16 (&cur)[2], f32x16 (&prev)[2], int bn, bool have_prev) __attribute__((always_inline)) {
        const int buf = it & 1;
        ++it;
        __syncthreads();     // buffer `buf` was written at the end of the previous iteration; nobody reads buf ^ 1 any more
        const bool more = bn + 64 < Kp;
        if (more) fetch(bn + 64);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) cur[i][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 b[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                b[pl] = *reinterpret_cast<const bf16x8*>(&Bs[buf * BUF + (pl * 64 + watom * 32 + l31) * B3_LD + ks * 16 + h * 8]);
            // small terms first; the two signal blocks alternate (a dependent MFMA waits for its predecessor); the
            // previous tile's stores are spread between the MFMA pairs (a store issues in the shadow of a running MFMA)
#define B3_ST(k0, cnt) do { if (have_prev) { if constexpr (XPOSE) epi(prev, bn - 64, ks * 6 + b3_slot(k0)); else store_some(prev, bn - 64, ks * 8 + (k0), (cnt)); } } while (0)
            cur[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[0][ks][2], b[0], cur[0], 0, 0, 0);
            cur[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[1][ks][2], b[0], cur[1], 0, 0, 0);
            B3_ST(0, 1);
            cur[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[0][ks][0], b[2], cur[0], 0, 0, 0);
            cur[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[1][ks][0], b[2], cur[1], 0, 0, 0);
            B3_ST(1, 1);
            cur[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[0][ks][1], b[1], cur[0], 0, 0, 0);
            cur[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[1][ks][1], b[1], cur[1], 0, 0, 0);
            B3_ST(2, 2);
            cur[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[0][ks][1], b[0], cur[0], 0, 0, 0);
            cur[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[1][ks][1], b[0], cur[1], 0, 0, 0);
            B3_ST(4, 1);
            cur[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[0][ks][0], b[1], cur[0], 0, 0, 0);
            cur[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[1][ks][0], b[1], cur[1], 0, 0, 0);
            B3_ST(5, 1);
            cur[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[0][ks][0], b[0], cur[0], 0, 0, 0);
            cur[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[1][ks][0], b[0], cur[1], 0, 0, 0);
            B3_ST(6, 2);
#undef B3_ST
        }
        // the next tile's planes go to the other LDS buffer HERE, in the block that issued their loads: the loads are
        // older than this iteration's 32 stores, so waiting for them (vmcnt counts in order) does not wait for the
        // stores -- a wait carried across the loop back edge is emitted as vmcnt(0..5) and drains the store queue
        if (more) stage_b(buf ^ 1);
    };
    bool pending_b = false;
    for (int bn = 0; bn < Kp; bn += 128) {
        tile(accA, accB, bn, bn > 0);
        if (bn + 64 < Kp) {
            tile(accB, accA, bn + 64, true);
            pending_b = true;
        } else {
            pending_b = false;
            if constexpr (XPOSE) {
#pragma unroll
                for (int slot = 0; slot < 24; ++slot) epi(accA, bn, slot);
            } else {
#pragma unroll
                for (int part = 0; part < 4; ++part) store_part(accA, bn, part);
            }
        }
    }
    if (pending_b) {
        if constexpr (XPOSE) {
#pragma unroll
            for (int slot = 0; slot < 24; ++slot) epi(accB, Kp - 64, slot);
        } else {
#pragma unroll
            for (int part = 0; part < 4; ++part) store_part(accB, Kp - 64, part);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Round 4: the same bf16x3 product for n > 64 (configs[2]: 256-dim patches, 4096 atoms; configs[3]: 128-dim descriptors) as a
// k-looped NT GEMM.  C[M x Nc] = A[M x Kin] * B[Nc x Kin]^T with B given as its three bf16 planes (split once per
// dictionary, rows padded to a multiple of 32 features), A split on the fly by the thread that loads it.  128 x 128 block
// tile, 4 waves x (2 x 2) MFMA tiles of v_mfma_f32_32x32x16_bf16, K consumed in slabs of 32 features; the next slab is
// fetched into registers behind the 48 MFMAs of the current one (the fp32 kernel above has no prefetch: with the matrix
// time cut 5x the exposed load latency would be all that is left).  LDS: 2 x 3 planes x 128 rows x 80 B = 60 KB, two
// workgroups per CU.  Block order: column tiles fastest, so that the 8 XCDs (blocks are dealt round-robin) each keep 1/8
// of the dictionary planes and the current signal tiles in their own L2.
// ------------------------------------------------------------------------------------------------
constexpr int GB_LDP = 40;  // bf16 elements per LDS row: 32 + 8 (80 B = 20 dwords: the b128 reads of 8 lanes cover all banks)

__global__ __launch_bounds__(256) void split_bf16x3_rows_kernel(const float* __restrict__ D, int ldd, int Kp, int n, int ldp,
                                                                unsigned* __restrict__ Dsp) {
    const int hp = ldp >> 1;                           // feature pairs per row
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (int64_t)Kp * hp) return;
    const int a = (int)(t / hp), f = (int)(t % hp) * 2;
    const float v0 = (f < n && f < ldd) ? D[(int64_t)a * ldd + f] : 0.f;
    const float v1 = (f + 1 < n && f + 1 < ldd) ? D[(int64_t)a * ldd + f + 1] : 0.f;
    unsigned p1, p2, p3;
    split3(v0, v1, p1, p2, p3);
    Dsp[t] = p1;
    Dsp[(size_t)Kp * hp + t] = p2;
    Dsp[(size_t)2 * Kp * hp + t] = p3;
}

// WAVES = 8 (512 threads, wave tile 64 x 32, 4 waves per SIMD with two workgroups per CU) is the product: with WAVES = 4
// (wave tile 64 x 64, 246 VGPRs, 2 waves per SIMD) the one-slab register prefetch does not cover the L2 latency behind 48
// MFMAs and the matrix cores idle 60 % of the time (measured: 1.75 ms against 2.60 for the fp32 kernel at configs[2]).
template <bool STREAM_C, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 2) void gemm_nt_bf16x3_kernel(const float* __restrict__ A, int64_t lda,
                                                                      const unsigned* __restrict__ Bsp, int ldp,
                                                                      float* __restrict__ C, int64_t ldc, int64_t M, int Nc,
                                                                      int Kin) {
    constexpr int NT = 64 * WAVES;
    constexpr int NJ = (WAVES == 4) ? 2 : 1;        // 32-atom tiles per wave
    constexpr int A_IT = 1024 / NT, B_IT = 1536 / NT;
    // ONE array: the epilogue's wave-private transposition regions (WAVES x 4 KB) start at its base and run past the A
    // planes into the B planes -- both are dead after the loop's last barrier
    constexpr int GB_PLANES = 3 * 128 * GB_LDP;     // bf16 elements of one operand's three planes
    __shared__ __attribute__((aligned(16))) unsigned short S3[2 * GB_PLANES];
    static_assert(WAVES * 4096 <= 2 * GB_PLANES * (int)sizeof(unsigned short), "epilogue transposition regions exceed the staging LDS");
    unsigned short* const As = S3;
    unsigned short* const Bs = S3 + GB_PLANES;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = (WAVES == 4) ? (wid >> 1) : (wid >> 2), wn = (WAVES == 4) ? (wid & 1) : (wid & 3);
    const int n_ct = Nc / 128;
    const int64_t bm = (int64_t)(blockIdx.x / n_ct) * 128;
    const int bn = (blockIdx.x % n_ct) * 128;
    const int h = lane >> 5, l31 = lane & 31;
    const int lrow = tid >> 3, lc4 = (tid & 7) * 4;
    const bool a_vec = ((lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
    const int hp = ldp >> 1;

    f32x16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4v pa[A_IT];   // next A slab: rows lrow + (NT / 8) i, features k0 + lc4 .. + 3
    u32x4 pb[B_IT];    // next B slab: 3 planes x 128 rows x 64 B = 1536 chunks of 16 B
    auto fetch = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            int64_t ga = bm + lrow + (NT / 8) * i;
            ga = (ga < M) ? ga : M - 1;                   // rows past the end repeat the last one (never stored)
            const float* p = A + ga * lda + k0 + lc4;
            const int kc = k0 + lc4;
            f32x4v v = {0.f, 0.f, 0.f, 0.f};
            if (a_vec && kc + 3 < Kin) {
                v = *reinterpret_cast<const f32x4v*>(p);
            } else {
                if (kc + 0 < Kin) v.x = p[0];
                if (kc + 1 < Kin) v.y = p[1];
                if (kc + 2 < Kin) v.z = p[2];
                if (kc + 3 < Kin) v.w = p[3];
            }
            pa[i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int e = tid + NT * i;                    // plane = e / 512, row = (e / 4) % 128, 16-byte chunk = e % 4
            const int pl = e >> 9, row = (e >> 2) & 127, ch = e & 3;
            pb[i] = *reinterpret_cast<const u32x4*>(Bsp + ((size_t)pl * Nc + bn + row) * hp + (k0 >> 1) + ch * 4);
        }
    };
    auto stage = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            unsigned p1[2], p2[2], p3[2];
            split3(pa[i].x, pa[i].y, p1[0], p2[0], p3[0]);
            split3(pa[i].z, pa[i].w, p1[1], p2[1], p3[1]);
            const int at = (lrow + (NT / 8) * i) * GB_LDP + lc4;
            *reinterpret_cast<uint2*>(&As[0 * 128 * GB_LDP + at]) = make_uint2(p1[0], p1[1]);
            *reinterpret_cast<uint2*>(&As[1 * 128 * GB_LDP + at]) = make_uint2(p2[0], p2[1]);
            *reinterpret_cast<uint2*>(&As[2 * 128 * GB_LDP + at]) = make_uint2(p3[0], p3[1]);
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int e = tid + NT * i;
            const int pl = e >> 9, row = (e >> 2) & 127, ch = e & 3;
            *reinterpret_cast<u32x4*>(&Bs[(pl * 128 + row) * GB_LDP + ch * 8]) = pb[i];
        }
    };
    fetch(0);
    stage();
    __syncthreads();
    for (int k0 = 0; k0 < Kin; k0 += 32) {
        const bool more = k0 + 32 < Kin;
        if (more) fetch(k0 + 32);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a[2][3], b[NJ][3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    a[i][pl] = *reinterpret_cast<const bf16x8*>(&As[(pl * 128 + wm * 64 + i * 32 + l31) * GB_LDP + ks * 16 + h * 8]);
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    b[j][pl] = *reinterpret_cast<const bf16x8*>(&Bs[(pl * 128 + wn * (32 * NJ) + j * 32 + l31) * GB_LDP + ks * 16 + h * 8]);
            }
            // the six plane products with i + j <= 4, small terms first; the output tiles alternate (a dependent MFMA waits
            // for its predecessor)
#define GB_MF(PA, PB)                                                                                              \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < NJ; ++j) acc[i][j] =       \
        __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA], b[j][PB], acc[i][j], 0, 0, 0)
            GB_MF(2, 0);
            GB_MF(0, 2);
            GB_MF(1, 1);
            GB_MF(1, 0);
            GB_MF(0, 1);
            GB_MF(0, 0);
#undef GB_MF
        }
        __syncthreads();                 // every wave is done with this slab
        if (more) {
            stage();
            __syncthreads();
        }
    }
    // epilogue: every 32 x 32 accumulator tile through a wave-private 4-KB LDS transposition (the staging buffers are free:
    // the loop ended with a barrier), then four dwordx4 row stores of 8 rows x 128 B instead of 16 dword stores -- the dword
    // form is bound by the number of store instructions the texture addresser takes (see alpha0_n64_bf16x3_kernel)
    float* Tw = reinterpret_cast<float*>(S3) + wid * 1024;   // WAVES x 4 KB inside the 60 KB of S3 (static_assert above)
    const int t_wr = (4 * h) * 32 + l31, t_rd = (lane >> 3) * 32 + (lane & 7) * 4;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r) Tw[((r & 3) + 8 * (r >> 2)) * 32 + t_wr] = acc[i][j][r];
            f32x4v tq[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) tq[p] = *reinterpret_cast<const f32x4v*>(&Tw[t_rd + 8 * p * 32]);
            const int col = bn + wn * (32 * NJ) + j * 32 + (lane & 7) * 4;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int64_t row = bm + wm * 64 + i * 32 + 8 * p + (lane >> 3);
                if (row < M) {
                    f32x4v* dst = reinterpret_cast<f32x4v*>(&C[row * ldc + col]);
                    if constexpr (STREAM_C)
                        __builtin_nontemporal_store(tq[p], dst);
                    else
                        *dst = tq[p];
                }
            }
        }
}

static int bf16x3_ldp(int n) { return ((n + 31) / 32) * 32; }  // plane row length in features

// room for the dictionary's three bf16 planes: n <= 64 -> [3][Kp][64] (alpha0_n64_bf16x3_kernel), else [3][Kp][ldp]
size_t alpha0_bf16x3_scratch_bytes(int Kp, int n) {
    return (size_t)3 * Kp * (n <= 64 ? 64 : bf16x3_ldp(n)) * sizeof(unsigned short);
}
bool alpha0_split_path(int n, int Kp) { return (Kp % 128) == 0 && n >= 1; }

int gemm_nt_bf16x3(const float* A, int64_t lda, const void* Bsp, int n, float* C, int64_t ldc, int64_t M, int Nc,
                   hipStream_t stream) {
    if (M <= 0) return LYS_OK;
    const int64_t blocks = ((M + 127) / 128) * (Nc / 128);
    if ((Nc % 128) != 0 || blocks > 0x7fffffffLL) {
        set_error("gemm_nt_bf16x3: Nc = %d, %lld blocks", Nc, (long long)blocks);
        return LYS_ENOSUP;
    }
    // 8 waves per workgroup (wave tile 32 x 64; the 4-wave form with 64 x 64 wave tiles measured 1.75 against 1.63 ms at
    // configs[2] in round 4 and is no longer instantiated)
    hipLaunchKernelGGL((gemm_nt_bf16x3_kernel<true, 8>), dim3((unsigned)blocks), dim3(512), 0, stream, A, lda,
                       static_cast<const unsigned*>(Bsp), bf16x3_ldp(n), C, ldc, M, Nc, n);
    LYS_LAUNCH_CHECK();
    return LYS_OK;
}

size_t alpha0_bf16x3_scratch_bytes(int Kp) { return (size_t)3 * Kp * 64 * sizeof(unsigned short); }

// whole 128-signal tiles through the bf16x3 kernel, the tail through the fp32 kernel; `scratch`: alpha0_bf16x3_scratch_bytes
int alpha0_n64(const float* X, int64_t ldx, const float* D, int ldd, float* C, int Kp, int64_t N, int n, hipStream_t stream);
// the dictionary's three bf16 planes, once per dictionary (callers that encode several tiles against one D split once)
int alpha0_bf16x3_split(const float* D, int ldd, int Kp, int n, void* scratch, hipStream_t stream) {
    if (n > 64) {
        const int ldp = bf16x3_ldp(n);
        const int64_t pairs = (int64_t)Kp * (ldp >> 1);
        hipLaunchKernelGGL(split_bf16x3_rows_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, stream, D, ldd, Kp, n,
                           ldp, static_cast<unsigned*>(scratch));
        LYS_LAUNCH_CHECK();
        return LYS_OK;
    }
    hipLaunchKernelGGL(split_bf16x3_kernel, dim3((unsigned)((Kp * 32 + 255) / 256)), dim3(256), 0, stream, D, ldd, Kp, n,
                       static_cast<unsigned*>(scratch));
    LYS_LAUNCH_CHECK();
    return LYS_OK;
}

// `presplit`: scratch already holds the planes of D (alpha0_bf16x3_split)
int alpha0_n64_bf16x3(const float* X, int64_t ldx, const float* D, int ldd, float* C, int Kp, int64_t N, int n,
                      void* scratch, hipStream_t stream, bool presplit) {
    if (N <= 0) return LYS_OK;
    const int64_t whole = N / 128, tail = N - whole * 128;
    if (whole > 0x7fffffffLL) {
        set_error("alpha0: grid too large");
        return LYS_ENOSUP;
    }
    if (whole) {
        static bool attr_set[64] = {false};
        int dev = 0;
        LYS_CHECK_HIP(hipGetDevice(&dev));
        const int lds0 = 2 * 3 * 64 * B3_LD * (int)sizeof(unsigned short);   // two atom-tile buffers (>= the one-off fp32 staging)
        const int lds1 = lds0 + 4 * 1024 * (int)sizeof(float);               // + the four waves' transposition regions
        if (dev >= 0 && dev < 64 && !attr_set[dev]) {
            LYS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(alpha0_n64_bf16x3_kernel<true>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, lds1));

            attr_set[dev] = true;
        }
        unsigned* Dsp = static_cast<unsigned*>(scratch);
        if (!presplit)
            hipLaunchKernelGGL(split_bf16x3_kernel, dim3((unsigned)((Kp * 32 + 255) / 256)), dim3(256), 0, stream, D, ldd, Kp, n, Dsp);
        // (the dword-store epilogue of rounds 2-3, XPOSE = false, lost its A/B in round 4 and is no longer instantiated)
        hipLaunchKernelGGL(alpha0_n64_bf16x3_kernel<true>, dim3((unsigned)whole), dim3(256), lds1, stream, X, ldx, Dsp, C, Kp, n);
        LYS_LAUNCH_CHECK();
    }
    if (tail) return alpha0_n64(X + whole * 128 * ldx, ldx, D, ldd, C + whole * 128 * Kp, Kp, tail, n, stream);
    return LYS_OK;
}

bool alpha0_fast_path(int n, int Kp) { return n <= 64 && (Kp % 128) == 0; }

int alpha0_n64(const float* X, int64_t ldx, const float* D, int ldd, float* C, int Kp, int64_t N, int n,
               hipStream_t stream) {
    if (N <= 0) return LYS_OK;
    // 128 atoms per workgroup tile (2 workgroups per CU: 80.7 TFLOP/s against 77.7 with 64 atoms and 3 per CU -- occupancy is
    // not the limiter, the alpha0 stores are) with software-pipelined stores; the other tile / epilogue forms of rounds 1-3
    // (LYS_ALPHA0_BN, LYS_ALPHA0_XPOSE) lost their A/Bs and are no longer instantiated
    const size_t lds = (size_t)(128 + 64 * 2) * A0_LD * sizeof(float);
    static bool attr_set[64] = {false};
    int dev = 0;
    LYS_CHECK_HIP(hipGetDevice(&dev));
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        LYS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(alpha0_n64_kernel<2, 2>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 128 * A0_LD * (int)sizeof(float)));
        LYS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(alpha0_n64_kernel<2, 0>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 128 * A0_LD * (int)sizeof(float)));
        attr_set[dev] = true;
    }
    const int64_t blocks = (N + 127) / 128;
    if (blocks > 0x7fffffffLL) {
        set_error("alpha0: grid too large");
        return LYS_ENOSUP;
    }
    const int64_t whole = N / 128, tail = N - whole * 128;
    if (whole)
        hipLaunchKernelGGL((alpha0_n64_kernel<2, 2>), dim3((unsigned)whole), dim3(256), lds, stream, X, ldx, D, ldd, C, Kp,
                           whole * 128, n);
    if (tail)
        hipLaunchKernelGGL((alpha0_n64_kernel<2, 0>), dim3(1), dim3(256), lds, stream, X + whole * 128 * ldx, ldx, D, ldd,
                           C + whole * 128 * Kp, Kp, tail, n);
    LYS_LAUNCH_CHECK();
    return LYS_OK;
}

// ---- small helpers living with the GEMM ---------------------------------------------------------
__global__ void pack_dictionary_kernel(const float* __restrict__ src, int n, int K, float* __restrict__ dst, int ldd,
                                       int Kp) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)Kp * ldd) return;
    const int a = (int)(t / ldd), f = (int)(t % ldd);
    dst[t] = (a < K && f < n) ? src[(int64_t)a * n + f] : 0.f;
}

int pack_dictionary(const float* src, int n, int K, float* dst, hipStream_t stream) {
    const int Kp = padded_atoms(K), ldd = padded_features(n);
    const int64_t tot = (int64_t)Kp * ldd;
    hipLaunchKernelGGL(pack_dictionary_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream, src, n, K,
                       dst, ldd, Kp);
    LYS_LAUNCH_CHECK();
    return LYS_OK;
}

// dst[f][a] = src[a][f]  (packed dictionary -> feature-major copy used by the online-DL update GEMM)
__global__ void transpose_kernel(const float* __restrict__ src, int rows, int cols, int lds_, float* __restrict__ dst,
                                 int ldd_) {
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    for (int i = threadIdx.y; i < 32; i += 8) {
        const int r = by + i, c = bx + threadIdx.x;
        tile[i][threadIdx.x] = (r < rows && c < cols) ? src[(int64_t)r * lds_ + c] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += 8) {
        const int r = bx + i, c = by + threadIdx.x;  // dst row = src col
        if (r < cols && c < rows) dst[(int64_t)r * ldd_ + c] = tile[threadIdx.x][i];
    }
}

int transpose(const float* src, int rows, int cols, int ld_src, float* dst, int ld_dst, hipStream_t stream) {
    dim3 grid((cols + 31) / 32, (rows + 31) / 32), block(32, 8);
    hipLaunchKernelGGL(transpose_kernel, grid, block, 0, stream, src, rows, cols, ld_src, dst, ld_dst);
    LYS_LAUNCH_CHECK();
    return LYS_OK;
}

}  // namespace lys
