// Instantiations and launch of the second-generation single-wave Batch-OMP kernel (bomp_wave2.h) for the shapes it
// serves: Kp in {256, 512, 1024} (R = 4, 8, 16 correlations per lane), k <= 10.  Everything else stays with the kernels
// of bomp.hip.
#include "bomp_wave2.h"

namespace lys {

// (vectors in LDS, unstored tail vectors, waves per SIMD the register budget is bounded for) per (R, KMAX).
// Round 5 (register vectors packed into 16-wide ones, bomp_wave2.h `State`): K = 512, k = 10 2.43 -> 1.61-1.68 ms per 2^20
// signals, K = 512, k = 5 0.91 (first kernel) -> 0.72 ms (now served here), K = 256 0.61 -> 0.59 (k = 5) and 1.33 -> 1.19-1.22 ms
// (k = 10); several signals per wave one after the other (2 / 4 / 8, next row prefetched): no change -- not launch bound.
// K = 1024, k = 10: 2 vectors in LDS + the last one never stored -> 162 VGPRs, 3 waves per SIMD, 35 KB LDS per 4-wave
// workgroup (measured, tools/omp_ab2.py: 3 in LDS 2 % slower -- every LDS vector costs 4 ds_read_b128 of latency per step;
// 1 in LDS spills).
template <int R, int KMAX>
struct W2Cfg {
    static constexpr int NLDS = (R == 16 && KMAX == 10) ? 2 : 0;
    static constexpr int NV = 1;
    static constexpr int WPS = (R == 16) ? (KMAX == 10 ? 3 : 4) : (R == 8) ? (KMAX == 10 ? 4 : 6) : (KMAX == 10 ? 6 : 8);
};

template <int R, int KMAX>
static int launch_w2(const float* alpha0, const float* G, int64_t N, int k, int32_t* idx, float* coef, int32_t* nnz,
                     hipStream_t stream, int unit_diag) {
    using Cf = W2Cfg<R, KMAX>;
    const int64_t blocks = (N + 3) / 4;
    if (blocks > 0x7fffffffLL) {
        set_error("bomp: too many signals per launch (%lld)", (long long)N);
        return LYS_ENOSUP;
    }
    const dim3 grid((unsigned)blocks), block(256);
    if (k == KMAX && unit_diag)  // k and the unit Gram diagonal known at compile time: the 'bomp' launches
        hipLaunchKernelGGL((w2::bomp_wave2_kernel<R, KMAX, Cf::WPS, Cf::NLDS, Cf::NV, true>), grid, block, 0, stream, alpha0,
                           G, N, k, idx, coef, nnz, unit_diag);
    else
        hipLaunchKernelGGL((w2::bomp_wave2_kernel<R, KMAX, Cf::WPS, Cf::NLDS, Cf::NV, false>), grid, block, 0, stream, alpha0,
                           G, N, k, idx, coef, nnz, unit_diag);
    LYS_LAUNCH_CHECK();
    return LYS_OK;
}

// returns 1 when (Kp, k) is not served here
int bomp_wave2_launch(int Kp, const float* alpha0, const float* G, int64_t N, int k, int32_t* idx, float* coef,
                      int32_t* nnz, hipStream_t stream, int unit_diag) {
    if (k < 1 || k > 10) return 1;
    switch (Kp) {
        case 256: return k <= 5 ? launch_w2<4, 5>(alpha0, G, N, k, idx, coef, nnz, stream, unit_diag)
                                : launch_w2<4, 10>(alpha0, G, N, k, idx, coef, nnz, stream, unit_diag);
        case 512: return k <= 5 ? launch_w2<8, 5>(alpha0, G, N, k, idx, coef, nnz, stream, unit_diag)
                                : launch_w2<8, 10>(alpha0, G, N, k, idx, coef, nnz, stream, unit_diag);
        case 1024: return k <= 5 ? launch_w2<16, 5>(alpha0, G, N, k, idx, coef, nnz, stream, unit_diag)
                                 : launch_w2<16, 10>(alpha0, G, N, k, idx, coef, nnz, stream, unit_diag);
        default: return 1;
    }
}

}  // namespace lys
