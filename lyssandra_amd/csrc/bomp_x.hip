// Experimental instantiations of the second-generation greedy kernel (timing A/B through lys_debug_bomp_variant >= 100).
#include "bomp_wave2.h"

namespace lys {

#define W2_LAUNCH(WPS, NLDS, NV)                                                                                        \
    hipLaunchKernelGGL((w2::bomp_wave2_kernel<16, 10, WPS, NLDS, NV, true>), grid, block, lds_bytes, stream, alpha0, G, N, \
                       k, idx, coef, nnz, 1)

int bomp_x_variant(const float* alpha0, const float* G, int64_t N, int k, int32_t* idx, float* coef, int32_t* nnz,
                   int variant, int lds_bytes, hipStream_t stream) {
    const dim3 grid((unsigned)((N + 3) / 4)), block(256);
    if (k != 10) {
        set_error("bomp_x: k must be 10");
        return LYS_EINVAL;
    }
    switch (variant) {
        case 100: W2_LAUNCH(3, 2, 0); break;
        case 101: W2_LAUNCH(3, 2, 1); break;
        case 106: W2_LAUNCH(3, 3, 1); break;
        case 107: W2_LAUNCH(3, 3, 0); break;
        case 150: hipLaunchKernelGGL((w2::bomp_wave2_kernel<16, 10, 3, 2, 0, true, 4, true>), grid, block, lds_bytes, stream, alpha0, G, N, k, idx, coef, nnz, 1); break;
        default: set_error("unknown variant %d", variant); return LYS_EINVAL;
    }
    LYS_LAUNCH_CHECK();
    return LYS_OK;
}

}  // namespace lys
