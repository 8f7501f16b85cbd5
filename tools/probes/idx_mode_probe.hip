// Cost of one VGPR-index-mode read (s_set_gpr_idx_on / v_mov / s_set_gpr_idx_off) on a wave's dependent chain, against a
// 15-select tree over the same 16 registers.  hipcc --offload-arch=gfx950 -O3 tools/probes/idx_mode_probe.hip -o /tmp/idx_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ void probe(float* out, unsigned long long* cyc, int iters, int seed) {
    v16 v;
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = (float)(threadIdx.x * 16 + i + seed);
    int r = seed & 15;
    float acc = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        float x;
        if constexpr (MODE == 0) {
            x = v[r];  // index mode
        } else if constexpr (MODE == 1) {
            x = v[15];
#pragma unroll
            for (int i = 14; i >= 0; --i) x = (r == i) ? v[i] : x;  // select chain
        } else {
            x = v[3];  // static
        }
        acc += x;
        // next index depends on the value read (wave-uniform through readfirstlane): a true dependent chain
        r = (__builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, acc)) >> 3) & 15;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 1 << 16);
    const int iters = 4096;
    for (int waves = 1; waves <= 8; waves *= 2) {
        for (int mode = 0; mode < 3; ++mode) {
            // one workgroup of `waves`*4 waves on one CU: waves per SIMD = waves
            dim3 g(1), b(64 * 4 * waves);
            if (b.x > 1024) { g.x = b.x / 1024; b.x = 1024; }
            for (int rep = 0; rep < 2; ++rep) {
                if (mode == 0) hipLaunchKernelGGL(probe<0>, g, b, 0, 0, out, cyc, iters, 5);
                if (mode == 1) hipLaunchKernelGGL(probe<1>, g, b, 0, 0, out, cyc, iters, 5);
                if (mode == 2) hipLaunchKernelGGL(probe<2>, g, b, 0, 0, out, cyc, iters, 5);
                hipDeviceSynchronize();
            }
            unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            printf("waves/SIMD~%d mode %d (%s): %.1f s_memtime ticks per iteration\n", waves, mode,
                   mode == 0 ? "index mode" : mode == 1 ? "15 selects" : "static", (double)c / iters);
        }
    }
    return 0;
}
