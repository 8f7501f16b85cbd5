// Stand-alone probe: HBM write bandwidth of the alpha0 GEMM's store pattern without any compute.
// hipcc --offload-arch=gfx950 -O3 tools/store_pattern_probe.hip -o gpurun_out/store_probe && gpurun_out/store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int KP = 1024;

// mode 0: linear fill.  mode 1: 128-row x 512-B segments, 8 atom steps per workgroup (the GEMM's pattern), nt.
// mode 2: same, plain stores.  mode 3: 128 rows x full 4-KB rows (one step), nt.  mode 4: dword stores in 128-B segments.
// mode 5: like 1 with `delay` s_sleep between steps (emulating the MFMA phase)
template <int MODE>
__global__ __launch_bounds__(256) void k(float* C, int64_t N, int delay) {
    const int tid = threadIdx.x;
    const f4 v = {1.f, 2.f, 3.f, (float)tid};
    if (MODE == 0) {
        const int64_t total4 = N * KP / 4;
        for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < total4; i += (int64_t)gridDim.x * 256)
            __builtin_nontemporal_store(v, reinterpret_cast<f4*>(C) + i);
        return;
    }
    for (int64_t bm = (int64_t)blockIdx.x * 128; bm < N; bm += (int64_t)gridDim.x * 128) {
        if (MODE == 3) {
            for (int t = 0; t < 128; ++t) {
                const int vv = tid + 256 * t;
                const int row = vv >> 8, c4 = (vv & 255) * 4;
                __builtin_nontemporal_store(v, reinterpret_cast<f4*>(C + (bm + row) * KP + c4));
            }
            continue;
        }
        for (int bn = 0; bn < KP; bn += 128) {
            if (MODE == 4) {
                const int w = tid >> 6, l = tid & 63, l31 = l & 31, h = l >> 5;
                for (int i = 0; i < 2; ++i)
                    for (int j = 0; j < 2; ++j)
                        for (int r = 0; r < 16; ++r) {
                            const int64_t row = bm + (w >> 1) * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                            __builtin_nontemporal_store(v.x, &C[row * KP + bn + (w & 1) * 64 + j * 32 + l31]);
                        }
            } else {
                for (int t = 0; t < 16; ++t) {
                    const int vv = tid + 256 * t;
                    const int row = vv >> 5, c4 = (vv & 31) * 4;
                    f4* p = reinterpret_cast<f4*>(C + (bm + row) * KP + bn + c4);
                    if (MODE == 2) *p = v; else __builtin_nontemporal_store(v, p);
                }
            }
            if (MODE == 5) for (int d = 0; d < delay; ++d) __builtin_amdgcn_s_sleep(127);
        }
    }
}

template <int MODE>
void run(const char* name, float* C, int64_t N, int grid, int delay = 0) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    for (int it = 0; it < 6; ++it) {
        hipEventRecord(a, 0);
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, C, N, delay);
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (it && ms < best) best = ms;
    }
    printf("%-46s grid %5d  %.3f ms  %.2f TB/s\n", name, grid, best, (double)N * KP * 4 / best / 1e9);
}

int main() {
    const int64_t N = 262144;
    float* C; hipMalloc(&C, N * KP * 4);
    run<0>("linear fill nt dwordx4", C, N, 2048);
    for (int grid : {512, 2048}) {
        run<1>("128 rows x 512 B, 8 steps, nt dwordx4", C, N, grid);
        run<2>("128 rows x 512 B, 8 steps, plain dwordx4", C, N, grid);
        run<3>("128 rows x 4 KB, nt dwordx4", C, N, grid);
        run<4>("mfma-layout dword nt (128-B segments)", C, N, grid);
    }
    for (int d : {1, 2, 4, 8}) { char nm[64]; snprintf(nm, 64, "pattern 1 + %d x s_sleep(127) per step", d); run<5>(nm, C, N, 512, d); }
    return 0;
}
