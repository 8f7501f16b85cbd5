// gfx950: v_permlane16_swap / v_permlane32_swap as a four-row sum (the narrow step's cross-team reduction in ksvd_block.hip).
// Every lane must end with x[l % 16] + x[16 + l % 16] + x[32 + l % 16] + x[48 + l % 16], summed as ((r0 + r1) + (r2 + r3)).
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 tools/probes/permlane_swap_probe.hip -o /tmp/pls && /tmp/pls
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ float xrow_sum(float x) {
    float a = x, b = x;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    const float s = a + b;
    float c = s, d = s;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(c), "+v"(d));
    return c + d;
}
__global__ void k(const float* in, float* out) { out[threadIdx.x] = xrow_sum(in[threadIdx.x]); }
int main() {
    float h[64], r[64], *di, *dout;
    for (int i = 0; i < 64; ++i) h[i] = 1.0f + 0.37f * i + 1e-3f * i * i;
    hipMalloc(&di, 256); hipMalloc(&dout, 256);
    hipMemcpy(di, h, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout);
    hipMemcpy(r, dout, 256, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) {
        const int q = i & 15;
        const float want = (h[q] + h[16 + q]) + (h[32 + q] + h[48 + q]);
        if (r[i] != want) { ++bad; printf("lane %d: got %.9g want %.9g\n", i, r[i], want); }
    }
    printf("permlane swap four-row sum: %s\n", bad ? "MISMATCH" : "ok (bit-exact in all 64 lanes)");
    return bad != 0;
}
