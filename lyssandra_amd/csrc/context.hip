// Library-owned context (SURVEY 8b's proposed ABI): a caller with nothing but host arrays -- no PyTorch, no HIP calls of
// its own -- can set a dictionary, encode signals and read the stage timings.  The context owns the device buffers
// (packed dictionary, Gram matrix, alpha0 workspace, pinned-free staging of the signal tile and the sparse result) and
// one stream; every call is synchronous on return.  Also here: the counter-based synthetic signal generator of SURVEY
// 8(d) (Philox4x32-10 + Box-Muller keyed by (seed, global signal index, feature block)), so that any shard of a
// benchmark regenerates the same patches on any device -- and, through oracle/bomp_oracle.c's identical generator, on
// the host for the CPU leg.
#include <math.h>
#include <stdlib.h>

#include "common.h"
#include "../../include/lyssa_hip.h"

namespace lys {

// ------------------------------------------------------------------------------------------------ Philox4x32-10
__host__ __device__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t out[4]) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1,
                       n3 = (uint32_t)p0;
        c0 = n0;
        c1 = n1;
        c2 = n2;
        c3 = n3;
        k0 += W0;
        k1 += W1;
    }
    out[0] = c0;
    out[1] = c1;
    out[2] = c2;
    out[3] = c3;
}

// Features 4b..4b+3 of signal i: two Box-Muller pairs from one Philox block, evaluated in double and rounded to fp32 (the
// double evaluation is what makes host and device agree: a <= 2 ulp(double) difference of the math libraries survives
// the rounding to fp32 in about one value per 10^8).
__host__ __device__ inline void synth_block(uint64_t seed, uint64_t i, uint32_t b, float z[4]) {
    uint32_t r[4];
    philox4x32_10((uint32_t)i, (uint32_t)(i >> 32), b, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    const double two32 = 1.0 / 4294967296.0, twopi = 6.283185307179586476925286766559;
    for (int h = 0; h < 2; ++h) {
        const double u1 = ((double)r[2 * h] + 0.5) * two32, u2 = ((double)r[2 * h + 1] + 0.5) * two32;
        const double rad = sqrt(-2.0 * log(u1));
        z[2 * h] = (float)(rad * cos(twopi * u2));
        z[2 * h + 1] = (float)(rad * sin(twopi * u2));
    }
}

__global__ __launch_bounds__(256) void synth_signals_kernel(uint64_t seed, int64_t first, int64_t N, int n,
                                                            float* __restrict__ X, int64_t ldx) {
    const int nb = (n + 3) >> 2;
    const int64_t tot = N * nb;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < tot; t += (int64_t)gridDim.x * 256) {
        const int64_t i = t / nb;
        const int b = (int)(t - i * nb);
        float z[4];
        synth_block(seed, (uint64_t)(first + i), (uint32_t)b, z);
        for (int e = 0; e < 4; ++e)
            if (4 * b + e < n) X[i * ldx + 4 * b + e] = z[e];
    }
}

int synth_signals(uint64_t seed, int64_t first, int64_t N, int n, float* X, int64_t ldx, hipStream_t stream) {
    if (N <= 0) return LYS_OK;
    const int64_t tot = N * ((n + 3) / 4);
    int64_t blocks = (tot + 255) / 256;
    const int64_t cap = (int64_t)num_cus() * 32;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(synth_signals_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, seed, first, N, n, X, ldx);
    LYS_LAUNCH_CHECK();
    return LYS_OK;
}

}  // namespace lys

// ------------------------------------------------------------------------------------------------ context
struct lys_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    int n = 0, K = 0, Kp = 0, ldd = 0;
    float *D = nullptr, *G = nullptr;
    void* ws = nullptr;
    size_t ws_bytes = 0;
    float* X = nullptr;  // device signal tile [tile][n]
    int32_t *idx = nullptr, *nnz = nullptr;
    float* coef = nullptr;
    int64_t tile = 0;
    int tile_k = 0;
    double ms[4] = {0, 0, 0, 0};  // last call: host->device, encode kernels, device->host, wall
};

using namespace lys;

#define CTX_HIP(call)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return LYS_EHIP;                                                                   \
        }                                                                                      \
    } while (0)

static void ctx_free_tiles(lys_ctx* c) {
    if (c->X) (void)hipFree(c->X);
    if (c->idx) (void)hipFree(c->idx);
    if (c->coef) (void)hipFree(c->coef);
    if (c->nnz) (void)hipFree(c->nnz);
    if (c->ws) (void)hipFree(c->ws);
    c->X = nullptr;
    c->idx = c->nnz = nullptr;
    c->coef = nullptr;
    c->ws = nullptr;
    c->tile = 0;
    c->tile_k = 0;
    c->ws_bytes = 0;
}

extern "C" {

int lys_synth_signals(uint64_t seed, int64_t first, int64_t N, int n, float* X, int64_t ldx, void* stream) {
    if (!X || n <= 0 || N < 0 || ldx < n) {
        set_error("synth_signals: bad arguments");
        return LYS_EINVAL;
    }
    return synth_signals(seed, first, N, n, X, ldx, reinterpret_cast<hipStream_t>(stream));
}

int lys_ctx_create(int device, lys_ctx** out) {
    if (!out) {
        set_error("ctx_create: null pointer");
        return LYS_EINVAL;
    }
    *out = nullptr;
    CTX_HIP(hipSetDevice(device));
    lys_ctx* c = new (std::nothrow) lys_ctx();
    if (!c) {
        set_error("ctx_create: out of host memory");
        return LYS_EINVAL;
    }
    c->device = device;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    for (int i = 0; i < 4 && e == hipSuccess; ++i) e = hipEventCreate(&c->ev[i]);
    if (e != hipSuccess) {
        set_error("ctx_create: %s", hipGetErrorString(e));
        lys_ctx_destroy(c);
        return LYS_EHIP;
    }
    *out = c;
    return LYS_OK;
}

void lys_ctx_destroy(lys_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    ctx_free_tiles(c);
    if (c->D) (void)hipFree(c->D);
    if (c->G) (void)hipFree(c->G);
    for (int i = 0; i < 4; ++i)
        if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int lys_ctx_set_dictionary(lys_ctx* c, const float* D_atom_major_host, int n, int K) {
    if (!c || !D_atom_major_host || n <= 0 || K <= 0) {
        set_error("ctx_set_dictionary: bad arguments");
        return LYS_EINVAL;
    }
    CTX_HIP(hipSetDevice(c->device));
    const int Kp = lys_padded_atoms(K), ldd = lys_padded_features(n);
    if (Kp != c->Kp || ldd != c->ldd) {
        if (c->D) (void)hipFree(c->D);
        if (c->G) (void)hipFree(c->G);
        c->D = c->G = nullptr;
        ctx_free_tiles(c);
        CTX_HIP(hipMalloc(reinterpret_cast<void**>(&c->D), (size_t)Kp * ldd * sizeof(float)));
        CTX_HIP(hipMalloc(reinterpret_cast<void**>(&c->G), (size_t)Kp * Kp * sizeof(float)));
    }
    c->n = n;
    c->K = K;
    c->Kp = Kp;
    c->ldd = ldd;
    float* tmp = nullptr;
    CTX_HIP(hipMalloc(reinterpret_cast<void**>(&tmp), (size_t)K * n * sizeof(float)));
    hipError_t e = hipMemcpyAsync(tmp, D_atom_major_host, (size_t)K * n * sizeof(float), hipMemcpyHostToDevice, c->stream);
    int rc = LYS_OK;
    if (e == hipSuccess) {
        rc = lys_pack_dictionary(tmp, n, K, c->D, c->stream);
        if (!rc) rc = lys_gram(c->D, n, K, c->G, c->stream);
        e = hipStreamSynchronize(c->stream);
    }
    (void)hipFree(tmp);
    if (e != hipSuccess) {
        set_error("ctx_set_dictionary: %s", hipGetErrorString(e));
        return LYS_EHIP;
    }
    return rc;
}

// device buffers for tiles of `tile` signals with k coefficient slots
static int ctx_reserve(lys_ctx* c, int64_t tile, int k) {
    if (c->tile >= tile && c->tile_k >= k) return LYS_OK;
    ctx_free_tiles(c);
    CTX_HIP(hipMalloc(reinterpret_cast<void**>(&c->X), (size_t)tile * c->n * sizeof(float)));
    CTX_HIP(hipMalloc(reinterpret_cast<void**>(&c->idx), (size_t)tile * k * sizeof(int32_t)));
    CTX_HIP(hipMalloc(reinterpret_cast<void**>(&c->coef), (size_t)tile * k * sizeof(float)));
    CTX_HIP(hipMalloc(reinterpret_cast<void**>(&c->nnz), (size_t)tile * sizeof(int32_t)));
    c->ws_bytes = lys_bomp_workspace_bytes(c->n, c->K, k, tile);
    CTX_HIP(hipMalloc(&c->ws, c->ws_bytes));
    c->tile = tile;
    c->tile_k = k;
    return LYS_OK;
}

static int64_t ctx_tile(const lys_ctx* c, int64_t N) {
    const int64_t pref = ((int64_t)4 << 30) / ((int64_t)c->Kp * 4);  // one alpha0 tile of the engine (4 GiB)
    return N < pref ? (N < 1 ? 1 : N) : pref;
}

int lys_ctx_bomp_encode(lys_ctx* c, const float* X_sig_major_host, int64_t N, int k, int32_t* idx_host, float* coef_host,
                        int32_t* nnz_host) {
    if (!c || !c->D || (N > 0 && (!X_sig_major_host || !idx_host || !coef_host || !nnz_host)) || N < 0 || k < 1 || k > 64) {
        set_error("ctx_bomp_encode: bad arguments (dictionary set? 1 <= k <= 64?)");
        return LYS_EINVAL;
    }
    CTX_HIP(hipSetDevice(c->device));
    c->ms[0] = c->ms[1] = c->ms[2] = c->ms[3] = 0.0;
    if (N == 0) return LYS_OK;
    const int64_t tile = ctx_tile(c, N);
    int rc = ctx_reserve(c, tile, k);
    if (rc) return rc;
    for (int64_t s0 = 0; s0 < N; s0 += tile) {
        const int64_t cnt = (N - s0 < tile) ? N - s0 : tile;
        CTX_HIP(hipEventRecord(c->ev[0], c->stream));
        CTX_HIP(hipMemcpyAsync(c->X, X_sig_major_host + s0 * c->n, (size_t)cnt * c->n * sizeof(float), hipMemcpyHostToDevice,
                               c->stream));
        CTX_HIP(hipEventRecord(c->ev[1], c->stream));
        rc = lys_bomp_encode(c->X, c->n, c->D, c->G, c->n, c->K, k, cnt, c->idx, c->coef, c->nnz, c->ws, c->ws_bytes, c->stream);
        if (rc) return rc;
        CTX_HIP(hipEventRecord(c->ev[2], c->stream));
        CTX_HIP(hipMemcpyAsync(idx_host + s0 * k, c->idx, (size_t)cnt * k * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
        CTX_HIP(hipMemcpyAsync(coef_host + s0 * k, c->coef, (size_t)cnt * k * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        CTX_HIP(hipMemcpyAsync(nnz_host + s0, c->nnz, (size_t)cnt * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
        CTX_HIP(hipEventRecord(c->ev[3], c->stream));
        CTX_HIP(hipStreamSynchronize(c->stream));
        float a = 0.f, b = 0.f, d = 0.f;
        CTX_HIP(hipEventElapsedTime(&a, c->ev[0], c->ev[1]));
        CTX_HIP(hipEventElapsedTime(&b, c->ev[1], c->ev[2]));
        CTX_HIP(hipEventElapsedTime(&d, c->ev[2], c->ev[3]));
        c->ms[0] += a;
        c->ms[1] += b;
        c->ms[2] += d;
        c->ms[3] += a + b + d;
    }
    return LYS_OK;
}

int lys_ctx_bomp_encode_synthetic(lys_ctx* c, uint64_t seed, int64_t first, int64_t N, int k, double* stats4) {
    if (!c || !c->D || N < 0 || k < 1 || k > 64 || !stats4) {
        set_error("ctx_bomp_encode_synthetic: bad arguments");
        return LYS_EINVAL;
    }
    CTX_HIP(hipSetDevice(c->device));
    c->ms[0] = c->ms[1] = c->ms[2] = c->ms[3] = 0.0;
    stats4[0] = stats4[1] = stats4[2] = stats4[3] = 0.0;
    if (N == 0) return LYS_OK;
    const int64_t tile = ctx_tile(c, N);
    int rc = ctx_reserve(c, tile, k);
    if (rc) return rc;
    int32_t* hn = static_cast<int32_t*>(malloc((size_t)tile * sizeof(int32_t)));
    if (!hn) {
        set_error("ctx_bomp_encode_synthetic: out of host memory");
        return LYS_EINVAL;
    }
    double nnz_sum = 0.0;
    for (int64_t s0 = 0; s0 < N && !rc; s0 += tile) {
        const int64_t cnt = (N - s0 < tile) ? N - s0 : tile;
        hipError_t e = hipEventRecord(c->ev[0], c->stream);
        rc = synth_signals(seed, first + s0, cnt, c->n, c->X, c->n, c->stream);
        if (e == hipSuccess) e = hipEventRecord(c->ev[1], c->stream);
        if (!rc)
            rc = lys_bomp_encode(c->X, c->n, c->D, c->G, c->n, c->K, k, cnt, c->idx, c->coef, c->nnz, c->ws, c->ws_bytes,
                                 c->stream);
        if (e == hipSuccess) e = hipEventRecord(c->ev[2], c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(hn, c->nnz, (size_t)cnt * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        float a = 0.f, b = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&a, c->ev[0], c->ev[1]);
        if (e == hipSuccess) e = hipEventElapsedTime(&b, c->ev[1], c->ev[2]);
        if (e != hipSuccess) {
            set_error("ctx_bomp_encode_synthetic: %s", hipGetErrorString(e));
            rc = LYS_EHIP;
            break;
        }
        c->ms[0] += a;
        c->ms[1] += b;
        c->ms[3] += a + b;
        for (int64_t i = 0; i < cnt; ++i) nnz_sum += hn[i];
    }
    free(hn);
    if (rc) return rc;
    stats4[0] = (double)N;
    stats4[1] = nnz_sum / (double)N;              // mean number of selected atoms
    stats4[2] = c->ms[1];                         // encode kernels, ms
    stats4[3] = (double)N / (c->ms[1] * 1e-3);    // patches per second, inputs resident
    return LYS_OK;
}

int lys_ctx_timings(const lys_ctx* c, double* ms4) {
    if (!c || !ms4) {
        set_error("ctx_timings: null pointer");
        return LYS_EINVAL;
    }
    for (int i = 0; i < 4; ++i) ms4[i] = c->ms[i];
    return LYS_OK;
}

}  // extern "C"
