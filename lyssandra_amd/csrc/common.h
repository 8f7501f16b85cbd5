// Shared host/device helpers of the gfx950 sparse-coding engine (wave = 64 lanes, CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/lyssa_hip.h"

namespace lys {

void set_error(const char* fmt, ...);

#define LYS_CHECK_HIP(expr)                                                              \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            lys::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return LYS_EHIP;                                                             \
        }                                                                                \
    } while (0)

#define LYS_REQUIRE(cond, ...)              \
    do {                                    \
        if (!(cond)) {                      \
            lys::set_error(__VA_ARGS__);    \
            return LYS_EINVAL;              \
        }                                   \
    } while (0)

#define LYS_LAUNCH_CHECK()                                                               \
    do {                                                                                 \
        hipError_t _e = hipGetLastError();                                               \
        if (_e != hipSuccess) {                                                          \
            lys::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
            return LYS_EHIP;                                                             \
        }                                                                                \
    } while (0)

constexpr float EPS64_F = 2.220446049250313e-16f;  // np.finfo(float).eps, the reference's +eps (utils/math.py:61,65)
constexpr float EPS32_F = 1.1920929e-07f;

inline int padded_atoms(int K) {
    if (K <= 1024) {
        int p = 64;
        while (p < K) p <<= 1;
        return p;
    }
    return ((K + 2047) / 2048) * 2048;  // workgroup-per-signal kernels: 512 threads x dwordx4
}
inline int padded_features(int n) { return ((n + 7) / 8) * 8; }

int num_cus();  // CUs of the current device (cached)
int64_t tile_signals(int Kp);  // signals per alpha0 tile of the encode drivers (api.hip)

// ------------------------------------------------------------------ device helpers
#if defined(__HIPCC__)
// DPP cross-lane move: lanes whose source is invalid keep `x` (old = x, bound_ctrl = false).
template <int CTRL>
__device__ __forceinline__ float dpp_f(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, x), __builtin_bit_cast(int, x),
                                                                 CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int x) {
    return __builtin_amdgcn_update_dpp(x, x, CTRL, 0xf, 0xf, false);
}
// DPP move with 0 for lanes whose source is invalid (old = 0, bound_ctrl).
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u0(unsigned x) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xf, 0xf, true);
}
// Wave-wide max of a NON-NEGATIVE float, result uniform (SGPR) -- 4 row steps + 2 row broadcasts, no LDS.
// Done on the bit patterns (for x >= 0 the unsigned order equals the float order): v_max_u32 needs no NaN
// canonicalisation and takes the DPP source directly (v_max_u32_dpp), 1 VALU op per step instead of 3.
__device__ __forceinline__ float wave_max_f(float x) {
    unsigned u = __builtin_bit_cast(unsigned, x);
    u = max(u, dpp_u0<0xB1>(u));   // quad_perm [1,0,3,2]
    u = max(u, dpp_u0<0x4E>(u));   // quad_perm [2,3,0,1]
    u = max(u, dpp_u0<0x124>(u));  // row_ror:4
    u = max(u, dpp_u0<0x128>(u));  // row_ror:8  -> every lane of a 16-lane row holds the row max
    u = max(u, dpp_u0<0x142>(u));  // row_bcast:15
    u = max(u, dpp_u0<0x143>(u));  // row_bcast:31 -> lane 63 holds the wave max
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane((int)u, 63));
}
__device__ __forceinline__ int wave_min_i(int x) {
    x = min(x, dpp_i<0xB1>(x));
    x = min(x, dpp_i<0x4E>(x));
    x = min(x, dpp_i<0x124>(x));
    x = min(x, dpp_i<0x128>(x));
    x = min(x, dpp_i<0x142>(x));
    x = min(x, dpp_i<0x143>(x));
    return __builtin_amdgcn_readlane(x, 63);
}
// Wave-wide sum, every lane gets the result (butterfly through ds_bpermute-free DPP + readlane).
__device__ __forceinline__ float wave_sum_f(float x) {
    x += dpp_f<0xB1>(x);
    x += dpp_f<0x4E>(x);
    x += dpp_f<0x124>(x);
    x += dpp_f<0x128>(x);
    // rows now hold their own sums in every lane; add the 4 row sums through readlane (uniform)
    float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 0));
    float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 16));
    float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 32));
    float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 48));
    return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ float readlane_f(float x, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), lane));
}
// vec with lane `lane` replaced by the (wave-uniform) value `val`; needs the caller's lane id.
// (clang has no writelane builtin; a compare+select is one v_cndmask once `my_lane == lane` is hoisted.)
// vec[lane] = val for a wave-uniform val that already lives in an SGPR (a readlane / readfirstlane result) and a
// compile-time-constant lane: one v_writelane_b32 (clang has no builtin for it) instead of compare + select
__device__ __forceinline__ float writelane_sgpr(float vec, float val_sgpr, int lane_const) {
    const int sv = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, val_sgpr));  // folds away when already scalar
    asm("v_writelane_b32 %0, %1, %2" : "+v"(vec) : "s"(sv), "i"(lane_const));
    return vec;
}

__device__ __forceinline__ float writelane_f(float val, int lane, float vec, int my_lane) {
    return (my_lane == lane) ? val : vec;
}
__device__ __forceinline__ int writelane_i(int val, int lane, int vec, int my_lane) {
    return (my_lane == lane) ? val : vec;
}
#endif

}  // namespace lys
