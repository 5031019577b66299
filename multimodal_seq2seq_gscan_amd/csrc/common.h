// Shared host/device helpers for libgscan_hip (gfx950 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <algorithm>

namespace gscan {

constexpr int kWave = 64;

void set_error(const char *fmt, ...);

#define GSCAN_CHECK(cond, ...)                                                      \
    do {                                                                            \
        if (!(cond)) { ::gscan::set_error(__VA_ARGS__); return 1; }                 \
    } while (0)

#define GSCAN_HIP(call)                                                             \
    do {                                                                            \
        hipError_t e_ = (call);                                                     \
        if (e_ != hipSuccess) {                                                     \
            ::gscan::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return 1;                                                               \
        }                                                                           \
    } while (0)

// Launch check: a bad configuration surfaces at launch, not at the next sync.
#define GSCAN_LAUNCHED(name)                                                        \
    do {                                                                            \
        hipError_t e_ = hipGetLastError();                                          \
        if (e_ != hipSuccess) {                                                     \
            ::gscan::set_error("launch of %s failed: %s", name, hipGetErrorString(e_)); \
            return 1;                                                               \
        }                                                                           \
    } while (0)

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

#if defined(__HIPCC__)
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// tanh through one exp and one division; |error| ~ 1e-7 relative, saturates cleanly.
__device__ __forceinline__ float tanhf_(float x) {
    float ax = fabsf(x);
    float e = __expf(-2.0f * ax);
    float t = (1.0f - e) / (1.0f + e);
    return copysignf(t, x);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
#endif

}  // namespace gscan
