// Shared helpers for the gfx950 kernels of libgaot_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/gaot_hip.h"

namespace gaot {

void set_error(const char* fmt, ...);

#define GAOT_REQUIRE(cond, ...)                      \
    do {                                             \
        if (!(cond)) {                               \
            ::gaot::set_error(__VA_ARGS__);          \
            return GAOT_ERR_BAD_ARG;                 \
        }                                            \
    } while (0)

#define GAOT_CHECK_LAUNCH(name)                                                  \
    do {                                                                         \
        hipError_t e__ = hipGetLastError();                                      \
        if (e__ != hipSuccess) {                                                 \
            ::gaot::set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return GAOT_ERR_LAUNCH;                                              \
        }                                                                        \
    } while (0)

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// 32x32 MFMA C/D fragment: lane holds column (lane & 31) and rows crow(r, lane >> 5), r = 0..15.
__device__ __forceinline__ int crow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

}  // namespace gaot
