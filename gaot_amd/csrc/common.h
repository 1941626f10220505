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

// erff of the ROCm device library (ocml erfF: two polynomial ranges split at |x| = 1) with BOTH ranges evaluated and the
// result selected: same operations per range, so the value equals erff(x) bit for bit, but there is no branch -- a
// fragment's 16-32 GELUs stay one basic block (the scheduler can interleave them with MFMAs; no exec-mask juggling).
__device__ __forceinline__ float erf_nb(float x) {
    const float ax = fabsf(x), t = x * x;
    float p = fmaf(t, -0x1.268bc2p-11f, 0x1.420828p-8f);
    p = fmaf(t, p, -0x1.b5937p-6f);
    p = fmaf(t, p, 0x1.ce077cp-4f);
    p = fmaf(t, p, -0x1.81266p-2f);
    p = fmaf(t, p, 0x1.06ebap-3f);
    const float small = fmaf(ax, p, ax);
    float q = fmaf(ax, 0x1.1d3156p-16f, -0x1.8d129p-12f);
    q = fmaf(ax, q, 0x1.f9a6d2p-9f);
    q = fmaf(ax, q, -0x1.8c3164p-6f);
    q = fmaf(ax, q, 0x1.b4e9c8p-4f);
    q = fmaf(ax, q, 0x1.4515fap-1f);
    q = fmaf(ax, q, 0x1.078e5p-3f);
    const float large = 1.0f - expf(-fmaf(ax, q, ax));
    return copysignf(ax < 1.0f ? small : large, x);
}

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erf_nb(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    const float cdf = 0.5f * (1.0f + erf_nb(x * 0.70710678118654752440f));
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
