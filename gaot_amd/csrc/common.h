// Shared helpers for the gfx950 kernels of libgaot_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/gaot_hip.h"
#include "../../include/gaot_hip_debug.h"

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

// erf(x) = sign(x) (1 - 2^(-|x| Q(|x|))), Q a degree-8 polynomial fitted (weighted least squares on Chebyshev nodes,
// weight = d erf / d Q) to -log2(erfc(x)) / x on (0, 4.2]: ONE range, no branch, 8 FMAs + v_exp_f32.  Max |error| against
// float64 erf over [0, 6] and 1e-8..1: 1.03e-7 (the library erff is ~6e-8 there; it costs ~3x the instructions and its
// two ranges are a divergent branch that splits a fragment's 16-32 GELUs into as many basic blocks).  GELU only ever
// uses 1 + erf, so the absolute error is the one that matters; tests/test_ops_gpu.py pins it.
__device__ __forceinline__ float erf_nb(float x) {
    const float ax = fminf(fabsf(x), 4.2f);
    float q = fmaf(ax, -0x1.87dddp-17f, 0x1.42346ap-13f);
    q = fmaf(ax, q, -0x1.be08dp-11f);
    q = fmaf(ax, q, 0x1.2acd2cp-9f);
    q = fmaf(ax, q, -0x1.7a3b26p-14f);
    q = fmaf(ax, q, -0x1.c62eecp-6f);
    q = fmaf(ax, q, 0x1.2fbb7cp-3f);
    q = fmaf(ax, q, 0x1.d63e2cp-1f);
    q = fmaf(ax, q, 0x1.a0be88p+0f);
    return copysignf(1.0f - __builtin_amdgcn_exp2f(-(ax * q)), x);
}

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erf_nb(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    const float cdf = 0.5f * (1.0f + erf_nb(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __builtin_amdgcn_exp2f(-0.72134752044448170368f * x * x);
    return cdf + x * pdf;
}

// ---- exact 3-way bf16 split of fp32 values (gemm_split.hip, attention.hip): x = x1 + x2 + x3, 8 + 8 + 8 significant bits
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// two elements at a time; returns the three packed bf16 pairs (element 0 in the low half).  Only the pieces that feed a
// subtraction are masked; packing is a byte permute.  The residual subtractions are SCALAR v_sub_f32 on purpose: packed-fp32
// instructions (v_pk_add/mul/fma_f32, v_pk_mov_b32) do not overlap with MFMAs on gfx950 -- one of them per MFMA serialises the
// matrix pipe with the whole vector stream (tools/pipe_overlap.hip: 7 fma + 1 pk_add per MFMA takes 56 ns where 8 fma take
// 39) -- while v_sub_f32 / v_and_b32 (VOP2) issue at twice the rate of three-operand instructions.  The asm keeps the SLP
// vectoriser from re-packing them.  CAUTION with inline-asm VALU: the compiler's hazard recogniser does not look inside it.  An asm
// instruction whose input was just written by v_exp_f32 (trans-use wait state) or by an MFMA reads garbage (seen: wrong row
// sums in an attention kernel).  Here every input is first read by a compiler-generated v_and_b32, which takes the wait.
__device__ __forceinline__ float sub_scalar(float a, float b) {
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
template <int ABL = 0>
__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned& ph, unsigned& pm, unsigned& pl) {
    if (ABL & 1) { ph = pm = pl = __float_as_uint(x0) ^ __float_as_uint(x1); return; }
    const unsigned h0 = __float_as_uint(x0) & 0xffff0000u, h1 = __float_as_uint(x1) & 0xffff0000u;
    const float r0 = sub_scalar(x0, __uint_as_float(h0)), r1 = sub_scalar(x1, __uint_as_float(h1));
    const unsigned m0 = __float_as_uint(r0) & 0xffff0000u, m1 = __float_as_uint(r1) & 0xffff0000u;
    const float s0 = sub_scalar(r0, __uint_as_float(m0)), s1 = sub_scalar(r1, __uint_as_float(m1));      // <= 8 significant bits left: exact in bf16
    ph = __builtin_amdgcn_perm(h1, h0, 0x07060302u);
    pm = __builtin_amdgcn_perm(m1, m0, 0x07060302u);
    pl = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
}

// TWO-piece split for operands whose own rounding noise is far above 2^-18 (softmax probabilities, dS = P (dP - delta)): both
// pieces ROUNDED to nearest even by the hardware conversion (v_cvt_pk_bf16_f32, what __builtin_convertvector lowers to on gfx950):
// h = rne(x), r = x - h (exact: at most 16 significant bits survive), m = rne(r); x = h + m + e with |e| <= 2^-18 |x|, unbiased.
// Six vector instructions per pair instead of eleven, and one piece product fewer per MFMA group (x3 * y1 has no x3).
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2_pair(float x0, float x1, unsigned& ph, unsigned& pm) {
    const f32x2 x = {x0, x1};
    ph = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2_t));
    const f32x2 r = {sub_scalar(x0, __uint_as_float(ph << 16)), sub_scalar(x1, __uint_as_float(ph & 0xffff0000u))};
    pm = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2_t));
}

// TWO fp16 pieces of sc * x, both rounded to nearest even (v_cvt_pk_f16_f32): h = rn16(sc x), r = sc x - h (exact in fp32), m = rn16(r).
// An fp32 value has 24 significant bits; h takes 11 of them and |r| <= half an ulp of h, so r has at most 13 significant bits of which
// m keeps 11: sc x = h + m + e where e is AT MOST THE OPERAND'S LAST BIT -- |e| <= 2^-23 |sc x|, and e = 0 for three values in four
// (r fits m exactly unless it sits in the upper half of its range AND x's last bit is set; rms 0.4 x 2^-23: below one fp32 rounding of
// the product).  That holds while m stays a NORMAL fp16 number, i.e. |sc x| >= 2^-2; below that m is subnormal and the error is
// absolute, <= 2^-25.  With sc = the power of two that puts the operand's largest magnitude into [2^13, 2^14), every element within
// 2^-16 of the largest keeps that precision and smaller ones are off by at most 2^-39 of the largest -- and nothing overflows (fp16
// max 65 504).  Measured against float64 the three piece products h h + h m + m h are at or below the six-product three-piece bf16
// form on every product kind (tests/test_ops_gpu.py test_gemm_fp16_piece_products): fewer accumulation steps on the matrix pipe.
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ float mul_scalar(float a, float b) {
    float r;
    asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void split2h_pair(float x0, float x1, float sc, unsigned& ph, unsigned& pm) {
    const f32x2 x = {mul_scalar(x0, sc), mul_scalar(x1, sc)};
    const f16x2_t h = __builtin_convertvector(x, f16x2_t);
    ph = __builtin_bit_cast(unsigned, h);
    const f32x2 r = {sub_scalar(x[0], (float)h[0]), sub_scalar(x[1], (float)h[1])};
    pm = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2_t));
}

// [r6] the same two pieces with the second one from v_fma_mixlo_f16 / v_fma_mixhi_f16: m = rn16(y - (float)h) in ONE instruction per element
// (the fp16 operand is read in place, the difference is exact in fp32, one rounding to fp16) instead of v_cvt_f32_f16 + v_sub_f32 + half a
// v_cvt_pk_f16_f32: five instructions per pair instead of eight, bit-identical (tools/probe/mix_split.hip: 10^7 pairs, five scales, zeros,
// subnormals, overflow).  The mix instructions issue at half rate but, unlike the fp32 VOP2 ops, overlap with the matrix pipe
// (tools/pipe_overlap.hip).  split2h_pre: the operand arrives already scaled (a compiler-generated multiply did it).
__device__ __forceinline__ unsigned mix_residual(unsigned ph, float y0, float y1) {
    unsigned m = 0u;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(m) : "v"(ph), "v"(y0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(m) : "v"(ph), "v"(y1));
    return m;
}
__device__ __forceinline__ void split2h_pair_mix(float x0, float x1, float sc, unsigned& ph, unsigned& pm) {
    const f32x2 x = {mul_scalar(x0, sc), mul_scalar(x1, sc)};
    ph = __builtin_bit_cast(unsigned, __builtin_convertvector(x, f16x2_t));
    pm = mix_residual(ph, x[0], x[1]);
}
// what the tile GEMMs call (GAOT_GEMM_SPLIT_MIX = 0: the eight-instruction form, for A/B builds): bit-identical either way
#ifndef GAOT_GEMM_SPLIT_MIX
#define GAOT_GEMM_SPLIT_MIX 1
#endif
__device__ __forceinline__ void split2h_pair_gemm(float x0, float x1, float sc, unsigned& ph, unsigned& pm) {
    if (GAOT_GEMM_SPLIT_MIX) split2h_pair_mix(x0, x1, sc, ph, pm); else split2h_pair(x0, x1, sc, ph, pm);
}
__device__ __forceinline__ void split2h_pre(float y0, float y1, unsigned& ph, unsigned& pm) {
    const f32x2 x = {y0, y1};
    ph = __builtin_bit_cast(unsigned, __builtin_convertvector(x, f16x2_t));
    pm = mix_residual(ph, y0, y1);
}

// transposing LDS read (gfx950 ds_read_b64_tr_b16): within each 16-lane group, lane t supplies the address of 4 consecutive 16-bit
// COLUMNS of one row and receives 4 consecutive ROWS of one column: with lane t addressing row rbase + (t >> 2), columns
// cbase + 4 (t & 3) .. + 3 it gets rows rbase .. rbase + 3 of column cbase + t (tools/probe/tr_read.hip prints the map).  Two of them
// give an MFMA A / B fragment (8 k-slots) of a matrix stored with k as its ROW index.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ u32x2 lds_tr(const unsigned char* p) {
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p)));
}
__device__ __forceinline__ u32x4 join8(u32x2 lo, u32x2 hi) { return u32x4{lo[0], lo[1], hi[0], hi[1]}; }
// byte offset of a lane inside a [16 rows][32 columns] bf16 block read as an MFMA fragment by two lds_tr (rows + 0 and + 8): lane
// (li = lane & 31, hi = lane >> 5) receives column li, rows 4 hi .. + 3
__device__ __forceinline__ int lds_tr_lane_offset(int lane, int row_bytes) {
    return (4 * (lane >> 5) + ((lane & 15) >> 2)) * row_bytes + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
}

// ---- per-tensor magnitude words for the fp16-piece products.  A "word" is GAOT_AMAX_SLOTS = 32 slots, one float at the head of each
// 128-byte line (GAOT_AMAX_STRIDE = 32 floats): thousands of waves of a producer publish max |x| of what they stored by an atomic max
// on a float's bit pattern (non-negative floats order like unsigned integers), each into the slot its wave index selects.  Device-scope
// atomics on ONE cache line serialise at ~11 ns each (4 096 waves on one word: +30 us per launch, measured; on two lines: +11 us), so
// the slots sit on separate lines; a consumer takes the maximum of the 32 slots.  The slots are zero before the pass (ops._amax_words).
__device__ __forceinline__ void amax_publish(float* word, float v, int lane, int slot) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    unsigned* w = reinterpret_cast<unsigned*>(word) + (slot & (GAOT_AMAX_SLOTS - 1)) * GAOT_AMAX_STRIDE;
    // fire and forget: a non-returning atomic (nothing waits for it; a look-before-you-leap load would put its round trip at the end of
    // every workgroup's life: measured +20 us on a 1 024-workgroup launch)
    if (lane == 0 && v > 0.f) (void)__hip_atomic_fetch_max(w, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the same with ONE atomic per workgroup (kernels whose workgroups all finish together: thousands of per-wave atomics at the very end of
// a launch cost microseconds, 8 x fewer do not): every thread of the workgroup must call it; `red` = NW floats of LDS
template <int NW>
__device__ __forceinline__ void amax_publish_block(float* word, float v, float* red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = red[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) m = fmaxf(m, red[w]);
        if (m > 0.f) (void)__hip_atomic_fetch_max(reinterpret_cast<unsigned*>(word) + ((int)blockIdx.x & (GAOT_AMAX_SLOTS - 1)) * GAOT_AMAX_STRIDE,
                                                  __float_as_uint(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// (scale, 1 / scale) for an operand whose largest magnitude is the maximum of the word's slots: scale = 2^(13 - floor(log2 amax)),
// exponent clamped to +-126 (amax = 0 or denormal: the clamp; the operand is zero or flushes to it).  NaN / inf magnitudes give a finite
// scale: the product's NaNs come from the data itself.  Every lane of the wave must call it (cross-lane maximum).
// word2 (optional): a second word whose tensor shares the scale (the two A operands of a concatenated-input product)
__device__ __forceinline__ void amax_scale(const float* word, float& sc, float& inv, const float* word2 = nullptr) {
    unsigned b = __float_as_uint(word[(threadIdx.x & (GAOT_AMAX_SLOTS - 1)) * GAOT_AMAX_STRIDE]);
    if (word2 != nullptr) { const unsigned b2 = __float_as_uint(word2[(threadIdx.x & (GAOT_AMAX_SLOTS - 1)) * GAOT_AMAX_STRIDE]); b = b2 > b ? b2 : b; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const unsigned o = __shfl_xor(b, off, 64); b = o > b ? o : b; }
    const int e = (int)((b >> 23) & 0xffu);                               // biased exponent of amax
    int se = 140 - e;                                                     // 13 - (e - 127)
    se = se > 126 ? 126 : (se < -126 ? -126 : se);
    sc = __uint_as_float((unsigned)(127 + se) << 23);
    inv = __uint_as_float((unsigned)(127 - se) << 23);
}

// the same in two halves, for kernels that want their first operand loads in flight before they wait for the word: amax_peek() issues
// the word's load (one slot per lane), amax_finish() does the cross-lane maximum and forms the scales
__device__ __forceinline__ unsigned amax_peek(const float* word) {
    return __float_as_uint(word[(threadIdx.x & (GAOT_AMAX_SLOTS - 1)) * GAOT_AMAX_STRIDE]);
}
__device__ __forceinline__ void amax_finish(unsigned b, float& sc, float& inv) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const unsigned o = __shfl_xor(b, off, 64); b = o > b ? o : b; }
    const int e = (int)((b >> 23) & 0xffu);
    int se = 140 - e;
    se = se > 126 ? 126 : (se < -126 ? -126 : se);
    sc = __uint_as_float((unsigned)(127 + se) << 23);
    inv = __uint_as_float((unsigned)(127 - se) << 23);
}

// ---- cross-lane reductions on the DPP path (no LDS round trips: a ds_bpermute chain is six dependent ~100-cycle waits per 64-lane reduction)
template <int CTRL> __device__ __forceinline__ int dpp_mov(int x) { return __builtin_amdgcn_update_dpp(x, x, CTRL, 0xF, 0xF, true); }
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_ROW_MIRROR = 0x140;          // quad_perm [1,0,3,2], [2,3,0,1]
// maximum over the wave of NON-NEGATIVE floats (they order like their bit patterns), returned wave-uniform (scalar registers)
__device__ __forceinline__ float wave_max_nonneg(float v) {
    int x = __float_as_int(v);
    x = max(x, dpp_mov<DPP_XOR1>(x)); x = max(x, dpp_mov<DPP_XOR2>(x));
    x = max(x, dpp_mov<DPP_HALF_MIRROR>(x)); x = max(x, dpp_mov<DPP_ROW_MIRROR>(x));          // every lane: its row's (16 lanes) maximum
    const int a = __builtin_amdgcn_readlane(x, 0), b = __builtin_amdgcn_readlane(x, 16), c = __builtin_amdgcn_readlane(x, 32), d = __builtin_amdgcn_readlane(x, 48);
    return __int_as_float(max(max(a, b), max(c, d)));
}
// sum over each group of eight consecutive lanes, in every lane of the group: the same additions as the __shfl_xor 1, 2, 4 butterfly (bit-identical)
__device__ __forceinline__ float oct_sum(float v) {
    v += __int_as_float(dpp_mov<DPP_XOR1>(__float_as_int(v)));
    v += __int_as_float(dpp_mov<DPP_XOR2>(__float_as_int(v)));
    v += __int_as_float(dpp_mov<DPP_HALF_MIRROR>(__float_as_int(v)));
    return v;
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

// grid size for grid-stride kernels: ceil(n / per) workgroups, at least 1, at most cap
static inline int cap_blocks(long n, int per, int cap) { long b = (n + per - 1) / per; return (int)(b > cap ? cap : (b < 1 ? 1 : b)); }
