// GemmArgs + the fused epilogue shared by the MFMA GEMM (gemm.hip) and the skinny VALU paths (skinny.hip).
#pragma once
#include "common.h"

namespace gaot {

struct GemmArgs {
    int M, N, K;
    const float* A; long lda; const float* A2; long lda2; int k_split;
    const float* B; long ldb;
    float* C; long ldc;
    const float* bias; const float* rowbias; int rb_period; long ld_rb;
    const float* rowscale; int act; const float* aux_in; float* aux_out; long ld_aux;
    const float* residual; long ldr;
    int split_k; int ktiles_per_split; float* ws;
    float* colsum;          // optional (A m-major only): colsum[m] = sum_k Aop[m,k]  (bias gradient fused into dW = dY^T X)
    int tiles_m, tiles_n;
    int vec_epi;            // 1: N, ldc and every epilogue operand allow 16-byte accesses -> LDS-transposed vector epilogue
    const float* a2_amax;   // with A2: max |A2| (the kernel scales both A operands by the larger of the two words)
    const float* a_amax; const float* b_amax;   // fp16-piece products (gemm_split.hip NP = 4): device words holding max |A|, max |B| (or a bound)
    const unsigned short* Bpl; long ld_bpl; long bpl_stride;   // optional (NP = 4): B pre-split into two k-contiguous fp16 planes of the scaled weight
    int bpl_flag;           // with Bpl: which float of b_amax's first line holds the planes' range verdict (1: along rows, 2: along columns of W as stored)
    float* c_amax;          // optional: max |C| as stored, accumulated by atomic max on the float's bit pattern (zero before the launch)
    int ablate;             // tuning only (gaot_debug_set_gemm_ablate): 1 = no in-loop global loads, 2 = no LDS staging/barriers, 4 = no stores
};

// SwiGLU gate and its gradient (attn.py:151): g = silu(u1) * u3
__device__ __forceinline__ float swiglu_f(float u1, float u3) { return (u1 / (1.0f + expf(-u1))) * u3; }
__device__ __forceinline__ void swiglu_grad_f(float dg, float u1, float u3, float& d1, float& d3) {
    const float sg = 1.0f / (1.0f + expf(-u1));
    d1 = dg * u3 * sg * (1.0f + u1 * (1.0f - sg));
    d3 = dg * u1 * sg;
}

__device__ __forceinline__ void epilogue_store(const GemmArgs& p, int m, int n, float v) {
    if (p.bias) v += p.bias[n];
    if (p.rowbias) v += p.rowbias[(long)(m % p.rb_period) * p.ld_rb + n];
    if (p.rowscale) v *= p.rowscale[m];
    if (p.aux_out) p.aux_out[(long)m * p.ld_aux + n] = v;
    switch (p.act) {
        case GAOT_ACT_GELU: v = gelu_f(v); break;
        case GAOT_ACT_RELU: v = fmaxf(v, 0.0f); break;
        case GAOT_ACT_GELU_BWD: v *= gelu_grad_f(p.aux_in[(long)m * p.ld_aux + n]); break;
        case GAOT_ACT_RELU_BWD: v = (p.aux_in[(long)m * p.ld_aux + n] > 0.0f) ? v : 0.0f; break;
        case GAOT_ACT_SWIGLU_BWD: {
            const float u1 = p.aux_in[(long)m * p.ld_aux + n], u3 = p.aux_in[(long)m * p.ld_aux + p.N + n];
            float d1, d3;
            swiglu_grad_f(v, u1, u3, d1, d3);
            p.C[(long)m * p.ldc + p.N + n] = d3;
            v = d1;
            break;
        }
        default: break;
    }
    if (p.residual) v += p.residual[(long)m * p.ldr + n];
    p.C[(long)m * p.ldc + n] = v;
}


// row-dependent part of the epilogue hoisted out of the per-element path (the row-bias modulo is an integer
// division; do it once per output row)
struct RowCtx {
    const float* rb;      // rowbias row or nullptr
    float rs;             // rowscale or 1
};
__device__ __forceinline__ RowCtx row_ctx(const GemmArgs& p, int m) {
    RowCtx c;
    c.rb = p.rowbias ? p.rowbias + (long)(m % p.rb_period) * p.ld_rb : nullptr;
    c.rs = p.rowscale ? p.rowscale[m] : 1.0f;
    return c;
}
__device__ __forceinline__ void epilogue_store_row(const GemmArgs& p, const RowCtx& c, int m, int n, float v) {
    if (p.bias) v += p.bias[n];
    if (c.rb) v += c.rb[n];
    v *= c.rs;
    if (p.aux_out) p.aux_out[(long)m * p.ld_aux + n] = v;
    switch (p.act) {
        case GAOT_ACT_GELU: v = gelu_f(v); break;
        case GAOT_ACT_RELU: v = fmaxf(v, 0.0f); break;
        case GAOT_ACT_GELU_BWD: v *= gelu_grad_f(p.aux_in[(long)m * p.ld_aux + n]); break;
        case GAOT_ACT_RELU_BWD: v = (p.aux_in[(long)m * p.ld_aux + n] > 0.0f) ? v : 0.0f; break;
        case GAOT_ACT_SWIGLU_BWD: {
            const float u1 = p.aux_in[(long)m * p.ld_aux + n], u3 = p.aux_in[(long)m * p.ld_aux + p.N + n];
            float d1, d3;
            swiglu_grad_f(v, u1, u3, d1, d3);
            p.C[(long)m * p.ldc + p.N + n] = d3;
            v = d1;
            break;
        }
        default: break;
    }
    if (p.residual) v += p.residual[(long)m * p.ldr + n];
    p.C[(long)m * p.ldc + n] = v;
}


// ---- vector epilogue shared by the GEMM kernels: each wave transposes its 32-row fragment band through a private LDS
// slab (the MFMA C-layout puts a COLUMN in a lane; stores want 4 consecutive columns per lane) and then does bias / row
// bias / row scale / activation / residual / store on 16-byte vectors: 4x fewer store instructions, 128-256 B per row.
// The caller must have synchronised the workgroup (smem is reused) and smem must hold NW * 32 * (WN + 4) floats.
template <int TM, int TN, int WM, int WN>
__device__ __forceinline__ void epilogue_vec(const GemmArgs& p, float* smem, const f32x16 (&acc)[TM][TN], int m0, int n0,
                                         int wm, int wn, int wave, int lane, int zs = -1, float so1 = 1.f, float so2 = 1.f) {
    float amax = 0.f;
    const int li = lane & 31, lh = lane >> 5;
    const int zslab = zs >= 0 ? zs : (int)blockIdx.z;          // K slab of a split-K product (grouped launches pass their own)
    // ---- vector epilogue: each wave transposes its 32-row fragment band through a private LDS slab (C-layout puts a
    // COLUMN in a lane; stores want 4 consecutive columns per lane) and then does bias / row bias / row scale /
    // activation / residual / store on 16-byte vectors: 4x fewer store instructions, 128-256 B contiguous per row.
    constexpr int LDC_S = WN + 4;
    constexpr int LPR = WN / 4;                        // lanes per output row
    constexpr int RPP = 64 / LPR;                      // rows per pass
    float* Cw = smem + wave * 32 * LDC_S;
    const int lr = lane / LPR, lc = (lane % LPR) * 4;
    const int n = n0 + wn * WN + lc;
    const bool nok = n < p.N;                          // N % 4 == 0: the whole vector is in or out
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (p.bias && nok && p.split_k <= 1) bv = *reinterpret_cast<const f32x4*>(p.bias + n);
    const bool has_auxin = (p.act == GAOT_ACT_GELU_BWD || p.act == GAOT_ACT_RELU_BWD);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) Cw[crow(r, lh) * LDC_S + j * 32 + li] = acc[i][j][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 2
        for (int ps = 0; ps < 32 / RPP; ++ps) {
            const int row = ps * RPP + lr;
            const int m = m0 + wm * WM + i * 32 + row;
            if (nok && m < p.M) {
                f32x4 v = *reinterpret_cast<const f32x4*>(Cw + row * LDC_S + lc) * so1 * so2;
                if (p.split_k > 1) {
                    *reinterpret_cast<f32x4*>(p.ws + ((long)zslab * p.M + m) * p.N + n) = v;
                } else {
                    v += bv;
                    if (p.rowbias) v += *reinterpret_cast<const f32x4*>(p.rowbias + (long)(m % p.rb_period) * p.ld_rb + n);
                    if (p.rowscale) v *= p.rowscale[m];
                    if (p.aux_out) *reinterpret_cast<f32x4*>(p.aux_out + (long)m * p.ld_aux + n) = v;
                    if (p.act == GAOT_ACT_GELU) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] = gelu_f(v[q]);
                    } else if (p.act == GAOT_ACT_RELU) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
                    } else if (has_auxin) {
                        const f32x4 ax = *reinterpret_cast<const f32x4*>(p.aux_in + (long)m * p.ld_aux + n);
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            v[q] = (p.act == GAOT_ACT_GELU_BWD) ? v[q] * gelu_grad_f(ax[q]) : (ax[q] > 0.f ? v[q] : 0.f);
                    } else if (p.act == GAOT_ACT_SWIGLU_BWD) {
                        const f32x4 u1 = *reinterpret_cast<const f32x4*>(p.aux_in + (long)m * p.ld_aux + n);
                        const f32x4 u3 = *reinterpret_cast<const f32x4*>(p.aux_in + (long)m * p.ld_aux + p.N + n);
                        f32x4 d3;
#pragma unroll
                        for (int q = 0; q < 4; ++q) { float d1, d3q; swiglu_grad_f(v[q], u1[q], u3[q], d1, d3q); v[q] = d1; d3[q] = d3q; }
                        *reinterpret_cast<f32x4*>(p.C + (long)m * p.ldc + p.N + n) = d3;
                        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(d3[0]), fabsf(d3[1])), fmaxf(fabsf(d3[2]), fabsf(d3[3]))));
                    }
                    if (p.residual) v += *reinterpret_cast<const f32x4*>(p.residual + (long)m * p.ldr + n);
                    *reinterpret_cast<f32x4*>(p.C + (long)m * p.ldc + n) = v;
                    amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (p.c_amax != nullptr && p.split_k <= 1) amax_publish(p.c_amax, amax, lane, (int)blockIdx.x * 4 + wave);
}

// SwiGLU forward epilogue (GAOT_ACT_SWIGLU): the kernel staged the B rows so that every wave's WN-column band holds
// u1 of WN/2 gate columns followed by u3 of the same columns; the gate is applied while the band sits in the wave's
// LDS slab, so u is written once (for the backward pass) and never read back.
template <int TM, int TN, int WM, int WN>
__device__ __forceinline__ void epilogue_swiglu(const GemmArgs& p, float* smem, const f32x16 (&acc)[TM][TN], int m0, int n0,
                                            int wm, int wn, int wave, int lane, float so1 = 1.f, float so2 = 1.f) {
    float amax = 0.f;
    const int li = lane & 31, lh = lane >> 5;
    constexpr int LDC_S = WN + 4;
    constexpr int HW = WN / 2;                         // gate columns per band
    constexpr int LPR = HW / 4, RPP = 64 / LPR;
    float* Cw = smem + wave * 32 * LDC_S;
    const int lr = lane / LPR, lc = (lane % LPR) * 4;
    const int F = p.N >> 1;
    const int gcol = (n0 >> 1) + wn * HW + lc;
    const bool nok = gcol < F;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) Cw[crow(r, lh) * LDC_S + j * 32 + li] = acc[i][j][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 2
        for (int ps = 0; ps < 32 / RPP; ++ps) {
            const int row = ps * RPP + lr;
            const int m = m0 + wm * WM + i * 32 + row;
            if (nok && m < p.M) {
                const f32x4 u1 = *reinterpret_cast<const f32x4*>(Cw + row * LDC_S + lc) * so1 * so2;
                const f32x4 u3 = *reinterpret_cast<const f32x4*>(Cw + row * LDC_S + HW + lc) * so1 * so2;
                if (p.aux_out) {
                    *reinterpret_cast<f32x4*>(p.aux_out + (long)m * p.ld_aux + gcol) = u1;
                    *reinterpret_cast<f32x4*>(p.aux_out + (long)m * p.ld_aux + F + gcol) = u3;
                }
                f32x4 o;
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = swiglu_f(u1[q], u3[q]);
                *reinterpret_cast<f32x4*>(p.C + (long)m * p.ldc + gcol) = o;
                amax = fmaxf(amax, fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (p.c_amax != nullptr) amax_publish(p.c_amax, amax, lane, (int)blockIdx.x * 4 + wave);
}

// skinny VALU paths (skinny.hip); return true if they handled the product
bool launch_skinny(const GemmArgs& a, bool a_kmajor, bool b_kmajor, hipStream_t st);
bool skinny_would(const GemmArgs& a, bool a_kmajor, bool b_kmajor);
// LDS-direct (global_load_lds) tile kernels (gemm_glds.hip); tile: 1 = 128x128 (8 waves), 2 = 128x64, 3 = 64x64
void launch_glds(GemmArgs& a, bool a_kmajor, bool b_kmajor, int tile, hipStream_t st);

// split-bf16 kernel (gemm_split.hip): fp32 product from six bf16 MFMA piece products, 128x128 tiles
void launch_split(GemmArgs& a, bool a_kmajor, bool b_kmajor, hipStream_t st, int bm = 128, int pieces = 3);
// all-DMA fp16-piece tiles (gemm_ad.hip): A k-contiguous, W pre-split into planes, K % 32 == 0, vector epilogue; bm x bn = 128 x 128, 64 x 128 or 64 x 64
void launch_ad(GemmArgs& a, bool b_kmajor, hipStream_t st, int bm, int bn = 128);
unsigned ad_redo_count(bool reset);
unsigned split_redo_count(bool reset);      // tiles that took the fp16 pieces' second (three-piece) pass: a device counter for tests / tools
// grouped weight-gradient launch (gemm_split.hip): prefix table / workspace need of n items; the launch itself
struct TnGroupArgs;
long plan_tn_grouped(const gaot_wgrad_item* items, int n, TnGroupArgs* args, int* n_counters, int* n_wg, int force_bm = 0);
void launch_tn_grouped(const gaot_wgrad_item* items, int n, float* ws, int* counters, int pieces, hipStream_t st);
constexpr int TN_GROUP_MAX = 24;          // products per launch (the table travels in the kernel arguments)

}  // namespace gaot
