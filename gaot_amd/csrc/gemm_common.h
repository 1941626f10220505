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
    int ablate;             // tuning only (gaot_debug_set_gemm_ablate): 1 = no in-loop global loads, 2 = no LDS staging/barriers, 4 = no stores
};

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
        default: break;
    }
    if (p.residual) v += p.residual[(long)m * p.ldr + n];
    p.C[(long)m * p.ldc + n] = v;
}

// skinny VALU paths (skinny.hip); return true if they handled the product
bool launch_skinny(const GemmArgs& a, bool a_kmajor, bool b_kmajor, hipStream_t st);

}  // namespace gaot
