// Micro-benchmark: do the matrix pipe (v_mfma_f32_32x32x16_bf16) and the vector pipe overlap on gfx950, within one wave and
// across the waves of a SIMD?  (tuning tool, not part of the library)
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/pipe_overlap.hip -o tools/bin/pipe_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define MFMA(acc) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z))
#define AND(x) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(m))
#define PERM(x) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(m2))
#define PKADD(x) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(y2))
#define SUB(x) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x) : "v"(z))
#define CVTPK(x) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x) : "v"(y))
#define PKMOV(x) asm volatile("v_pk_mov_b32 %0, %0, %1" : "+v"(x) : "v"(y2))
#define LSHL(x) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(x))
#define BFI(x) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(x) : "v"(m), "v"(m2))
#define MAX3(x) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z))
#define PKFMA16(x) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(m2))
#define MULF(x) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(y))
#define ADDU(x) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(m))
#define FMAC(x) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x) : "v"(y), "v"(z))
#define MIXLO(x) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(x) : "v"(m), "v"(y))
#define MIXHI(x) asm volatile("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(x) : "v"(m), "v"(y))
#define EXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))

// MODE bit 0: MFMAs; bit 1: VALU ops; KIND: 0 fma, 1 and, 2 perm, 3 pk_add, 4 exp; NV = vector ops per MFMA slot
template <int MODE, int KIND, int NV, int CH = 2>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    u32x4 au = {threadIdx.x, 1u, 2u, 3u};
    const bf16x8 a = __builtin_bit_cast(bf16x8, au), b = a;
    float x[8]; unsigned xi[8]; double xd[8];
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 xp[8];
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 1e-3f + i; xi[i] = threadIdx.x + i; xp[i] = f32x2{x[i], x[i]}; }
    const float y = 0.999f, z = 1e-3f; const unsigned m = 0xffff0000u | threadIdx.x, m2 = 0x07060302u; const f32x2 y2 = {1e-3f, 1e-3f};
    __shared__ unsigned lds[8192];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const unsigned laddr = (threadIdx.x & 63) * 80 + (threadIdx.x >> 6) * 5120;
    u32x4 ld[4] = {}; typedef unsigned u32x2_ __attribute__((ext_vector_type(2))); u32x2_ ld2[4] = {};
    for (int it = 0; it < iters; ++it) {
        if (KIND >= 20) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (MODE & 1) { if ((s & 1) && CH == 2) MFMA(acc1); else MFMA(acc0); }
            if (MODE & 2) {
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    if (KIND == 0) FMA(x[v & 7]);
                    if (KIND == 1) AND(xi[v & 7]);
                    if (KIND == 2) PERM(xi[v & 7]);
                    if (KIND == 3) PKADD(xp[v & 7]);
                    if (KIND == 4) EXP(x[v & 7]);
                    if (KIND == 5) { if (v == 0) PKADD(xp[0]); else FMA(x[v & 7]); }           // one packed op per slot, the rest fma
                    if (KIND == 6) SUB(x[v & 7]);
                    if (KIND == 7) CVTPK(xi[v & 7]);
                    if (KIND == 8) PKMOV(xp[v & 7]);
                    if (KIND == 9) LSHL(xi[v & 7]);
                    if (KIND == 10) BFI(xi[v & 7]);
                    if (KIND == 11) MAX3(x[v & 7]);
                    if (KIND == 12) PKFMA16(xi[v & 7]);
                    if (KIND == 16) { if (v & 1) MIXHI(xi[v & 7]); else MIXLO(xi[v & 7]); }
                    if (KIND == 20) { asm volatile("ds_read_b128 %0, %1" : "=v"(ld[v & 3]) : "v"(laddr)); }
                    if (KIND == 21) { asm volatile("ds_read_b64 %0, %1" : "=v"(ld2[v & 3]) : "v"(laddr)); }
                    if (KIND == 22) { asm volatile("ds_write_b32 %0, %1" : : "v"(laddr), "v"(xi[v & 7]) : "memory"); }
                }
            }
        }
    }
    float r = 0.f;
    for (int i = 0; i < 16; ++i) r += acc0[i] + acc1[i];
    for (int i = 0; i < 8; ++i) r += x[i] + (float)xi[i] + xp[i][0] + xp[i][1];
    if (KIND >= 20) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); for (int i = 0; i < 4; ++i) r += ld[i][0] + ld[i][3] + ld2[i][0] + ld2[i][1]; }
    if (r == 123.456f) out[0] = r;
}

// waves 0..3 of the workgroup (one per SIMD) run MFMAs only, waves 4..7 (the second wave of each SIMD) run vector ops only
template <int KIND, int NV>
__global__ __launch_bounds__(512) void k_opposed(float* out, int iters, int who) {
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    u32x4 au = {threadIdx.x, 1u, 2u, 3u};
    const bf16x8 a = __builtin_bit_cast(bf16x8, au), b = a;
    float x[8]; unsigned xi[8];
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 1e-3f + i; xi[i] = threadIdx.x + i; }
    const float y = 0.999f, z = 1e-3f; const unsigned m = 0xffff0000u | threadIdx.x, m2 = 0x07060302u;
    const bool mf = threadIdx.x < 256;
    if (mf && (who & 1)) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int s = 0; s < 8; ++s) { if (s & 1) MFMA(acc1); else MFMA(acc0); }
        }
    }
    if (!mf && (who & 2)) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    if (KIND == 0) FMA(x[v & 7]);
                    if (KIND == 1) AND(xi[v & 7]);
                    if (KIND == 2) PERM(xi[v & 7]);
                    if (KIND == 4) EXP(x[v & 7]);
                    if (KIND == 6) SUB(x[v & 7]);
                    if (KIND == 7) CVTPK(xi[v & 7]);
                    if (KIND == 11) MAX3(x[v & 7]);
                    if (KIND == 13) MULF(x[v & 7]);
                    if (KIND == 14) ADDU(xi[v & 7]);
                    if (KIND == 15) FMAC(x[v & 7]);
                    if (KIND == 16) { if (v & 1) MIXHI(xi[v & 7]); else MIXLO(xi[v & 7]); }
                }
        }
    }
    float r = 0.f;
    for (int i = 0; i < 16; ++i) r += acc0[i] + acc1[i];
    for (int i = 0; i < 8; ++i) r += x[i] + (float)xi[i];
    if (r == 123.456f) out[0] = r;
}
template <int KIND, int NV>
static void opposed(const char* name, float* out) {
    const int iters = 20000;
    double t[4];
    for (int who = 1; who <= 3; ++who) {
        hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
        hipLaunchKernelGGL((k_opposed<KIND, NV>), dim3(256), dim3(512), 0, 0, out, 200, who);
        hipEventRecord(s, 0);
        hipLaunchKernelGGL((k_opposed<KIND, NV>), dim3(256), dim3(512), 0, 0, out, iters, who);
        hipEventRecord(e, 0); hipEventSynchronize(e);
        float ms; hipEventElapsedTime(&ms, s, e);
        t[who] = ms * 1e6 / (iters * 8.0);
    }
    printf("opposed waves on one SIMD, %s NV=%d: mfma wave alone %.1f ns/slot, vector wave alone %.1f, both running %.1f\n", name, NV, t[1], t[2], t[3]);
}

template <int MODE, int KIND, int NV, int CH = 2>
static double run(int threads, float* out) {
    const int iters = 20000;
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    hipLaunchKernelGGL((k<MODE, KIND, NV, CH>), dim3(256), dim3(threads), 0, 0, out, 200);
    hipEventRecord(s, 0);
    hipLaunchKernelGGL((k<MODE, KIND, NV, CH>), dim3(256), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e, 0); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    return ms * 1e6 / (iters * 8.0);       // ns per slot (one MFMA and/or NV vector ops) per wave
}

template <int KIND, int NV, int CH = 2>
static void row(const char* name, float* out) {
    for (int threads : {256, 512}) {
        const double m = run<1, KIND, NV, CH>(threads, out), v = run<2, KIND, NV, CH>(threads, out), b = run<3, KIND, NV, CH>(threads, out);
        printf("chains=%d %-8s NV=%2d waves/SIMD=%d : mfma only %.1f ns/slot, vector only %.1f, both %.1f  (sum %.1f, max %.1f)\n", CH, name, NV, threads / 256, m, v, b, m + v,
               m > v ? m : v);
    }
}

int main() {
    float* out; hipMalloc(&out, 4);
    opposed<0, 4>("fma", out); opposed<0, 8>("fma", out); opposed<0, 12>("fma", out); opposed<1, 8>("and", out); opposed<2, 8>("perm", out);
    opposed<4, 4>("exp", out); opposed<6, 8>("sub_f32", out); opposed<7, 8>("cvt_pk_bf16", out); opposed<11, 8>("max3", out); opposed<13, 8>("mul_f32", out);
    opposed<14, 8>("add_u32", out); opposed<15, 8>("fmac_f32 (VOP2)", out); opposed<16, 8>("fma_mixlo/hi_f16", out);
    row<16, 8>("fma_mix", out);
    row<20, 1>("ds_read_b128", out); row<20, 2>("ds_read_b128", out); row<21, 2>("ds_read_b64", out); row<22, 2>("ds_write_b32", out);
    row<0, 8>("fma", out); row<3, 8>("pk_add", out); row<5, 8>("1pk+7fma", out); row<6, 8>("sub", out); row<7, 8>("cvt_pk_bf16", out);
    row<8, 8>("pk_mov", out); row<9, 8>("lshl", out); row<10, 8>("bfi", out); row<11, 8>("max3", out); row<1, 8>("and", out); row<2, 8>("perm", out);
    row<12, 8>("pk_fma_f16", out); row<4, 4>("exp", out);
    return 0;
}
