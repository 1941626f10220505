// fp32-level GEMM on the bf16 matrix pipe, LDS-direct variant ("gsplit"): the operand tiles travel HBM/L2 -> LDS as fp32 with
// `global_load_lds_dwordx4` exactly as in gemm_glds.hip (no VGPR round trip, no ds_write pass, swizzled image, 2-stage ring, one
// raw barrier per k-tile) and are split into their three bf16 pieces IN REGISTERS, right after the fragment reads:
//     x = x1 + x2 + x3 (truncation),  x*y ~= x1y1 + (x1y2 + x2y1) + (x1y3 + x2y2 + x3y1)      (gemm_split.hip has the error analysis)
// Compared with gemm_split.hip (split on the way INTO LDS, three bf16 planes per operand): no global->VGPR->LDS staging in the
// k-loop, no plane writes, 4 B instead of 6 B of LDS per element -- the price is that every wave splits the fragments it reads
// (a 64x64 wave tile: 2x the split arithmetic of the shared-plane scheme).  gemm_split's ablations showed its k-loop phases
// (loads, split + plane writes, fragment reads + MFMA) running back to back rather than overlapped; here loads are asynchronous
// DMA and the only in-loop VALU work sits between the fragment reads and the MFMAs of the same wave.
// MFMA operand layout (v_mfma_f32_32x32x16_bf16): lane (row = lane & 31, half = lane >> 5) supplies 8 consecutive k.  A k-tile of
// 32 is two 16-k steps u; half h takes k = 16u + 8h .. +7 = the two 16-byte chunks 4u + 2h, 4u + 2h + 1 of the row (same for A
// and B, so any consistent assignment is a valid reduction order).
#include "gemm_common.h"

namespace gaot {

constexpr int SGBK = 32;

// 8 consecutive-k fp32 values of one row -> their bf16 pieces as MFMA operands (NP = 3: exact three-way split by truncation;
// NP = 1: one piece, rounded to nearest even: the `--dtype bf16` bench variant)
template <int NP>
__device__ __forceinline__ void split8(const float (&x)[8], bf16x8 (&out)[3]) {
    u32x4 h, m, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        unsigned a_, b_, c_;
        if (NP == 1) {
            unsigned u0 = __float_as_uint(x[2 * e]), u1 = __float_as_uint(x[2 * e + 1]);
            u0 += 0x7fffu + ((u0 >> 16) & 1u);
            u1 += 0x7fffu + ((u1 >> 16) & 1u);
            a_ = __builtin_amdgcn_perm(u1, u0, 0x07060302u); b_ = c_ = 0u;
        } else {
            split3_pair<0>(x[2 * e], x[2 * e + 1], a_, b_, c_);
        }
        h[e] = a_; m[e] = b_; l[e] = c_;
    }
    out[0] = __builtin_bit_cast(bf16x8, h);
    out[1] = __builtin_bit_cast(bf16x8, m);
    out[2] = __builtin_bit_cast(bf16x8, l);
}

template <int BM, int BN, int WAVES_M, bool AK, bool BKM, int NW, int NS, int NP>
__global__ __launch_bounds__(64 * NW) void gemm_gsplit_kernel(const GemmArgs p) {
    constexpr int WAVES_N = NW / WAVES_M;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int A_ST = BM * SGBK, B_ST = BN * SGBK;        // floats per stage per operand
    constexpr int STAGE = A_ST + B_ST;
    constexpr int LA = BM / (8 * NW), LB = BN / (8 * NW);  // 1-KiB DMA pieces per wave per tile
    static_assert(LA >= 1 && LB >= 1, "tile too small for the wave count");
    constexpr int EPI = NW * 32 * (WN + 4);
    constexpr int SMEM = NS * STAGE > EPI ? NS * STAGE : EPI;
    __shared__ __attribute__((aligned(16))) float smem[SMEM];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    const int tiles = p.tiles_m * p.tiles_n;
    int logical;
    {   // same XCD-aware tile order as gemm.hip
        const int q = tiles >> 3, r = tiles & 7, x = blockIdx.x & 7, slot = blockIdx.x >> 3;
        logical = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + slot;
    }
    const int m0 = (logical / p.tiles_n) * BM;
    const int n0 = (logical % p.tiles_n) * BN;

    const int nkt = p.K / SGBK;
    int kt_begin = 0, kt_end = nkt;
    if (p.split_k > 1) {
        kt_begin = blockIdx.z * p.ktiles_per_split;
        kt_end = min(nkt, kt_begin + p.ktiles_per_split);
    }

    // per-lane source description of this wave's DMA pieces (constant over k except for the k offset)
    long a_off[LA], b_off[LB];          // element offset of the lane's 16-byte chunk at k0 = 0
#pragma unroll
    for (int q = 0; q < LA; ++q) {
        const int t = (q * NW + wave) * 64 + lane;
        if (AK) {
            const int row = t >> 3, pc = t & 7, lc = pc ^ ((row >> 1) & 7);
            a_off[q] = (((long)min(m0 + row, p.M - 1)) << 8) | lc;                   // pack (row, chunk); ld applied per tile
        } else {
            const int kk = t / (BM / 4), r4 = t % (BM / 4);
            a_off[q] = (((long)min(m0 + r4 * 4, p.M - 4)) << 8) | kk;                 // pack (col, k)
        }
    }
#pragma unroll
    for (int q = 0; q < LB; ++q) {
        const int t = (q * NW + wave) * 64 + lane;
        if (BKM) {
            const int row = t >> 3, pc = t & 7, lc = pc ^ ((row >> 1) & 7);
            int nrow = min(n0 + row, p.N - 1);
            if (p.act == GAOT_ACT_SWIGLU) {      // band layout [u1 cols | u3 cols] per wave band (epilogue_swiglu)
                const int F = p.N >> 1, within = row % WN;
                const int gcol = (n0 >> 1) + (row / WN) * (WN / 2) + within % (WN / 2);
                nrow = (within / (WN / 2)) * F + min(gcol, F - 1);
            }
            b_off[q] = ((long)nrow << 8) | lc;
        } else {
            const int kk = t / (BN / 4), r4 = t % (BN / 4);
            b_off[q] = (((long)min(n0 + r4 * 4, p.N - 4)) << 8) | kk;
        }
    }

    auto issue = [&](int kt, int stage) {
        const int k0 = kt * SGBK;
        const float* abase = p.A; long lda = p.lda; int ka = k0;
        if (p.A2 != nullptr && k0 >= p.k_split) { abase = p.A2; lda = p.lda2; ka = k0 - p.k_split; }
        float* As = smem + stage * STAGE;
        float* Bs = As + A_ST;
#pragma unroll
        for (int q = 0; q < LA; ++q) {
            const long hi = a_off[q] >> 8; const int lo = (int)(a_off[q] & 255);
            const float* g = AK ? abase + hi * lda + ka + lo * 4 : abase + (long)(ka + lo) * lda + hi;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(As + (q * NW + wave) * 256), 16, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < LB; ++q) {
            const long hi = b_off[q] >> 8; const int lo = (int)(b_off[q] & 255);
            const float* g = BKM ? p.B + hi * p.ldb + k0 + lo * 4 : p.B + (long)(k0 + lo) * p.ldb + hi;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(Bs + (q * NW + wave) * 256), 16, 0, 0);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // swizzled read offsets (floats) of this lane's A / B rows
    int a_row[TM], a_sw[TM], b_row[TN], b_sw[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) { a_row[i] = wm * WM + i * 32 + li; a_sw[i] = (a_row[i] >> 1) & 7; }
#pragma unroll
    for (int j = 0; j < TN; ++j) { b_row[j] = wn * WN + j * 32 + li; b_sw[j] = (b_row[j] >> 1) & 7; }

    // fused column sum of an m-major A operand (bias gradient when A = dY): thread t < BM owns column t of the staged tile
    const bool do_colsum = !AK && p.colsum != nullptr && (logical % p.tiles_n) == 0;
    float csum = 0.f;

    if (kt_begin < kt_end) issue(kt_begin, 0);
    if (NS == 3 && kt_begin + 1 < kt_end) issue(kt_begin + 1, 1);
    int stage = 0;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        // tile kt must have landed (this wave's pieces); with a 3-deep ring the pieces of tile kt+1 may still be in flight
        if (NS == 3 && kt + 1 < kt_end) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LA + LB) : "memory");
        else                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();      // everyone's pieces landed; everyone is done reading the stage refilled next
        asm volatile("" ::: "memory");
        if (NS == 3) { if (kt + 2 < kt_end) issue(kt + 2, stage >= 1 ? stage - 1 : 2); }      // (stage + 2) % 3
        else         { if (kt + 1 < kt_end) issue(kt + 1, stage ^ 1); }
        const float* As = smem + stage * STAGE;
        const float* Bs = As + A_ST;
        if (!AK && do_colsum && tid < BM) {
#pragma unroll
            for (int kk = 0; kk < SGBK; ++kk) csum += As[kk * BM + tid];      // consecutive lanes -> consecutive banks
        }
#pragma unroll
        for (int u = 0; u < SGBK / 16; ++u) {
            bf16x8 a[TM][3], b[TN][3];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                float x[8];
                if (AK) {
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(As + a_row[i] * 32 + (((4 * u + 2 * lh) ^ a_sw[i]) << 2));
                    const f32x4 v1 = *reinterpret_cast<const f32x4*>(As + a_row[i] * 32 + (((4 * u + 2 * lh + 1) ^ a_sw[i]) << 2));
#pragma unroll
                    for (int s = 0; s < 4; ++s) { x[s] = v0[s]; x[4 + s] = v1[s]; }
                } else {
#pragma unroll
                    for (int s = 0; s < 8; ++s) x[s] = As[(16 * u + 8 * lh + s) * BM + a_row[i]];
                }
                split8<NP>(x, a[i]);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float x[8];
                if (BKM) {
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(Bs + b_row[j] * 32 + (((4 * u + 2 * lh) ^ b_sw[j]) << 2));
                    const f32x4 v1 = *reinterpret_cast<const f32x4*>(Bs + b_row[j] * 32 + (((4 * u + 2 * lh + 1) ^ b_sw[j]) << 2));
#pragma unroll
                    for (int s = 0; s < 4; ++s) { x[s] = v0[s]; x[4 + s] = v1[s]; }
                } else {
#pragma unroll
                    for (int s = 0; s < 8; ++s) x[s] = Bs[(16 * u + 8 * lh + s) * BN + b_row[j]];
                }
                split8<NP>(x, b[j]);
            }
            if (NP == 1) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {          // small terms first
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], acc[i][j], 0, 0, 0);
                    }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);
                    }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
                    }
            }
        }
        stage = (NS == 3) ? (stage == 2 ? 0 : stage + 1) : (stage ^ 1);
    }
    if (!AK && do_colsum && tid < BM && m0 + tid < p.M) {
        if (p.split_k > 1) p.ws[(long)p.split_k * p.M * p.N + (long)blockIdx.z * p.M + m0 + tid] = csum;
        else p.colsum[m0 + tid] = csum;
    }
    __syncthreads();
    if (BKM && p.act == GAOT_ACT_SWIGLU) epilogue_swiglu<TM, TN, WM, WN>(p, smem, acc, m0, n0, wm, wn, wave, lane);
    else                                 epilogue_vec<TM, TN, WM, WN>(p, smem, acc, m0, n0, wm, wn, wave, lane);
}

template <int BM, int BN, int WAVES_M, int NW, int NP>
static void launch_gsplit_cfg(GemmArgs& a, bool ak, bool bk, hipStream_t st) {
    a.tiles_m = cdiv(a.M, BM);
    a.tiles_n = cdiv(a.N, BN);
    dim3 grid(a.tiles_m * a.tiles_n, 1, a.split_k > 1 ? a.split_k : 1);
    dim3 block(64 * NW);
    if (ak && bk)        hipLaunchKernelGGL((gemm_gsplit_kernel<BM, BN, WAVES_M, true, true, NW, 2, NP>), grid, block, 0, st, a);
    else if (ak && !bk)  hipLaunchKernelGGL((gemm_gsplit_kernel<BM, BN, WAVES_M, true, false, NW, 2, NP>), grid, block, 0, st, a);
    else if (!ak && !bk) hipLaunchKernelGGL((gemm_gsplit_kernel<BM, BN, WAVES_M, false, false, NW, 2, NP>), grid, block, 0, st, a);
    else                 hipLaunchKernelGGL((gemm_gsplit_kernel<BM, BN, WAVES_M, false, true, NW, 2, NP>), grid, block, 0, st, a);
}

// bm = 128: 128x128 tile, 4 waves of 64x64;  bm = 64: 64x128 tile, 4 waves of 32x64
void launch_gsplit(GemmArgs& a, bool ak, bool bk, hipStream_t st, int bm, int pieces) {
    if (pieces == 1) {
        if (bm == 64) launch_gsplit_cfg<64, 128, 2, 4, 1>(a, ak, bk, st); else launch_gsplit_cfg<128, 128, 2, 4, 1>(a, ak, bk, st);
    } else {
        if (bm == 64) launch_gsplit_cfg<64, 128, 2, 4, 3>(a, ak, bk, st); else launch_gsplit_cfg<128, 128, 2, 4, 3>(a, ak, bk, st);
    }
}

}  // namespace gaot
