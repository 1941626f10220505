// fp32 MFMA GEMM, LDS-direct variant: operand tiles go HBM/L2 -> LDS with `global_load_lds_dwordx4` (no VGPR round
// trip, no ds_write pass), through a 2- (default) or 3-stage LDS ring, with ONE raw s_barrier per k-tile and explicit
// s_waitcnt vmcnt (counted, never 0 in steady state, for the 3-stage ring).
//
// The LDS-DMA destination is lane-linear (wave-uniform base + lane*16 B), so no row padding is possible:
//   * a k-contiguous operand tile is stored [row][32 floats] with the 16-byte chunks of a row XOR-swizzled by
//     ((row >> 1) & 7).  The swizzle is applied on the SOURCE address (which chunk a lane fetches) and again on the
//     ds_read_b128 address; 16 consecutive rows (and the b128 lane groups {0-3,12-15,20-27}...) then hit 16 distinct
//     4-bank slots: conflict-free, and one 128-bit read still feeds four MFMAs;
//   * a row-contiguous operand tile is stored [k][rows] as it comes; it is read with row-contiguous ds_read_b32.
// Rows beyond M / N are clamped to the last valid row (their products land in output rows that are never stored);
// K must be a multiple of 32 (all processor shapes) -- other cases use the register-staged kernel in gemm.hip.
#include "gemm_common.h"

namespace gaot {

constexpr int GBK = 32;

template <int BM, int BN, int WAVES_M, bool AK, bool BKM, int NW, int NS>
__global__ __launch_bounds__(64 * NW) void gemm_glds_kernel(const GemmArgs p) {
    constexpr int WAVES_N = NW / WAVES_M;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int A_ST = BM * GBK, B_ST = BN * GBK;        // floats per stage per operand
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

    const int nkt = p.K / GBK;
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
        const int k0 = kt * GBK;
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
            for (int kk = 0; kk < GBK; ++kk) csum += As[kk * BM + tid];      // consecutive lanes -> consecutive banks
        }
#pragma unroll
        for (int g = 0; g < GBK / 8; ++g) {
            float a[TM][4], b[TN][4];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if (AK) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(As + a_row[i] * 32 + (((2 * g + lh) ^ a_sw[i]) << 2));
                    a[i][0] = v[0]; a[i][1] = v[1]; a[i][2] = v[2]; a[i][3] = v[3];
                } else {
#pragma unroll
                    for (int s = 0; s < 4; ++s) a[i][s] = As[(8 * g + 4 * lh + s) * BM + a_row[i]];
                }
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (BKM) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(Bs + b_row[j] * 32 + (((2 * g + lh) ^ b_sw[j]) << 2));
                    b[j][0] = v[0]; b[j][1] = v[1]; b[j][2] = v[2]; b[j][3] = v[3];
                } else {
#pragma unroll
                    for (int s = 0; s < 4; ++s) b[j][s] = Bs[(8 * g + 4 * lh + s) * BN + b_row[j]];
                }
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
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

static int g_glds_stages = 2;     // 2-stage ring: half the LDS -> twice the resident workgroups; measured ahead of 3 stages
void set_glds_stages(int n) { g_glds_stages = n; }

template <int BM, int BN, int WAVES_M, int NW, int NS>
static void launch_glds_ns(GemmArgs& a, bool ak, bool bk, hipStream_t st) {
    dim3 grid(a.tiles_m * a.tiles_n, 1, a.split_k > 1 ? a.split_k : 1);
    dim3 block(64 * NW);
    if (ak && bk)        hipLaunchKernelGGL((gemm_glds_kernel<BM, BN, WAVES_M, true, true, NW, NS>), grid, block, 0, st, a);
    else if (ak && !bk)  hipLaunchKernelGGL((gemm_glds_kernel<BM, BN, WAVES_M, true, false, NW, NS>), grid, block, 0, st, a);
    else if (!ak && !bk) hipLaunchKernelGGL((gemm_glds_kernel<BM, BN, WAVES_M, false, false, NW, NS>), grid, block, 0, st, a);
    else                 hipLaunchKernelGGL((gemm_glds_kernel<BM, BN, WAVES_M, false, true, NW, NS>), grid, block, 0, st, a);
}

template <int BM, int BN, int WAVES_M, int NW>
static void launch_glds_cfg(GemmArgs& a, bool ak, bool bk, hipStream_t st) {
    a.tiles_m = cdiv(a.M, BM);
    a.tiles_n = cdiv(a.N, BN);
    if (g_glds_stages == 2) launch_glds_ns<BM, BN, WAVES_M, NW, 2>(a, ak, bk, st);
    else                    launch_glds_ns<BM, BN, WAVES_M, NW, 3>(a, ak, bk, st);
}

// tile: 1 = 128x128 (8 waves), 2 = 128x64, 3 = 64x64
void launch_glds(GemmArgs& a, bool ak, bool bk, int tile, hipStream_t st) {
    if (tile == 1)      launch_glds_cfg<128, 128, 2, 8>(a, ak, bk, st);
    else if (tile == 2) launch_glds_cfg<128, 64, 2, 4>(a, ak, bk, st);
    else                launch_glds_cfg<64, 64, 2, 4>(a, ak, bk, st);
}

}  // namespace gaot
