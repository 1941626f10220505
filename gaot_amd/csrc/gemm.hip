// fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32), fused epilogue, optional split-K.
//
// Design notes (MI355X):
//  * f32-input MFMA takes ONE f32 per lane per operand, so any operand can be fed from LDS in either
//    orientation; reduction order inside a k-tile is free as long as A and B agree.  We use that:
//      - an operand whose reduction dim is contiguous in HBM ("k-major") is staged as [row][k] with a
//        +4 pad (stride 36 dwords: 16 consecutive rows land on 16 distinct 4-bank slots -> conflict-free
//        ds_read_b128) and each 128-bit read feeds FOUR MFMAs (k-slots: half-wave h takes k = 8g+4h+s);
//      - an operand whose row dim is contiguous ("m-major", the transposed products of backward) is staged
//        as [k][row] and read with one conflict-free ds_read_b32 per MFMA at the same k = 8g+4h+s.
//  * 256 threads = 4 waves; 64-cycle MFMA issue makes the kernel matrix-pipe bound with register-prefetched
//    single-buffer staging (next k-tile's global loads in flight under 16*TM*TN MFMAs).
//  * workgroup ids are remapped so that one XCD (private L2) owns whole row-panels of A.
#include "gemm_common.h"

namespace gaot {

constexpr int BK = 32;

// global -> registers for one [ROWS x BK] operand tile.  KMAJ: elem(row,k) = base[row*ld + k].
template <bool KMAJ, bool VEC, int ROWS, int NT>
__device__ __forceinline__ void load_tile(f32x4 (&r)[ROWS * 8 / NT], const float* __restrict__ base, long ld,
                                          int row0, int nrows, int k0, int klim, int tid) {
#pragma unroll
    for (int p = 0; p < ROWS * 8 / NT; ++p) {
        const int t = tid + p * NT;
        int row, k;
        if (KMAJ) { row = row0 + (t >> 3); k = k0 + (t & 7) * 4; }
        else      { k = k0 + t / (ROWS / 4); row = row0 + (t % (ROWS / 4)) * 4; }
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (VEC) {
            if (row < nrows && k < klim) {
                const float* src = KMAJ ? base + (long)row * ld + k : base + (long)k * ld + row;
                v = *reinterpret_cast<const f32x4*>(src);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int rr = KMAJ ? row : row + j;
                const int kk = KMAJ ? k + j : k;
                if (rr < nrows && kk < klim) v[j] = KMAJ ? base[(long)rr * ld + kk] : base[(long)kk * ld + rr];
            }
        }
        r[p] = v;
    }
}

template <bool KMAJ, int ROWS, int NT>
__device__ __forceinline__ void store_tile(float* __restrict__ s, const f32x4 (&r)[ROWS * 8 / NT], int tid) {
#pragma unroll
    for (int p = 0; p < ROWS * 8 / NT; ++p) {
        const int t = tid + p * NT;
        if (KMAJ) *reinterpret_cast<f32x4*>(s + (t >> 3) * (BK + 4) + (t & 7) * 4) = r[p];
        else      *reinterpret_cast<f32x4*>(s + (t / (ROWS / 4)) * ROWS + (t % (ROWS / 4)) * 4) = r[p];
    }
}

// bijective "one XCD owns a contiguous chunk of logical tiles" remap (dispatcher places id on XCD id%8)
__device__ __forceinline__ int xcd_remap(int id, int total) {
    const int q = total >> 3, r = total & 7, x = id & 7, slot = id >> 3;
    const int start = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return start + slot;
}

template <int BM, int BN, int WAVES_M, bool AK, bool BKM, bool VEC, int NW = 4>
__global__ __launch_bounds__(64 * NW) void gemm_kernel(const GemmArgs p) {
    constexpr int NT = 64 * NW;
    constexpr int WAVES_N = NW / WAVES_M;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int LDA_S = AK ? BK + 4 : BM;
    constexpr int LDB_S = BKM ? BK + 4 : BN;
    constexpr int A_ELEMS = AK ? BM * (BK + 4) : BK * BM;
    constexpr int B_ELEMS = BKM ? BN * (BK + 4) : BK * BN;
    constexpr int EPI_ELEMS = NW * 32 * (WN + 4);          // per-wave transposition slabs of the vector epilogue
    constexpr int SMEM_ELEMS = (A_ELEMS + B_ELEMS) > EPI_ELEMS ? (A_ELEMS + B_ELEMS) : EPI_ELEMS;
    __shared__ __attribute__((aligned(16))) float smem[SMEM_ELEMS];
    float* As = smem;
    float* Bs = smem + A_ELEMS;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    const int tiles = p.tiles_m * p.tiles_n;
    const int logical = xcd_remap(blockIdx.x, tiles);
    const int m0 = (logical / p.tiles_n) * BM;
    const int n0 = (logical % p.tiles_n) * BN;

    const int nkt = (p.K + BK - 1) / BK;
    int kt_begin = 0, kt_end = nkt;
    if (p.split_k > 1) {
        kt_begin = blockIdx.z * p.ktiles_per_split;
        kt_end = min(nkt, kt_begin + p.ktiles_per_split);
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Two register sets: tile kt+2 is in flight while tile kt is computed and tile kt+1 waits in the other set, so a
    // load has TWO compute phases to land (HBM misses take ~2-3k cycles; with ~2 resident waves per SIMD a one-deep
    // prefetch left the matrix pipe idle 13-42 % of the time -- tools/gemm_ablate.py).
    f32x4 ra0[BM * 8 / NT], rb0[BN * 8 / NT], ra1[BM * 8 / NT], rb1[BN * 8 / NT];
    // fused column sum of the m-major A operand (= bias gradient when A is dY): every thread owns 4 consecutive
    // rows (m) of the tile at some k; blocks of the first n-tile column do the work
    const bool do_colsum = !AK && p.colsum != nullptr && (logical % p.tiles_n) == 0;
    f32x4 csum = {0.f, 0.f, 0.f, 0.f};

    auto fetch = [&](f32x4 (&ra)[BM * 8 / NT], f32x4 (&rb)[BN * 8 / NT], int kt) {
        const int k0 = kt * BK;
        if (p.A2 != nullptr && k0 >= p.k_split)
            load_tile<AK, VEC, BM, NT>(ra, p.A2, p.lda2, m0, p.M, k0 - p.k_split, p.K - p.k_split, tid);
        else
            load_tile<AK, VEC, BM, NT>(ra, p.A, p.lda, m0, p.M, k0, p.A2 ? p.k_split : p.K, tid);
        load_tile<BKM, VEC, BN, NT>(rb, p.B, p.ldb, n0, p.N, k0, p.K, tid);
    };
    auto compute = [&]() {
#pragma unroll
        for (int g = 0; g < BK / 8; ++g) {
            float a[TM][4], b[TN][4];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = wm * WM + i * 32 + li;
                if (AK) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(As + row * LDA_S + 8 * g + 4 * lh);
                    a[i][0] = v[0]; a[i][1] = v[1]; a[i][2] = v[2]; a[i][3] = v[3];
                } else {
#pragma unroll
                    for (int s = 0; s < 4; ++s) a[i][s] = As[(8 * g + 4 * lh + s) * LDA_S + row];
                }
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = wn * WN + j * 32 + li;
                if (BKM) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(Bs + col * LDB_S + 8 * g + 4 * lh);
                    b[j][0] = v[0]; b[j][1] = v[1]; b[j][2] = v[2]; b[j][3] = v[3];
                } else {
#pragma unroll
                    for (int s = 0; s < 4; ++s) b[j][s] = Bs[(8 * g + 4 * lh + s) * LDB_S + col];
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
    };
    // one pipeline step: stage the set holding tile kt, refill it with tile kt+2, compute tile kt
    auto step = [&](f32x4 (&ra)[BM * 8 / NT], f32x4 (&rb)[BN * 8 / NT], int kt) {
        if (!(p.ablate & 2)) {
            store_tile<AK, BM, NT>(As, ra, tid);
            store_tile<BKM, BN, NT>(Bs, rb, tid);
        }
        if (!AK && do_colsum) {
#pragma unroll
            for (int q = 0; q < BM * 8 / NT; ++q) csum += ra[q];
        }
        if (!(p.ablate & 2)) __syncthreads();
        if (kt + 2 < kt_end && !(p.ablate & 1)) fetch(ra, rb, kt + 2);
        compute();
        if (!(p.ablate & 2)) __syncthreads();
    };

    if (kt_begin < kt_end) fetch(ra0, rb0, kt_begin);
    if (kt_begin + 1 < kt_end) fetch(ra1, rb1, kt_begin + 1);
    if (p.ablate & 2) {          // ablation: stage once, then compute on the same LDS contents without barriers
        store_tile<AK, BM, NT>(As, ra0, tid);
        store_tile<BKM, BN, NT>(Bs, rb0, tid);
        __syncthreads();
    }
    int kt = kt_begin;
    for (; kt + 1 < kt_end; kt += 2) {
        step(ra0, rb0, kt);
        step(ra1, rb1, kt + 1);
    }
    if (kt < kt_end) step(ra0, rb0, kt);
    if (p.ablate & 4) {          // ablation: keep the accumulators alive but store (almost) nothing
        float keep = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) keep += acc[i][j][r];
        if (keep == 123.456f) p.C[0] = keep;
        return;
    }

    if (!AK && do_colsum) {      // threads with equal (tid % (BM/4)) hold partials of the same 4 rows: reduce through LDS
        constexpr int G = BM / 4, NG = NT / G;
        __syncthreads();
        *reinterpret_cast<f32x4*>(smem + (tid / G) * BM + (tid % G) * 4) = csum;
        __syncthreads();
        if (tid < BM) {
            float v = 0.f;
#pragma unroll
            for (int g2 = 0; g2 < NG; ++g2) v += smem[g2 * BM + tid];
            const int m = m0 + tid;
            if (m < p.M) {
                if (p.split_k > 1) p.ws[(long)p.split_k * p.M * p.N + (long)blockIdx.z * p.M + m] = v;
                else p.colsum[m] = v;
            }
        }
    }
    if (p.vec_epi) {
        __syncthreads();                                   // all waves are done with As / Bs
        epilogue_vec<TM, TN, WM, WN>(p, smem, acc, m0, n0, wm, wn, wave, lane);
        return;
    }
    // scalar fallback epilogue (odd N / unaligned operands): C-layout rows crow(r, lh), column li
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {          // static indices only: runtime-indexed accumulators would go to scratch
            const int m = m0 + wm * WM + i * 32 + crow(r, lh);
            if (m >= p.M) continue;
            if (p.split_k > 1) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int n = n0 + wn * WN + j * 32 + li;
                    if (n < p.N) p.ws[((long)blockIdx.z * p.M + m) * p.N + n] = acc[i][j][r];
                }
            } else {
                const RowCtx rc = row_ctx(p, m);
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int n = n0 + wn * WN + j * 32 + li;
                    if (n < p.N) epilogue_store_row(p, rc, m, n, acc[i][j][r]);
                }
            }
        }
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmArgs p) {
    const long total = (long)p.M * p.N;
    float amax = 0.f;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        float v = 0.f;
        for (int z = 0; z < p.split_k; ++z) v += p.ws[(long)z * total + idx];
        const int m = (int)(idx / p.N), n = (int)(idx % p.N);
        epilogue_store(p, m, n, v);
        if (p.c_amax != nullptr) {          // the output's magnitude word: over C as stored (both halves of a SwiGLU gradient)
            amax = fmaxf(amax, fabsf(p.C[(long)m * p.ldc + n]));
            if (p.act == GAOT_ACT_SWIGLU_BWD) amax = fmaxf(amax, fabsf(p.C[(long)m * p.ldc + p.N + n]));
        }
    }
    __shared__ float red_amax[4];
    if (p.c_amax != nullptr) amax_publish_block<4>(p.c_amax, amax, red_amax);
    if (p.colsum) {
        const float* cs = p.ws + (long)p.split_k * total;
        for (long m = (long)blockIdx.x * blockDim.x + threadIdx.x; m < p.M; m += (long)gridDim.x * blockDim.x) {
            float v = 0.f;
            for (int z = 0; z < p.split_k; ++z) v += cs[(long)z * p.M + m];
            p.colsum[m] = v;
        }
    }
}

// z-parallel sum of `nz` slabs of `count` floats each: wave w takes z = w, w+4, ... with 8 loads in flight,
// partial sums meet in LDS.  One lane per output element (VEC: per 4 elements).  Fixed order -> deterministic.
template <typename T>
__device__ __forceinline__ T zsum(const T* __restrict__ base, long stride, int nz, int wave) {
    T v = T{};
    int z = wave;
    for (; z + 28 < nz; z += 32) {
        T t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = base[(long)(z + 4 * u) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) v += t[u];
    }
    for (; z < nz; z += 4) v += base[(long)z * stride];
    return v;
}
// plain-sum fast path (no epilogue extras, N % 4 == 0, 16-byte aligned rows); block = 4 z-waves x 64 outputs
__global__ __launch_bounds__(256) void splitk_reduce_vec_kernel(const GemmArgs p) {
    __shared__ f32x4 red[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long total4 = (long)p.M * p.N / 4;
    const int n4 = p.N / 4;
    const long nblk_main = (total4 + 63) / 64;
    if (blockIdx.x < nblk_main) {
        const long idx = (long)blockIdx.x * 64 + lane;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (idx < total4) v = zsum(reinterpret_cast<const f32x4*>(p.ws) + idx, total4, p.split_k, wave);
        red[wave][lane] = v;
        __syncthreads();
        if (wave == 0) {
            float av = 0.f;
            if (idx < total4) {
                f32x4 r = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
                const long m = idx / n4;
                const int n = (int)(idx - m * n4) * 4;
                if (p.residual) r += *reinterpret_cast<const f32x4*>(p.residual + m * p.ldr + n);
                *reinterpret_cast<f32x4*>(p.C + m * p.ldc + n) = r;
                av = fmaxf(fmaxf(fabsf(r[0]), fabsf(r[1])), fmaxf(fabsf(r[2]), fabsf(r[3])));
            }
            if (p.c_amax != nullptr) amax_publish(p.c_amax, av, lane, (int)blockIdx.x);      // the output's magnitude word
        }
    } else {                                     // trailing blocks: the fused column-sum slab [split_k][M]
        float* redf = reinterpret_cast<float*>(&red[0][0]);
        const long m = (long)(blockIdx.x - nblk_main) * 64 + lane;
        float v = 0.f;
        if (m < p.M) v = zsum(p.ws + (long)p.split_k * p.M * p.N + m, (long)p.M, p.split_k, wave);
        redf[wave * 64 + lane] = v;
        __syncthreads();
        if (wave == 0 && m < p.M) p.colsum[m] = redf[lane] + redf[64 + lane] + redf[128 + lane] + redf[192 + lane];
    }
}

template <int BM, int BN, int WAVES_M, int NW = 4>
static int launch_cfg(GemmArgs& a, bool ak, bool bk, bool vec, hipStream_t st) {
    a.tiles_m = cdiv(a.M, BM);
    a.tiles_n = cdiv(a.N, BN);
    dim3 grid(a.tiles_m * a.tiles_n, 1, a.split_k > 1 ? a.split_k : 1);
    dim3 block(64 * NW);
#define GAOT_LAUNCH(AKv, BKv, Vv) hipLaunchKernelGGL((gemm_kernel<BM, BN, WAVES_M, AKv, BKv, Vv, NW>), grid, block, 0, st, a)
    if (ak && bk)        { if (vec) GAOT_LAUNCH(true, true, true);   else GAOT_LAUNCH(true, true, false); }
    else if (ak && !bk)  { if (vec) GAOT_LAUNCH(true, false, true);  else GAOT_LAUNCH(true, false, false); }
    else if (!ak && !bk) { if (vec) GAOT_LAUNCH(false, false, true); else GAOT_LAUNCH(false, false, false); }
    else                 { if (vec) GAOT_LAUNCH(false, true, true);  else GAOT_LAUNCH(false, true, false); }
#undef GAOT_LAUNCH
    return 0;
}

}  // namespace gaot

using namespace gaot;

static thread_local int g_last_path = 0;   // 1 = MFMA tile kernel, 2 = skinny VALU path (for the bench's roofline accounting)
extern "C" int gaot_debug_last_gemm_path(void) { return g_last_path; }
static int g_use_glds = 1;   // eligible products run on the LDS-direct kernels (gemm_glds.hip); 0 = register-staged only
namespace gaot { void set_glds_stages(int n); }
// on: 0 = register-staged kernels only, 1 = LDS-direct with the default 2-stage ring, 3 = LDS-direct with a 3-stage ring,
// 4 = + split-bf16 tiles by heuristic (DEFAULT: +5 % on the GAOT step, same-box A/B on two boxes),
// 5 = split-bf16 wherever eligible, 6 = split-bf16 for the SwiGLU-gate product only.
static int g_split_bm256 = 0;    // 1: 256x128 (8-wave) split tiles when they still fill the chip, 2: wherever M >= 256 (tuning)
static int g_use_split = 1;  // 1: eligible products run on the split-bf16 kernel (gemm_split.hip) per the heuristic; 2: always when eligible
extern "C" int gaot_debug_set_gemm_glds(int on) {
    const int old = g_use_split ? 3 + g_use_split : g_use_glds;
    g_use_glds = on != 0; g_use_split = on == 4 ? 1 : (on == 5 ? 2 : (on == 6 ? 3 : (on == 7 ? 4 : (on == 8 ? 5 : (on == 9 ? 6 : 0)))));   // 6: SwiGLU product only, 7: 64-row tiles wherever eligible, 8: as 4 without the 64-row tiles
    g_split_bm256 = on == 10 ? 1 : (on == 11 ? 2 : 0);
    if (on == 10) g_use_split = 1;
    if (on == 11) g_use_split = 2;
    gaot::set_glds_stages(on == 3 ? 3 : 2);
    return old;
}
// precision override of the split tiles: 0 (default) = every call's own gaot_gemm_desc.pieces; 1 = operands rounded to bf16, one piece
// product (bench `--dtype bf16` only); 2 / 3 = forced for A/B runs
static int g_split_pieces = 3, g_split_pieces_forced = 0;
extern "C" int gaot_debug_set_gemm_pieces(int n) {
    const int old = g_split_pieces_forced ? g_split_pieces : 0;
    g_split_pieces_forced = (n >= 1 && n <= 3); g_split_pieces = g_split_pieces_forced ? n : 3;
    return old;
}
int gaot_forced_pieces() { return g_split_pieces_forced ? g_split_pieces : 0; }

static int g_use_ad = 1;         // all-DMA fp16-piece tiles: 0 off, 1 per the heuristic, 2 / 3: 64- / 128-row tiles wherever eligible (A/B switch)
extern "C" int gaot_debug_set_gemm_ad(int on) { const int old = g_use_ad; g_use_ad = on; return old; }
static int g_ad_flush = 1;      // [r6] unsplit reductions of 1 025 .. 2 048 on the 64 x 64 all-DMA tiles with the in-kernel flush (gemm_ad.hip FL); 0 = off (K slabs as before)
extern "C" int gaot_debug_set_gemm_ad_flush(int on) { const int old = g_ad_flush; if (on >= 0) g_ad_flush = on; return old; }      // (on < 0: query)
static int g_ad_narrow = 5;     // 64 x 64 all-DMA tiles (A/B switch, bits): 1 = half-filled launches of outputs two tiles wide (default), 2 = also N <= 256, K <= 256 at full launches, 4 = also outputs THREE tiles wide (4 096 tokens x 384: the 3-D configuration's o_proj-shaped products, which fell to the fp32-MFMA tiles; default), 8 = K slabs of such products too (4 096 x 384 x 1 152 in two slabs: C5 5.20 -> 5.27 ms same-box, tools/c45_ab.py 5 13: off), 16 = the concatenated-input (A2) product of the skip block too (C4 1.498 -> 1.500, C5 5.12 -> 5.09: nothing: off)
extern "C" int gaot_debug_set_gemm_ad_narrow(int on) { const int old = g_ad_narrow; g_ad_narrow = on; return old; }
static int g_use_planes = 1;     // 0: ignore gaot_gemm_desc.b_planes (A/B switch)
extern "C" unsigned gaot_debug_split_redo_count(int reset) { return gaot::split_redo_count(reset != 0) + gaot::ad_redo_count(reset != 0); }
extern "C" int gaot_debug_set_gemm_planes(int on) { const int old = g_use_planes; g_use_planes = on; return old; }
static int g_ablate = 0;
extern "C" int gaot_debug_set_gemm_ablate(int bits) { const int old = g_ablate; g_ablate = bits; return old; }
static int g_tile_override = 0;   // tuning hook: 0 = heuristic, 1 = 128x128, 2 = 128x64, 3 = 64x64, 4 = 128x32
extern "C" int gaot_debug_set_gemm_tile(int cfg) { const int old = g_tile_override; g_tile_override = cfg; return old; }

// dry: only decide which kernel family would serve the product (g_last_path), launch nothing
static int gemm_run(const gaot_gemm_desc* d, gaot_stream_t stream, const bool dry) {
    GAOT_REQUIRE(d != nullptr, "gemm: null descriptor");
    GAOT_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "gemm: M,N,K must be positive (got %d,%d,%d)", d->M, d->N, d->K);
    GAOT_REQUIRE(d->A && d->B && (d->C || d->raw_slabs), "gemm: A, B, C must be non-null");
    if (d->raw_slabs)
        GAOT_REQUIRE(d->split_k > 1 && d->workspace && !d->bias && !d->rowbias && !d->rowscale && d->act == GAOT_ACT_NONE && !d->aux_in && !d->aux_out &&
                     !d->residual && !d->colsum && d->N % 4 == 0 && aligned16(d->workspace),
                     "gemm: raw_slabs needs split_k > 1, a 16-byte aligned workspace, N %% 4 == 0 and no epilogue operand");
    GAOT_REQUIRE(d->act >= GAOT_ACT_NONE && d->act <= GAOT_ACT_SWIGLU, "gemm: bad act %d", d->act);
    if (d->act == GAOT_ACT_GELU_BWD || d->act == GAOT_ACT_RELU_BWD || d->act == GAOT_ACT_SWIGLU_BWD)
        GAOT_REQUIRE(d->aux_in != nullptr, "gemm: *_BWD activation needs aux_in");
    if (d->act == GAOT_ACT_SWIGLU_BWD)
        GAOT_REQUIRE(!d->residual && !d->aux_out && d->ldc >= 2 * (int64_t)d->N && d->ld_aux >= 2 * (int64_t)d->N,
                     "gemm: SWIGLU_BWD writes [M,2N] (ldc, ld_aux >= 2N) and takes no residual / aux_out");
    if (d->act == GAOT_ACT_SWIGLU) {
        const bool ok = d->b_kmajor && d->K % 32 == 0 && d->N % 8 == 0 && !d->bias && !d->rowbias && !d->rowscale &&
                        !d->residual && !d->colsum && d->split_k <= 1 && d->A2 == nullptr && aligned16(d->A) && aligned16(d->B) &&
                        aligned16(d->C) && d->lda % 4 == 0 && d->ldb % 4 == 0 && d->ldc % 4 == 0 &&
                        (d->a_kmajor || d->M % 4 == 0) && (!d->aux_out || (aligned16(d->aux_out) && d->ld_aux % 4 == 0));
        GAOT_REQUIRE(ok, "gemm: SWIGLU needs a k-major [2F,K] weight, K %% 32 == 0, F %% 4 == 0, 16-byte aligned operands and no "
                         "bias / row bias / row scale / residual / colsum / split_k / A2");
    }
    if (d->rowbias) GAOT_REQUIRE(d->rowbias_period > 0, "gemm: rowbias needs rowbias_period > 0");
    if (d->A2) GAOT_REQUIRE(d->k_split > 0 && d->k_split < d->K && d->k_split % BK == 0,
                            "gemm: A2 needs 0 < k_split < K and k_split %% %d == 0 (got %d)", BK, d->k_split);
    const int nkt = cdiv(d->K, BK);
    int split = d->split_k > 1 ? d->split_k : 1;
    if (split > nkt) split = nkt;
    if (split > 1) GAOT_REQUIRE(d->workspace != nullptr, "gemm: split_k > 1 needs a workspace");
    if (d->colsum) GAOT_REQUIRE(d->a_kmajor == 0 && d->A2 == nullptr, "gemm: colsum needs an m-major A operand (a_kmajor = 0)");

    GAOT_REQUIRE(d->pieces == 0 || (d->pieces >= 2 && d->pieces <= 4), "gemm: pieces must be 0 / 3 (three bf16 pieces), 4 (two fp16 pieces) or 2 (two bf16 pieces), got %d", d->pieces);
    GAOT_REQUIRE(d->pieces != 4 || (d->a_absmax && d->b_absmax && (!d->A2 || d->a2_absmax)), "gemm: pieces = 4 (fp16 pieces) needs a_absmax and b_absmax (and a2_absmax with A2)");
    // the debug override (1: `--dtype bf16` bench variant; 2 / 3 forced for A/B runs) wins over the call's own precision
    int pieces = g_split_pieces_forced ? g_split_pieces : (d->pieces == 2 ? 2 : (d->pieces == 4 ? 4 : 3));
    if (pieces >= 4 && !(d->a_absmax && d->b_absmax)) pieces = 3;

    GemmArgs a;
    a.M = d->M; a.N = d->N; a.K = d->K;
    a.A = d->A; a.lda = d->lda; a.A2 = d->A2; a.lda2 = d->lda2; a.k_split = d->k_split;
    a.B = d->B; a.ldb = d->ldb; a.C = d->C; a.ldc = d->ldc;
    a.bias = d->bias; a.rowbias = d->rowbias; a.rb_period = d->rowbias_period; a.ld_rb = d->ld_rowbias;
    a.rowscale = d->rowscale; a.act = d->act; a.aux_in = d->aux_in; a.aux_out = d->aux_out; a.ld_aux = d->ld_aux;
    a.residual = d->residual; a.ldr = d->ldr;
    a.split_k = split; a.ktiles_per_split = cdiv(nkt, split); a.ws = d->workspace;
    a.colsum = d->colsum;
    a.ablate = g_ablate;
    a.a_amax = pieces >= 4 ? d->a_absmax : nullptr; a.b_amax = pieces >= 4 ? d->b_absmax : nullptr; a.c_amax = d->c_absmax;
    a.a2_amax = pieces >= 4 ? d->a2_absmax : nullptr;
    // B pre-split into fp16 planes (weights, once per pass): only the split tiles with fp16 pieces read them
    a.Bpl = nullptr; a.ld_bpl = 0; a.bpl_stride = 0; a.bpl_flag = 0;
    if (pieces >= 4 && d->b_planes != nullptr && g_use_planes && aligned16(d->b_planes) && d->ld_bplanes % 8 == 0 && d->b_plane_stride == 16 &&
        d->K % 16 == 0) {
        a.Bpl = reinterpret_cast<const unsigned short*>(d->b_planes); a.ld_bpl = d->ld_bplanes; a.bpl_stride = d->b_plane_stride;
    }
    {
        auto ok4 = [](const void* ptr, long ld) { return ptr == nullptr || (aligned16(ptr) && ld % 4 == 0); };
        a.vec_epi = (a.N % 4 == 0) && aligned16(a.C) && (a.ldc % 4 == 0) && ok4(a.bias, 4) && ok4(a.rowbias, a.ld_rb) &&
                    ok4(a.aux_in, a.ld_aux) && ok4(a.aux_out, a.ld_aux) && ok4(a.residual, a.ldr) && ok4(a.ws, 4);
    }
    a.split_k = cdiv(nkt, a.ktiles_per_split);  // no empty splits
    const bool raw = d->raw_slabs != 0;
    if (raw) {
        a.c_amax = nullptr;
        if (a.split_k <= 1) { a.C = a.ws; a.ldc = a.N; a.vec_epi = 1; }      // a reduction too short to cut: the product itself is slab 0
        else if (a.C == nullptr) a.C = a.ws;                                  // (never written)
    }

    const bool ak = d->a_kmajor != 0, bk = d->b_kmajor != 0;
    // 16-byte vector path: contiguous extents and leading dims multiples of 4 floats, bases 16B aligned
    bool vec = aligned16(d->A) && aligned16(d->B) && (d->lda % 4 == 0) && (d->ldb % 4 == 0);
    vec = vec && (ak ? (d->K % 4 == 0) : (d->M % 4 == 0)) && (bk ? (d->K % 4 == 0) : (d->N % 4 == 0));
    if (d->A2) vec = vec && aligned16(d->A2) && (d->lda2 % 4 == 0) && ((d->K - d->k_split) % 4 == 0);

    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (a.act == GAOT_ACT_SWIGLU) {          // only the LDS-staged kernels lay the gate's bands out
        const long nb128 = (long)cdiv(a.M, 128) * cdiv(a.N, 128);
        a.vec_epi = 1;
        if (g_use_split == 2 || (g_use_split && nb128 >= 256)) {
            g_last_path = 3;
            if (dry) return GAOT_OK;
            const bool big = a.M >= 256 && (g_split_bm256 == 2 || (g_split_bm256 == 1 && nb128 >= 500));
            launch_split(a, ak, bk, st, big && pieces != 1 ? 256 : 128, pieces);
        }
        else { g_last_path = 1; if (dry) return GAOT_OK; launch_glds(a, ak, bk, nb128 >= 512 ? 1 : ((long)cdiv(a.M, 128) * cdiv(a.N, 64) >= 512 ? 2 : 3), st); }
        GAOT_CHECK_LAUNCH("gaot_gemm_f32(swiglu)");
        return GAOT_OK;
    }
    if (g_tile_override == 0 && !raw && dry && skinny_would(a, ak, bk)) { g_last_path = 2; return GAOT_OK; }
    // kernels without the vector epilogue do not publish C's magnitude word themselves: one absmax launch over C as stored
    auto publish_after = [&]() -> int {
        if (d->c_absmax == nullptr) return GAOT_OK;
        const int32_t cols = a.act == GAOT_ACT_SWIGLU_BWD ? 2 * a.N : a.N;
        gaot_absmax_item it = {a.C, (int64_t)a.ldc, a.M, cols, d->c_absmax};
        return gaot_absmax_grouped(&it, 1, stream);
    };
    if (g_tile_override == 0 && !raw && !dry && launch_skinny(a, ak, bk, st)) {
        g_last_path = 2;
        GAOT_CHECK_LAUNCH("gaot_gemm_f32(skinny)");
        return publish_after();
    }
    g_last_path = 1;
    // tile choice: the largest tile that still gives every CU (256) a workgroup; skinny N gets a 128x32 tile
    const long z = a.split_k;
    auto blocks = [&](int bm, int bn) { return (long)cdiv(a.M, bm) * cdiv(a.N, bn) * z; };
    (void)z;
    const bool glds_ok = g_use_glds && vec && a.vec_epi && a.K % 32 == 0 && a.M >= 4 && a.N >= 4;
    // the bf16 MFMA does not round its fp32 accumulator to nearest: same-signed increments drift (4096^3 with positive
    // operands: 1.1e-5 relative against 8e-7 on the fp32 MFMA, tools/acc_bias.py), so one workgroup accumulates at most 1 024
    // values of k on that pipe; longer reductions either arrive split (the slabs are summed on the vector pipe) or stay on the
    // fp32-MFMA tiles
    const long k_per_wg = a.split_k > 1 ? (long)a.ktiles_per_split * 32 : a.K;
    // [r6] unsplit reductions of 1 025 .. 2 048 on the 64 x 64 all-DMA tiles, which flush their accumulators every 1 024 values of k in the
    // kernel (gemm_ad.hip FL): the narrow outputs whose K slabs would not fill the split tiles and fell to the fp32-MFMA tiles (4 096 x 384 x
    // 1 152, the 3-D configuration's q|k|v input gradient: two slabs of 39 us + a reduce)
    const bool long_k = g_ad_flush && pieces == 4 && a.split_k <= 1 && a.K > 1024 && a.K <= 2048 && !raw;
    const bool split_ok = g_use_split && g_use_split != 3 && glds_ok && (a.A2 == nullptr || (ak && a.k_split % 16 == 0)) && g_tile_override == 0 &&
                          (k_per_wg <= 1024 || pieces == 1 || g_use_split != 1 || long_k);      // forced tuning modes bypass the cap
    // measured (round 2, tools/gemm_bench.py sweeps): the 128x128 split-bf16 tiles win once they fill the chip (>= 256 workgroups
    // counting split-K slabs) on outputs at least one tile wide; narrower / smaller products stay on the fp32 MFMA tiles
    // two-piece products, outputs at most six 128-wide tiles across (N <= 768): the 64-row tiles win although the 128-row ones would
    // fill the chip (8192 x 768 x 256 NT: 21.7 vs 24.2 us, NN x 512 x 256: 18.6 vs 19.5; N >= 1 024: the other way round)
    const bool two_pl = pieces == 2 || pieces >= 4;       // two planes per operand in LDS
    const bool prefer64 = split_ok && !long_k && g_use_split == 1 && two_pl && ak && cdiv(a.N, 128) <= 6 && blocks(64, 128) >= 250 && a.M >= 64 &&
                          a.N >= 128 && a.split_k <= 1;
    const bool split128 = split_ok && !long_k && g_use_split != 4 && !prefer64 &&
                          (g_use_split == 2 || (blocks(128, 128) >= 250 && a.M >= 128 && a.N >= 128 && (long)cdiv(a.M, 128) * cdiv(a.N, 128) >= 8));
    // outputs only a few 128-wide tiles across (N = 256): 64-row tiles double the workgroup count
    // (with pre-split B planes an NN product stages B exactly like an NT one; with two-piece products the 64-row split tiles beat the
    // fp32-MFMA tiles on the NN products too: 8192 x 256 x 768 37.4 -> 25.6 us, x 512 26.5 -> 19.0, x 256 15.5 -> 12.8, tools/gemm_modes_2p.py)
    const bool split64 = split_ok && !long_k && !split128 && g_use_split != 5 && g_use_split != 2 &&
                         (g_use_split == 4 || prefer64 || ((ak && (bk || a.Bpl != nullptr || g_use_split == 6 || two_pl)) && blocks(64, 128) >= 250 && a.M >= 64 && a.N >= 128 && a.split_k <= 1));   // measured: NT +6-14 %, NN +-0
    // 64 x 64 all-DMA tiles (gaot_debug_set_gemm_ad_narrow, bits): (1, default) half-filled launches of outputs two tiles wide (4 096 tokens x
    // 256: 128 workgroups of 64 rows, which fall to the fp32-MFMA tiles) get 256 workgroups -- tools/ad_bench.hip: 4096 x 256 x 256 in 8.3 us
    // against ~15; same-box step A/B at the 4 096-token batch (tools/step_ab.py --c4, twice): 1.6486 -> 1.6249, 1.6481 -> 1.6234 ms, C2
    // untouched (its launches are full).  (2, off) N <= 256 with K <= 256 at 8 192 rows: 11.9 -> 11.2 us alone, but C2 2.1499 -> 2.1663,
    // 2.1399 -> 2.1705 ms per step.  Bit-identical to the other fp16-piece tiles in every test (tests/test_ops_gpu.py, tools/ad_stress.py);
    // a product that (1) moves off the fp32-MFMA tiles is rounded as the fp16-piece family rounds (same error class, other last bits).
    // (An earlier series of A/B runs looked inconsistent and twice ended at an unexplained loss: that was the host-side slot bookkeeping,
    // DESIGN 7, not these tiles.)
    const bool narrow_shape = g_use_ad == 1 && (g_ad_narrow & 1) && split_ok && !split128 && !split64 && pieces == 4 && ak && a.K % 32 == 0 && a.vec_epi &&
                              (a.split_k <= 1 || ((g_ad_narrow & 8) && !raw)) && (a.A2 == nullptr || ((g_ad_narrow & 16) && a.k_split % 32 == 0)) && (cdiv(a.N, 128) == 2 || ((g_ad_narrow & 4) && cdiv(a.N, 128) == 3)) && blocks(64, 64) >= 128 && (long)cdiv(a.M, 64) * cdiv(a.N, 128) < 250 &&
                              (a.split_k <= 1 ? a.K : cdiv(cdiv(a.K, 32), a.split_k) * 32) <= (long_k ? 2048 : 1024);          // (bit 8: K slabs of such a product too -- 4 096 x 384 x 1 152 in two slabs)
    const bool ad_narrow = narrow_shape && (dry ? d->b_planes != nullptr : a.Bpl != nullptr);
    // planes handed in but switched off (gaot_debug_set_gemm_planes(0)): the same tile family on the staged 64-row kernel, so that the
    // switch changes where B's pieces come from and nothing else (products WITHOUT planes -- B an activation -- stay where they were)
    const bool narrow_staged = narrow_shape && !long_k && !dry && a.Bpl == nullptr && d->b_planes != nullptr;
    if (dry) { g_last_path = (split128 || split64 || ad_narrow) ? 3 : 1; return GAOT_OK; }
    if (split128 || split64 || ad_narrow || narrow_staged) {
        g_last_path = 3;
        // 256x128 (8-wave) tiles: measured +3-7 % on the NT / TN products that still give ~200 workgroups, -2 % on NN
        const bool big = split128 && a.M >= 512 && (g_split_bm256 == 2 || (g_split_bm256 == 1 && ak == bk && blocks(256, 128) >= 190));
        // all-DMA tiles (gemm_ad.hip; tools/ad_bench.hip, same box, us staged -> all-DMA): outputs two tiles wide (N <= 256) on 64-row
        // tiles: 8192 x 256 x 256 13.5 -> 11.8, x 768 26.7 -> 22.1, x 1024 32.8 -> 26.8; wider outputs on 64-row tiles while those fit
        // one round of two workgroups per CU (the 4 096-token batches), else on 128-row ones while THOSE fit one round; what needs more
        // rounds (8192 x 2048 x 256: 49 -> 51-54 us; x 768: 23.2 -> 26.8 on 64-row tiles) and the K-slab products (8192 x 256 x 2048 in
        // two slabs: 42.9 -> 42.6) stay on the staged kernel -- there the output stores / the slab count decide.  Step level
        // (tools/step_ab.py, same box): C2 2.3225 -> 2.2666 ms, C4-shaped 1.8169 -> 1.7888
        const bool ad_ok = g_use_ad && pieces == 4 && a.Bpl != nullptr && ak && a.K % 32 == 0 && a.vec_epi && (a.A2 == nullptr || a.k_split % 32 == 0);
        const bool ad64 = ad_ok && (g_use_ad == 2 || (a.split_k <= 1 && (cdiv(a.N, 128) <= 2 || blocks(64, 128) <= 512) && blocks(64, 128) >= 128));
        const bool ad128 = ad_ok && !ad64 && !big && (g_use_ad == 3 || (a.split_k <= 1 && split128 && blocks(128, 128) <= 512));
        if (ad_narrow) launch_ad(a, bk, st, 64, 64);
        else if (ad64) launch_ad(a, bk, st, 64, ((g_ad_narrow & 2) && a.K <= 256 && cdiv(a.N, 128) <= 2) ? 64 : 128);      // (8192 x 256 x 256: 11.9 -> 11.2 us on 64 x 64 tiles)
        else if (ad128) launch_ad(a, bk, st, 128);
        else
        launch_split(a, ak, bk, st, (split64 || narrow_staged) ? 64 : (big && pieces != 1 ? 256 : 128), pieces);
    }
    else if (glds_ok && g_tile_override >= 0 && g_tile_override <= 3) {
        int tile = g_tile_override;
        if (tile == 0) {      // from the on-box sweep (round 2, tools/gemm_bench.py; 2-stage ring)
            if (!ak && !bk) tile = blocks(128, 128) >= 256 ? 1 : (blocks(128, 64) >= 256 && a.M >= 128 ? 2 : 3);
            else if (blocks(128, 128) >= 512) tile = 1;
            else if (blocks(128, 64) >= 512) tile = 2;
            else tile = 3;
        }
        launch_glds(a, ak, bk, tile, st);
    }
    else if (g_tile_override == 1)      launch_cfg<128, 128, 2>(a, ak, bk, vec, st);
    else if (g_tile_override == 2)      launch_cfg<128, 64, 2>(a, ak, bk, vec, st);
    else if (g_tile_override == 3)      launch_cfg<64, 64, 2>(a, ak, bk, vec, st);
    else if (g_tile_override == 4)      launch_cfg<128, 32, 4>(a, ak, bk, vec, st);
    else if (g_tile_override == 5)      launch_cfg<128, 128, 2, 8>(a, ak, bk, vec, st);
    else if (g_tile_override == 6)      launch_cfg<128, 64, 4, 8>(a, ak, bk, vec, st);
    else if (a.N <= 32)                 launch_cfg<128, 32, 4>(a, ak, bk, vec, st);
    // measured on MI355X (tools/gemm_bench.py): short reductions want SEVERAL block-waves in flight so that the
    // load / MFMA / store phases of different workgroups overlap; big tiles only pay off on large outputs.
    // transposed-A weight gradients (long reduction, split-K): 128x64 measured 10-17 % ahead of 64x64 once it fills the chip
    else if (!ak && !bk && blocks(128, 64) >= 256 && a.M >= 128) launch_cfg<128, 64, 2>(a, ak, bk, vec, st);
    else if (blocks(64, 64) <= 1536)    launch_cfg<64, 64, 2>(a, ak, bk, vec, st);
    else if (blocks(128, 64) <= 3072)   launch_cfg<128, 64, 2>(a, ak, bk, vec, st);
    else                                launch_cfg<128, 128, 2>(a, ak, bk, vec, st);
    GAOT_CHECK_LAUNCH("gaot_gemm_f32");
    if (a.split_k > 1 && !raw) {
        const long total = (long)a.M * a.N;
        const bool plain = !a.bias && !a.rowbias && !a.rowscale && !a.aux_out && a.act == GAOT_ACT_NONE &&
                           (!a.residual || (aligned16(a.residual) && a.ldr % 4 == 0)) &&
                           a.N % 4 == 0 && a.ldc % 4 == 0 && aligned16(a.C) && aligned16(a.ws);
        if (plain) {
            const long nb = cdiv(total / 4, 64) + (a.colsum ? cdiv(a.M, 64) : 0);
            hipLaunchKernelGGL(splitk_reduce_vec_kernel, dim3((unsigned)nb), dim3(256), 0, st, a);
        } else {
            int nb = cdiv(total, 256);
            if (nb > 2048) nb = 2048;
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3(nb), dim3(256), 0, st, a);
        }
        GAOT_CHECK_LAUNCH("gaot_gemm_f32(split-k reduce)");
    }
    else if (!a.vec_epi && !raw) return publish_after();
    return GAOT_OK;
}

extern "C" int32_t gaot_gemm_slab_count(int32_t K, int32_t split_k) {
    const int nkt = cdiv(K > 0 ? K : 1, BK);
    int split = split_k > 1 ? split_k : 1;
    if (split > nkt) split = nkt;
    return cdiv(nkt, cdiv(nkt, split));
}


extern "C" int gaot_gemm_f32(const gaot_gemm_desc* d, gaot_stream_t stream) { return gemm_run(d, stream, false); }
// which kernel family WOULD serve this product (as gaot_debug_last_gemm_path: 1 fp32-MFMA tiles, 2 skinny, 3 split tiles on the
// bf16 / fp16 matrix pipe); launches nothing.  Callers use it to decide whether the fp16-piece operands' absmax words are needed.
extern "C" int gaot_gemm_path(const gaot_gemm_desc* d) {
    const int keep = g_last_path;
    const int rc = gemm_run(d, nullptr, true);
    const int path = rc == GAOT_OK ? g_last_path : -1;
    g_last_path = keep;
    return path;
}

// ---- grouped weight-gradient products (kernel: gemm_split.hip)
namespace gaot { void set_tn_kslab(int k); void set_tn_bm(int bm); void set_tn_rule(int r); }
extern "C" int gaot_debug_set_wgrad_slab_rule(int r) { gaot::set_tn_rule(r); return 0; }
extern "C" int gaot_debug_set_wgrad_kslab(int k) { gaot::set_tn_kslab(k); return 0; }
extern "C" int gaot_debug_set_wgrad_tile_rows(int bm) { gaot::set_tn_bm(bm); return 0; }
static int check_wgrad_items(const gaot_wgrad_item* items, int n) {
    GAOT_REQUIRE(items != nullptr && n > 0, "gemm_tn_grouped: no items");
    for (int i = 0; i < n; ++i) {
        const gaot_wgrad_item& it = items[i];
        GAOT_REQUIRE(it.g && it.x && it.out, "gemm_tn_grouped: item %d has a null operand", i);
        GAOT_REQUIRE(it.M >= 4 && it.N >= 4 && it.K >= 32 && it.M % 4 == 0 && it.N % 4 == 0 && it.K % 32 == 0,
                     "gemm_tn_grouped: item %d needs M, N %% 4 == 0 and K %% 32 == 0 (got %d, %d, %d)", i, it.M, it.N, it.K);
        GAOT_REQUIRE(it.ldg % 4 == 0 && it.ldx % 4 == 0 && it.ldo % 4 == 0 && it.ldg >= it.M && it.ldx >= it.N && it.ldo >= it.N &&
                     aligned16(it.g) && aligned16(it.x) && aligned16(it.out),
                     "gemm_tn_grouped: item %d needs 16-byte aligned operands and leading dimensions that are multiples of 4", i);
    }
    return GAOT_OK;
}

extern "C" int64_t gaot_gemm_tn_grouped_workspace(const gaot_wgrad_item* items, int32_t n, int32_t* n_counters) {
    if (check_wgrad_items(items, n) != GAOT_OK) return -1;
    long ws = 0; int cnt = 0;
    for (int i0 = 0; i0 < n; i0 += TN_GROUP_MAX) {          // launches of at most TN_GROUP_MAX products share the buffers: sizes add up
        int c = 0, c2 = 0;          // sized for either tile height (the launch picks by `pieces`, which this query does not know)
        const int m = n - i0 < TN_GROUP_MAX ? n - i0 : TN_GROUP_MAX;
        const long w1 = plan_tn_grouped(items + i0, m, nullptr, &c, nullptr, 128), w2 = plan_tn_grouped(items + i0, m, nullptr, &c2, nullptr);
        ws += w1 > w2 ? w1 : w2;
        cnt += c > c2 ? c : c2;
    }
    if (n_counters) *n_counters = cnt;
    return ws;
}

extern "C" int gaot_gemm_tn_grouped(const gaot_wgrad_item* items, int32_t n, int32_t pieces, float* workspace, int32_t* counters, gaot_stream_t stream) {
    GAOT_REQUIRE(pieces == 0 || (pieces >= 2 && pieces <= 4), "gemm_tn_grouped: pieces must be 0 / 3, 4 (two fp16 pieces) or 2, got %d", pieces);
    if (gaot_forced_pieces() >= 2) pieces = gaot_forced_pieces();
    if (int rc = check_wgrad_items(items, n)) return rc;
    if (pieces >= 4) for (int i = 0; i < n; ++i) if (!items[i].g_absmax || !items[i].x_absmax) { pieces = 3; break; }
    int cnt = 0;
    const long need = gaot_gemm_tn_grouped_workspace(items, n, &cnt);
    GAOT_REQUIRE((need == 0 || (workspace != nullptr && aligned16(workspace))) && (cnt == 0 || counters != nullptr),
                 "gemm_tn_grouped: needs a 16-byte aligned workspace of %ld floats and %d zeroed counters", need, cnt);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    long ws_off = 0; int cnt_off = 0;
    for (int i0 = 0; i0 < n; i0 += TN_GROUP_MAX) {
        const int m = n - i0 < TN_GROUP_MAX ? n - i0 : TN_GROUP_MAX;
        int c = 0, c2 = 0;
        const long w1 = plan_tn_grouped(items + i0, m, nullptr, &c, nullptr, 128), w2 = plan_tn_grouped(items + i0, m, nullptr, &c2, nullptr);
        const long w = w1 > w2 ? w1 : w2;
        c = c > c2 ? c : c2;
        launch_tn_grouped(items + i0, m, workspace + ws_off, counters + cnt_off, pieces >= 4 ? 4 : (pieces == 2 ? 2 : 3), st);
        GAOT_CHECK_LAUNCH("gaot_gemm_tn_grouped");
        ws_off += w; cnt_off += c;
    }
    return GAOT_OK;
}
