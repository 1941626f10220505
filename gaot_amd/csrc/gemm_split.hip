// fp32 GEMM on the f16 / bf16 matrix pipe.  Default (NP = 4, round 4): two fp16 pieces of the power-of-two scaled operand, three piece
// products on v_mfma_f32_32x32x16_f16 (common.h split2h_pair has the error argument; the scales come from device-resident magnitude
// words).  The original form, still what runs without magnitude words (NP = 3): every fp32 operand is split EXACTLY into three bf16 pieces
//     x = x1 + x2 + x3        (8 + 8 + 8 significant bits, by truncation: h = x & 0xffff0000, r = x - h, ...)
// and the product is accumulated in fp32 from the six piece products whose weight is >= 2^-16:
//     x*y ~= x1y1 + (x1y2 + x2y1) + (x1y3 + x2y2 + x3y1)          (dropped: x2y3 + x3y2 + x3y3 <= 2^-23 |x||y|)
// i.e. the result carries fp32-level error (one fp32 rounding is 2^-24) while running on
// v_mfma_f32_32x32x16_bf16, whose rate is 16x the fp32 MFMA's: six of them per fp32 product = 2.67x the fp32-MFMA
// peak (157 -> ~400 TFLOP/s effective on MI355X).  The split is done ONCE per element per workgroup, on the way from
// the global-load registers into LDS; LDS holds three bf16 planes per operand.
//
// Tile: 128x128 per workgroup, 4 waves of 64x64 (2x2 MFMA tiles -> every 16-byte operand read feeds 2 MFMAs x 3),
// k-tile 16, two LDS stages, one barrier per k-tile.  LDS plane layouts (bf16):
//   k-contiguous operand : [row][16 k], row stride 48 B (conflict-free ds_read_b128: lane (i, kh) reads 8 k)
//   row-contiguous operand: [k pair][row] of packed (k, k+1) dwords (ds_write_b128 of 4 rows, 4 x ds_read_b32)
// The reduction order inside a k-tile is free, and both operands use the same one.
#include "gemm_common.h"

namespace gaot {


constexpr int SBK = 16;                 // k per tile
constexpr int S_BM = 128, S_BN = 128;
constexpr int S_PLANE = 128 * 48;       // bytes per plane (k-contiguous layout is the larger one)
constexpr int S_STAGE = 6 * S_PLANE;    // 3 planes x 2 operands

// two fp32 values rounded to nearest-even bf16, packed (element 0 in the low half)
__device__ __forceinline__ unsigned rne_pair(float x0, float x1) {
    unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
    u0 += 0x7fffu + ((u0 >> 16) & 1u);
    u1 += 0x7fffu + ((u1 >> 16) & 1u);
    return __builtin_amdgcn_perm(u1, u0, 0x07060302u);
}
// NP = 4: two fp16 pieces of sc * x (sc a power of two that brings the operand's largest magnitude to ~2^14, see split2h_pair): three
// piece products h h + h m + m h on v_mfma_f32_32x32x16_f16 (the fourth, m m <= 2^-22 of the term, was measured: 1.87e-7 instead of
// 1.90e-7 against float64 on random data -- the fp32 accumulation dominates -- at +7 % time: not taken)
template <int NP, int ABL>
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& ph, unsigned& pm, unsigned& pl, float sc = 1.f) {
    if (NP == 1) { ph = rne_pair(x0, x1); pm = pl = 0u; }
    else if (NP == 2) { split2_pair(x0, x1, ph, pm); pl = 0u; }       // two ROUNDED pieces: x = h + m + e, |e| <= 2^-18 |x|, unbiased
    else if (NP >= 4) { split2h_pair_gemm(x0, x1, sc, ph, pm); pl = 0u; }
    else split3_pair<ABL>(x0, x1, ph, pm, pl);
}

template <int ABL, bool F16 = false>
__device__ __forceinline__ f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) {
    if (ABL & 2) { c[0] += __builtin_bit_cast(u32x4, a)[0] * 1e-30f + __builtin_bit_cast(u32x4, b)[1] * 1e-30f; return c; }
    if (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// ABL (tuning builds only, tools/split_bench.hip): 1 = no split arithmetic (raw words to LDS), 2 = no MFMA, 4 = no
// epilogue stores, 8 = no LDS fragment reads, 16 = no global loads in the loop, 32 = no LDS plane writes, 64 = no barrier per k-tile
// BM = 128 (4 waves of 64x64) or 64 (4 waves of 32x64: twice the workgroups for outputs only two tiles wide, N = 256)
// BM = 256: 8 waves (4 x 2) of 64x64, one workgroup per CU: a quarter less split work and LDS write traffic per MFMA
// NP = 3: the fp32-level product (three pieces per operand, six piece products).  NP = 1: a plain bf16 GEMM on the same skeleton
// (operands ROUNDED to nearest-even bf16, one piece product, fp32 accumulation) -- the separately reported `--dtype bf16` bench
// variant (BASELINE configs[1] names bf16); never used by the fp32 parity path.
#ifndef GAOT_SPLIT2_WG_PER_CU
#define GAOT_SPLIT2_WG_PER_CU 3        // two-piece tiles: workgroups per CU the kernels are compiled for (A/B builds: 2)
#endif
template <int BM, int NP = 3>
struct SplitGeom {
    static constexpr int NPL = (NP == 2 || NP >= 4) ? 2 : 3;            // planes per operand in a stage (two-piece tiles: 48 KB, three workgroups per CU)
    static constexpr int BN = S_BN, NW = BM == 256 ? 8 : 4, NT = 64 * NW, WAVES_N = 2, WM = BM / (NW / 2), WN = 64, TM = WM / 32, TN = 2;
    static constexpr int PA = (BM > 128 ? BM : 128) * 48, PB = 128 * 48;      // bytes per plane
    static constexpr int STAGE = NPL * PA + NPL * PB;
    static constexpr int EPI_BYTES = NW * 32 * (WN + 4) * 4;
    static constexpr int SMEM_BYTES = 2 * STAGE > EPI_BYTES ? 2 * STAGE : EPI_BYTES;
};

// ---- Dynamic range of the fp16 pieces (NP = 4).  One power of two per TENSOR brings the operand's largest magnitude to [2^13, 2^14);
// an element v of the scaled operand is carried as h + m with |v - h - m| <= max(2^-23 |v|, 2^-25): m goes subnormal below |v| ~ 2^-3, so
// elements more than ~2^16 below the tensor's largest carry an ABSOLUTE error F = 2^-25 / scale instead of a relative one.  For one
// output y = sum_i a_i b_i the floors add  F_a sum |b_i| + F_b sum |a_i|  to the error 2^-22 sum |a_i b_i| that fp32-level products
// carry anyway; with t_a, t_b the typical magnitudes of the row of A and the row of B being contracted, the floors stay below that
// as long as   (max_a / t_a) x (max_b / t_b) <= ~2^16.5   (max = the TENSOR's largest magnitude, which sets the scale).
// Normalised activations, weights and gradients sit at 2^3 .. 2^10 per operand (measured on the bench step: tools/redo_count.py); what
// breaks the bound is the "massive activation" pattern -- one token or one channel 10^4 .. 10^6 times larger than the rest, or rows /
// channels that much smaller -- where every ordinary row sits 2^20 below the tensor's maximum.
// So every workgroup watches what it stages: per thread and k-tile the largest magnitude of its 8 elements (8 consecutive k of one
// row, or 2 k x 4 rows), and over its k range the SUM OF THE EXPONENTS of the non-zero group maxima (their geometric mean: the "typical"
// magnitude of the thread's row, robust against the outliers themselves) -- five vector instructions per operand and k-tile.  At the end
// L = log2(scaled tensor maximum / that mean) per thread, and if L_a + L_b exceeds 17 for an eighth of the workgroup's threads -- or
// a mean sits ABOVE the tensor's claimed maximum: a stale word -- the workgroup computes its tile AGAIN on the fp32 MFMA, straight from
// the fp32 operands (no pieces, no scales: exact for any spread; one LDS stage, a plain loop: slow, and rare).  Exact zeros are exempt
// (empty latent tokens, ReLU outputs).  Pre-split weight planes carry their L in the weight's magnitude word
// (gaot_split_f16_planes_grouped: float 1 = along the rows, float 2 = along the columns of the matrix as stored).
// Where a workgroup stages BOTH operands in the same layout (the weight gradients dY^T X; x W^T without planes) a thread's groups of A
// and of B cover the same contraction indices, and it evaluates the bound itself instead of the two means: sum over its slices of
// max(g_a g_b, floor_a g_b, floor_b g_a) against 4 sum g_a g_b (`PAIRED` below) -- a slice counts with the weight it has in the output.
__device__ unsigned g_split_redo_tiles = 0;          // tiles that took the second pass since the last gaot_debug_split_redo_count(1)

// ONE output tile (`logical` in the XCD-aware order of the caller, K slab `zs` of p.split_k) of the product described by p.
// Shared by the per-product kernel below and by the grouped weight-gradient kernel (one launch over many products).
// FLUSH > 0: every FLUSH k-tiles (16 k each) the MFMA accumulators are added to running sums on the VECTOR pipe and restart from
// zero: the bf16 MFMA does not round its accumulator to nearest, so one matrix-pipe accumulation run stays at FLUSH * 16 <= 1 024
// values of k however long the workgroup's reduction is (costs TM * TN * 16 more registers: the grouped TN kernel has them)
// BPL (NP = 4, B staged k-contiguous): the B operand arrives ALREADY split -- the two fp16 pieces of the scaled weight, k-contiguous in
// groups of 16: element (n, k) of piece q at Bpl[n * ld_bpl + (k / 16) * 32 + q * 16 + k % 16], built once per pass from the same magnitude word the kernel
// reads its inverse scale from (gaot_split_f16_planes_grouped): B tiles go from the load registers to LDS as they are -- no
// vector arithmetic for B at all (half of the k-loop's split work), same number of loads and LDS writes, bit-identical products.
struct BRegs { f32x4 f[2]; u32x4 pl[2]; };
// NP = 4 watches the operands' range while it stages them and, when the tile needs it, computes it a second time on the fp32 MFMA (above).
// BREAL (with BPL, where BKM describes the PLANES and is true): is B itself k-contiguous in memory?  (NT: yes, the planes of W as stored;
// NN: no, the planes of W^T) -- what the second pass reads
template <bool AK, bool BKM, int BM, int ABL, int NP, int FLUSH = 0, bool BPL = false, bool BREAL = BKM>
__device__ __forceinline__ void split_tile(const GemmArgs& p, unsigned char* smem_raw, const int logical, const int zs) {
    using G = SplitGeom<BM, NP>;
#ifndef GAOT_NO_DETECT
    constexpr bool DETECT = NP == 4 && ABL == 0;
#else
    constexpr bool DETECT = false;          // A/B builds: without the range tracking (and without the second pass)
#endif
    // per operand ONE register: sum of the biased exponents of this thread's non-zero group maxima (low 20 bits: < 4 096 k-tiles x 255)
    // and their count (high 12 bits)
    unsigned ec_a = 0u, ec_b = 0u;
    // PAIRED tracking (both operands staged here AND laid out alike, so that a thread's group of A and its group of B hold the SAME
    // contraction indices: the weight-gradient products dY^T X, and x W^T without planes): per k-tile the slice's weight w = g_a g_b (the
    // groups' largest magnitudes) and its error in units of 2^-22 w: max(1, floor_a / g_a, floor_b / g_b), floor = 2^-3 / scale (below it
    // the second piece goes subnormal).  The tile is suspect when the floors add up to more than four times the rounding the
    // products carry anyway: sum max(w, floor_a g_b, floor_b g_a) > 4 sum w.  A tiny gradient row that meets an ordinary activation
    // row weighs nothing in either sum (thousands of empty latent tokens on the 3-D cloud: 230 tiles per step were redone before this
    // form); a tiny gradient row that meets a MASSIVE activation row -- what RMSNorm's backward makes of a massive token -- does.
    constexpr bool PAIRED = !BPL && AK == BKM;
    float trk_w = 0.f, trk_r = 0.f, trk_m = 0.f, thr_a = 0.f, thr_b = 0.f;          // (trk_m: largest scaled magnitude -- a stale word)
    constexpr int NPL = G::NPL;
    constexpr int BN = G::BN, NW = G::NW, NT = G::NT, WAVES_N = G::WAVES_N, WM = G::WM, WN = G::WN, TM = G::TM, TN = G::TN;
    constexpr int ABYTES = BM * 4;       // row stride of a row-contiguous A plane ([k pair][BM rows] of packed dwords; BM = 64 only)
    constexpr int PA = G::PA, PB = G::PB, STAGE = G::STAGE;
    constexpr bool A_FULL = BM * 2 == NT;        // two float4 per thread (else one / half the threads)
    constexpr bool B_FULL = 128 * 2 == NT;
    // LANES4: with pre-split weight planes, 128-row tiles fetch both operands with FOUR lanes per row (64 contiguous bytes: a row's whole
    // k-tile) and two rows 64 apart per thread, instead of two lanes per row and 32 bytes each: a wave instruction then covers 16 rows x 64 B
    // instead of 64 separate 16-byte pieces, which the texture addresser takes one at a time (tools/ad_bench.hip: that address path, not
    // the LDS or the matrix pipe, is what these tiles wait for at K = 256).  Same planes in LDS, same products: bit-identical.
    constexpr bool A4 = AK && A_FULL && BPL && NT == 256;
    constexpr bool B4 = BPL && B_FULL && NT == 256;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    const int m0 = (logical / p.tiles_n) * BM;
    const int n0 = (logical % p.tiles_n) * BN;

    const int nkt = p.K / SBK;
    int kt_begin = 0, kt_end = nkt;
    if (p.split_k > 1) {                     // ktiles_per_split counts 32-wide tiles
        kt_begin = zs * p.ktiles_per_split * 2;
        kt_end = min(nkt, kt_begin + p.ktiles_per_split * 2);
    }

    // ---- global -> register staging (2 float4 per operand per thread per k-tile)
    // k-contiguous: row = tid >> 1, k half = tid & 1 (8 k = two float4 q = 0, 1 -> one 16-byte write per plane)
    // row-contiguous (128-row tiles): kp = tid & 7 (k pair), r4 = tid >> 3 (4 rows); q = which k of the pair.  The packed
    // (k, k+1) dwords are written TRANSPOSED into the k-contiguous plane layout (4 x ds_write_b32 per plane), so every
    // operand is read back with ds_read_b128 whatever its layout in memory
    constexpr bool F16 = NP >= 4;
    static_assert(!BPL || (BKM && NP == 4), "pre-split B planes: fp16 pieces, k-contiguous");
    // fp16 pieces: power-of-two operand scales from the operands' magnitude words (uniform: two scalar loads per workgroup)
    float sc_a = 1.f, sc_b = 1.f, so_a = 1.f, so_b = 1.f;
    // (the words are read AFTER the first two tiles' loads have been issued, below: one round trip instead of two at the head of every
    // workgroup's life -- at K = 256 a workgroup lives for sixteen k-tiles)
    const float* a_src[2]; const float* b_src[2];
    // concatenated input (k-contiguous A only): columns k >= k_split come from A2 -- as an element offset from this thread's A row
    long a2_delta = 0, a2_delta1 = 0;          // (a2_delta1: the second row of a LANES4 thread)
    if (AK && p.A2 != nullptr) {
        const long row = min(m0 + (A4 ? tid >> 2 : (A_FULL ? tid >> 1 : tid >> 2)), p.M - 1);
        a2_delta = (reinterpret_cast<long>(p.A2) - reinterpret_cast<long>(p.A)) / 4 + row * (p.lda2 - p.lda) - p.k_split;
        if (A4) a2_delta1 = (reinterpret_cast<long>(p.A2) - reinterpret_cast<long>(p.A)) / 4 + min(m0 + 64 + (tid >> 2), p.M - 1) * (long)(p.lda2 - p.lda) - p.k_split;
    }
    const unsigned short* bp_src = nullptr;
    const unsigned short* bp_src4[2] = {nullptr, nullptr};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        if (AK) {
            if (A4) a_src[q] = p.A + (long)min(m0 + 64 * q + (tid >> 2), p.M - 1) * p.lda + (tid & 3) * 4;
            else if (A_FULL) a_src[q] = p.A + (long)min(m0 + (tid >> 1), p.M - 1) * p.lda + (tid & 1) * 8 + q * 4;
            else        a_src[q] = p.A + (long)min(m0 + (tid >> 2), p.M - 1) * p.lda + (tid & 3) * 4;      // one float4 per thread (q = 0)
        } else {
            if (A_FULL) a_src[q] = p.A + (long)(2 * (tid & 7) + q) * p.lda + min(m0 + (tid >> 3) * 4, p.M - 4);
            else        a_src[q] = p.A + (long)(2 * ((tid >> 4) & 7) + q) * p.lda + min(m0 + (tid & 15) * 4, p.M - 4);   // BM = 64: threads 0..127
        }
        if (BKM) {
            const int row = B_FULL ? tid >> 1 : tid >> 2;
            int nrow = min(n0 + row, p.N - 1);
            if (p.act == GAOT_ACT_SWIGLU) {      // band layout [u1 cols | u3 cols] per wave band (epilogue_swiglu)
                const int F = p.N >> 1, within = row % WN;
                const int gcol = (n0 >> 1) + (row / WN) * (WN / 2) + within % (WN / 2);
                nrow = (within / (WN / 2)) * F + min(gcol, F - 1);
            }
            b_src[q] = B_FULL ? p.B + (long)nrow * p.ldb + (tid & 1) * 8 + q * 4 : p.B + (long)nrow * p.ldb + (tid & 3) * 4;
            if (BPL) bp_src = B_FULL ? p.Bpl + (long)nrow * p.ld_bpl + (tid & 1) * 8 : p.Bpl + (long)nrow * p.ld_bpl + (tid & 3) * 4;
            if (B4) {          // row 64 q + (tid >> 2), 16-byte chunk tid & 3 of the k-group's [h 16 | m 16]
                const int row4 = 64 * q + (tid >> 2);
                int nrow4 = min(n0 + row4, p.N - 1);
                if (p.act == GAOT_ACT_SWIGLU) {
                    const int F = p.N >> 1, within = row4 % WN;
                    const int gcol = (n0 >> 1) + (row4 / WN) * (WN / 2) + within % (WN / 2);
                    nrow4 = (within / (WN / 2)) * F + min(gcol, F - 1);
                }
                bp_src4[q] = p.Bpl + (long)nrow4 * p.ld_bpl + (tid & 3) * 8;
            }
        } else {      // 128 rows: (k pair = tid & 7, row quad = tid >> 3) for the first 256 threads
            b_src[q] = p.B + (long)(2 * (tid & 7) + q) * p.ldb + min(n0 + ((tid >> 3) & 31) * 4, p.N - 4);
        }
    }
    // two register sets: tile j lives in set j & 1 (loads run two k-tiles ahead of the MFMAs, the split one ahead)
    f32x4 ra[2][2];
    BRegs rb[2];
    auto gload = [&](int kt, f32x4 (&xa)[2], BRegs& xb) {
        const long k0 = (long)min(kt, kt_end - 1) * SBK;      // past the end: re-load the last tile (never consumed)
        if ((ABL & 16) && kt > kt_begin + 1) return;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const bool a_live = A_FULL || (AK ? q == 0 : tid < 128);
            const bool b_live = B_FULL || (BKM ? q == 0 : tid < 256);
            if (a_live) xa[q] = *reinterpret_cast<const f32x4*>(AK ? a_src[q] + k0 + ((p.A2 != nullptr && k0 >= p.k_split) ? ((A4 && q == 1) ? a2_delta1 : a2_delta) : 0L) : a_src[q] + k0 * p.lda);
            else xa[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (!BPL) {
                if (b_live) xb.f[q] = *reinterpret_cast<const f32x4*>(BKM ? b_src[q] + k0 : b_src[q] + k0 * p.ldb);
                else xb.f[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        if (BPL) {
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                if (B4) xb.pl[pl] = *reinterpret_cast<const u32x4*>(bp_src4[pl] + k0 * 2);
                else if (B_FULL) xb.pl[pl] = *reinterpret_cast<const u32x4*>(bp_src + pl * 16 + k0 * 2);
                else { const u32x2 v = *reinterpret_cast<const u32x2*>(bp_src + pl * 16 + k0 * 2); xb.pl[pl] = u32x4{v[0], v[1], 0u, 0u}; }
            }
        }
    };

    // fused column sums of a row-contiguous A (bias gradient): this thread's 4 rows, its k pair
    const bool do_colsum = !AK && p.colsum != nullptr && (logical % p.tiles_n) == 0;
    f32x4 csum = {0.f, 0.f, 0.f, 0.f};

    // split + store one operand's registers into its three planes of `plane` bytes.  full: two float4 per thread; else one
    // float4 (k-contiguous) or the first `half_threads` threads with two (row-contiguous).  Row-contiguous operands are written
    // TRANSPOSED into the k-contiguous layout, except the 64-row A tile (legacy [k pair][row] layout, not picked by the heuristic).
    auto stage_store = [&](unsigned char* base, const f32x4 (&r)[2], bool kmajor, bool full, int plane, int half_threads, bool legacy64, float sc) {
        if (kmajor) {
            if (full) {
                u32x4 h, m, l;
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        unsigned a_, b_, c_;
                        split_pair<NP, ABL>(r[q][2 * e], r[q][2 * e + 1], a_, b_, c_, sc);
                        h[2 * q + e] = a_; m[2 * q + e] = b_; l[2 * q + e] = c_;
                    }
                unsigned char* dst = base + (tid >> 1) * 48 + (tid & 1) * 16;
                *reinterpret_cast<u32x4*>(dst) = h;
                if (NP >= 2) *reinterpret_cast<u32x4*>(dst + plane) = m;
                if (NP == 3) *reinterpret_cast<u32x4*>(dst + 2 * plane) = l;
            } else {
                u32x2 h, m, l;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    unsigned a_, b_, c_;
                    split_pair<NP, ABL>(r[0][2 * e], r[0][2 * e + 1], a_, b_, c_, sc);
                    h[e] = a_; m[e] = b_; l[e] = c_;
                }
                unsigned char* dst = base + (tid >> 2) * 48 + (tid & 3) * 8;
                *reinterpret_cast<u32x2*>(dst) = h;
                if (NP >= 2) *reinterpret_cast<u32x2*>(dst + plane) = m;
                if (NP == 3) *reinterpret_cast<u32x2*>(dst + 2 * plane) = l;
            }
        } else {
            if (!full && tid >= half_threads) return;
            u32x4 h, m, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) { unsigned a_, b_, c_; split_pair<NP, ABL>(r[0][e], r[1][e], a_, b_, c_, sc); h[e] = a_; m[e] = b_; l[e] = c_; }
            if (!legacy64) {
                unsigned char* dst = base + (tid >> 3) * 4 * 48 + (tid & 7) * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    *reinterpret_cast<unsigned*>(dst + e * 48) = h[e];
                    if (NP >= 2) *reinterpret_cast<unsigned*>(dst + e * 48 + plane) = m[e];
                    if (NP == 3) *reinterpret_cast<unsigned*>(dst + e * 48 + 2 * plane) = l[e];
                }
            } else {
                unsigned char* dst = base + (tid >> 4) * 256 + (tid & 15) * 16;
                *reinterpret_cast<u32x4*>(dst) = h;
                if (NP >= 2) *reinterpret_cast<u32x4*>(dst + plane) = m;
                if (NP == 3) *reinterpret_cast<u32x4*>(dst + 2 * plane) = l;
            }
        }
    };
    auto gmax8 = [](const f32x4 (&r)[2]) -> float {
        return fmaxf(fmaxf(fmaxf(fabsf(r[0][0]), fabsf(r[0][1])), fmaxf(fabsf(r[0][2]), fabsf(r[0][3]))),
                     fmaxf(fmaxf(fabsf(r[1][0]), fabsf(r[1][1])), fmaxf(fabsf(r[1][2]), fabsf(r[1][3]))));
    };
    auto track = [&](unsigned& ec, const f32x4 (&r)[2]) {
        const float t = gmax8(r);
        const unsigned e = __float_as_uint(t) >> 23;          // 0 for a zero (or denormal) group: exempt
        ec += e + (e != 0u ? (1u << 20) : 0u);
    };
    auto sstore = [&](int stage, const f32x4 (&xa)[2], const BRegs& xb, bool live) {
        if (ABL & 32) { asm volatile("" :: "v"(xa[0][0]), "v"(xb.f[0][0]), "v"(xa[1][3]), "v"(xb.f[1][3])); return; }     // tuning: no LDS plane writes
        unsigned char* sa = smem_raw + stage * STAGE;
        if (DETECT) {
            if (PAIRED) {
                const float ga = gmax8(xa), gb = gmax8(xb.f);
                const float w = ga * gb;
                trk_w += w;
                trk_m = fmaxf(trk_m, fmaxf(ga * sc_a, gb * sc_b));
                trk_r += w > 0.f ? fmaxf(fmaxf(w, thr_a * gb), thr_b * ga) : 0.f;
            } else {
                track(ec_a, xa);
                if (!BPL) track(ec_b, xb.f);
            }
        }
        if (A4) {          // rows (tid >> 2) and 64 + (tid >> 2), k = 4 (tid & 3) .. + 3: one 8-byte write per row and plane
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                u32x2 h, m;
                unsigned a_, b_, c_;
                split_pair<NP, ABL>(xa[q][0], xa[q][1], a_, b_, c_, sc_a); h[0] = a_; m[0] = b_;
                split_pair<NP, ABL>(xa[q][2], xa[q][3], a_, b_, c_, sc_a); h[1] = a_; m[1] = b_;
                unsigned char* dst = sa + (64 * q + (tid >> 2)) * 48 + (tid & 3) * 8;
                *reinterpret_cast<u32x2*>(dst) = h;
                *reinterpret_cast<u32x2*>(dst + PA) = m;
            }
        } else
        stage_store(sa, xa, AK, A_FULL, PA, 128, BM == 64, sc_a);
        if (BPL) {
            unsigned char* sb = sa + NPL * PA;
            if (B4) {
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    *reinterpret_cast<u32x4*>(sb + ((tid & 3) >> 1) * PB + (64 * q + (tid >> 2)) * 48 + (tid & 1) * 16) = xb.pl[q];
            } else if (B_FULL) {
                unsigned char* dst = sb + (tid >> 1) * 48 + (tid & 1) * 16;
                *reinterpret_cast<u32x4*>(dst) = xb.pl[0];
                *reinterpret_cast<u32x4*>(dst + PB) = xb.pl[1];
            } else {
                unsigned char* dst = sb + (tid >> 2) * 48 + (tid & 3) * 8;
                *reinterpret_cast<u32x2*>(dst) = u32x2{xb.pl[0][0], xb.pl[0][1]};
                *reinterpret_cast<u32x2*>(dst + PB) = u32x2{xb.pl[1][0], xb.pl[1][1]};
            }
        } else
        stage_store(sa + NPL * PA, xb.f, BKM, B_FULL, PB, 256, false, sc_b);
        if (!AK) { const float w = (do_colsum && live && (A_FULL || tid < 128)) ? 1.f : 0.f; csum += (xa[0] + xa[1]) * w; }   // branch-free: keeps the k-loop one block
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto frag = [&](const unsigned char* plane, int row, bool kmajor, int rowbytes) -> bf16x8 {
        if (ABL & 8) { u32x4 v = {(unsigned)row, (unsigned)lh, 1u, 2u}; asm volatile("" : "+v"(v)); return __builtin_bit_cast(bf16x8, v); }
        if (kmajor) return *reinterpret_cast<const bf16x8*>(plane + row * 48 + lh * 16);
        u32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const unsigned*>(plane + (lh * 4 + j) * rowbytes + row * 4);
        return __builtin_bit_cast(bf16x8, v);
    };

    // one k-tile: MFMAs on `stage` while the NEXT tile (registers xa/xb) is split into the other stage and the tile
    // after that is fetched into (ya/yb).  Branch-free, so the scheduler can interleave the three streams.
    auto compute = [&](int stage) {
        const unsigned char* sa = smem_raw + stage * STAGE;
        const unsigned char* sb = sa + NPL * PA;
        bf16x8 a[TM][3], b[TN][3];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) a[i][pl] = frag(sa + pl * PA, wm * WM + i * 32 + li, AK || BM != 64, ABYTES);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) b[j][pl] = frag(sb + pl * PB, wn * WN + j * 32 + li, true, 512);
        if (NP == 1) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = mfma<ABL>(a[i][0], b[j][0], acc[i][j]);
        } else if (NP == 2 || NP >= 4) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = mfma<ABL, F16>(a[i][1], b[j][0], acc[i][j]);
                    acc[i][j] = mfma<ABL, F16>(a[i][0], b[j][1], acc[i][j]);
                    acc[i][j] = mfma<ABL, F16>(a[i][0], b[j][0], acc[i][j]);
                }
        } else {
        // small terms first
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                acc[i][j] = mfma<ABL>(a[i][2], b[j][0], acc[i][j]);
                acc[i][j] = mfma<ABL>(a[i][0], b[j][2], acc[i][j]);
            }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                acc[i][j] = mfma<ABL>(a[i][1], b[j][1], acc[i][j]);
                acc[i][j] = mfma<ABL>(a[i][1], b[j][0], acc[i][j]);
            }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                acc[i][j] = mfma<ABL>(a[i][0], b[j][1], acc[i][j]);
                acc[i][j] = mfma<ABL>(a[i][0], b[j][0], acc[i][j]);
            }
        }
    };
    auto step = [&](int kt, int stage, f32x4 (&xa)[2], BRegs& xb, f32x4 (&ya)[2], BRegs& yb) {
        gload(kt + 2, ya, yb);
        compute(stage);
        sstore(stage ^ 1, xa, xb, kt + 1 < kt_end);
        // interleave: one MFMA, a few VALU ops of the split, now and then one of its LDS writes
        __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);       // the prefetch loads go first: a whole k-tile to land
#pragma unroll
        for (int g = 0; g < TM * TN * (NP == 3 ? 6 : (NP == 2 || NP == 4) ? 3 : 1); ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
        if (!(ABL & 64)) __syncthreads();                  // tuning: 64 = no barrier per k-tile (results garbage)
    };

    gload(kt_begin, ra[0], rb[0]);
    gload(kt_begin + 1, ra[1], rb[1]);
    if (F16) { amax_scale(p.a_amax, sc_a, so_a, p.A2 != nullptr ? p.a2_amax : nullptr); amax_scale(p.b_amax, sc_b, so_b); }
    thr_a = 0.125f * so_a; thr_b = 0.125f * so_b;
    sstore(0, ra[0], rb[0], kt_begin < kt_end);
    __syncthreads();
    if (FLUSH > 0) {
        f32x16 tot[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) tot[i][j][r] = 0.f;
        int since = 0;
        for (int kt = kt_begin; kt < kt_end; kt += 2) {
            step(kt, 0, ra[1], rb[1], ra[0], rb[0]);
            if (kt + 1 < kt_end) step(kt + 1, 1, ra[0], rb[0], ra[1], rb[1]);
            since += 2;
            if (since >= FLUSH) {          // wave-uniform
                since = 0;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) { tot[i][j][r] += acc[i][j][r]; acc[i][j][r] = 0.f; }
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] += tot[i][j][r];
    } else {
    for (int kt = kt_begin; kt < kt_end; kt += 2) {
        step(kt, 0, ra[1], rb[1], ra[0], rb[0]);
        if (kt + 1 < kt_end) step(kt + 1, 1, ra[0], rb[0], ra[1], rb[1]);
    }
    }

    if (DETECT) {
        // L = log2(scaled tensor maximum, taken as 2^13.5) - mean scaled exponent of the thread's group maxima, in 1/16 steps (>= 0 for a
        // word that bounds the data); the workgroup's largest per operand through LDS (the stages are free now)
        auto spread = [&](unsigned ec, float sc) -> int {
            const unsigned es = ec & 0xfffffu, cn = ec >> 20;
            if (cn == 0u) return 0;
            const float mean_e = (float)es / (float)cn - 127.f + (float)((int)(__float_as_uint(sc) >> 23) - 127);
            return (int)((13.5f - mean_e) * 16.f);
        };
        const int la = spread(ec_a, sc_a);
        int bad, stale;
        if (BPL) {          // the weight's L is uniform: every thread judges its own rows
            const int lb_w = (int)(fminf(fmaxf(__uint_as_float(reinterpret_cast<const unsigned*>(p.b_amax)[p.bpl_flag]), 0.f), 64.f) * 16.f);      // (clamped: a NaN or a garbage verdict is a number of binades, never an int overflow)
            bad = la + lb_w > 17 * 16 ? 1 : 0;
            stale = la < -24 ? 1 : 0;          // a mean 1.5 binades above the claimed maximum: a stale word
        } else if (PAIRED) {
            bad = trk_r > 4.f * trk_w ? 1 : 0;
            stale = !(trk_w < 3.0e38f && trk_m < 32768.f) ? 1 : 0;          // a word 2x too small or more (fresh ones scale the maximum
                                                                            // into [2^13, 2^14)): the first piece would leave fp16; inf / NaN operands
        } else {            // both operands tracked here, laid out differently: each thread votes on its own pair of streams
            const int lb = spread(ec_b, sc_b);
            bad = la + lb > 17 * 16 ? 1 : 0;
            stale = (la < -24 || lb < -24) ? 1 : 0;
        }
        if (kt_end - kt_begin >= 4096) stale = 1;          // (a k range too long for the packed counters: play safe)
        // A VOTE, not a veto: the tile is redone when an eighth of its threads (16 of 128 rows) see the spread.  The patterns that matter
        // -- a massive token or channel, a quarter of the rows tiny -- raise (nearly) every thread; a few tiny rows among ordinary ones
        // (tokens of an almost empty latent patch: thousands per step on the skewed NACA meshes) do not: those rows keep the absolute
        // floor, 2^-39 of the tensor's largest magnitude, which only their own (tiny) outputs see.
        // (one barrier: every wave leaves its count and its stale flag in LDS; __syncthreads_count + __syncthreads_or are six)
        __shared__ int s_vote[8];
        {
            const int nb = __popcll(__ballot(bad != 0)), ns = __ballot(stale != 0) != 0ull ? 1 : 0;
            if (lane == 0) s_vote[wave] = nb | (ns << 16);
        }
        __syncthreads();
        int vsum = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) vsum += s_vote[w];
        const int votes = vsum & 0xffff;
        if (__builtin_expect((vsum >> 16) != 0 || votes * 8 >= NT, 0)) {
            // ---- the second pass: the operands as they are, fp32, through ONE LDS stage ([row][16 k], row stride 20 floats) into the
            // fp32 MFMA (v_mfma_f32_32x32x2_f32: the arithmetic of the fp32-MFMA tiles, gemm.hip) -- no pieces, no scales, nothing that
            // depends on the operands' range.  A plain loop (load -> store -> barrier -> MFMAs -> barrier); it shares the accumulators,
            // the column sums (already complete: they come from the raw values) and the epilogue with the first pass.  1/16 of the f16
            // pipe's rate, and rare.
            if (tid == 0) atomicAdd(&g_split_redo_tiles, 1u);
            constexpr int LDS_S = 20;
            static_assert((BM + BN) * LDS_S * 4 <= G::SMEM_BYTES, "second pass: one fp32 stage");
            float* As = reinterpret_cast<float*>(smem_raw);
            float* Bs = As + BM * LDS_S;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            // B as it lies in memory (with planes, the first pass never read it): k-contiguous rows (b_src has them, SwiGLU band map
            // included) or row-contiguous (k pair = tid & 7, row quad = tid >> 3)
            const float* bf0 = nullptr; const float* bf1 = nullptr;
            if (BREAL) { bf0 = b_src[0]; bf1 = b_src[1]; }
            else {
                bf0 = p.B + (long)(2 * (tid & 7)) * p.ldb + min(n0 + ((tid >> 3) & 31) * 4, p.N - 4);
                bf1 = bf0 + p.ldb;
            }
            for (int kt = kt_begin; kt < kt_end; ++kt) {
                const long k0 = (long)kt * SBK;
                f32x4 xa0 = {0.f, 0.f, 0.f, 0.f}, xa1 = xa0, xb0 = xa0, xb1 = xa0;
                if (AK) {
                    const bool sec = p.A2 != nullptr && k0 >= p.k_split;
                    const long d = sec ? a2_delta : 0L, d1 = sec ? (A4 ? a2_delta1 : a2_delta) : 0L;
                    xa0 = *reinterpret_cast<const f32x4*>(a_src[0] + k0 + d);
                    if (A_FULL) xa1 = *reinterpret_cast<const f32x4*>(a_src[1] + k0 + d1);
                } else if (A_FULL || tid < 128) {
                    xa0 = *reinterpret_cast<const f32x4*>(a_src[0] + k0 * p.lda);
                    xa1 = *reinterpret_cast<const f32x4*>(a_src[1] + k0 * p.lda);
                }
                if (BREAL) {
                    xb0 = *reinterpret_cast<const f32x4*>(bf0 + k0);
                    if (B_FULL) xb1 = *reinterpret_cast<const f32x4*>(bf1 + k0);
                } else if (tid < 256) {
                    xb0 = *reinterpret_cast<const f32x4*>(bf0 + k0 * p.ldb);
                    xb1 = *reinterpret_cast<const f32x4*>(bf1 + k0 * p.ldb);
                }
                if (AK) {
                    if (A4) {
                        *reinterpret_cast<f32x4*>(As + (tid >> 2) * LDS_S + (tid & 3) * 4) = xa0;
                        *reinterpret_cast<f32x4*>(As + (64 + (tid >> 2)) * LDS_S + (tid & 3) * 4) = xa1;
                    } else if (A_FULL) {
                        *reinterpret_cast<f32x4*>(As + (tid >> 1) * LDS_S + (tid & 1) * 8) = xa0;
                        *reinterpret_cast<f32x4*>(As + (tid >> 1) * LDS_S + (tid & 1) * 8 + 4) = xa1;
                    } else *reinterpret_cast<f32x4*>(As + (tid >> 2) * LDS_S + (tid & 3) * 4) = xa0;
                } else if (A_FULL || tid < 128) {
                    const int kp = A_FULL ? (tid & 7) : ((tid >> 4) & 7), r0 = A_FULL ? (tid >> 3) * 4 : (tid & 15) * 4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) *reinterpret_cast<f32x2*>(As + (r0 + e) * LDS_S + 2 * kp) = f32x2{xa0[e], xa1[e]};
                }
                if (BREAL) {
                    if (B_FULL) {
                        *reinterpret_cast<f32x4*>(Bs + (tid >> 1) * LDS_S + (tid & 1) * 8) = xb0;
                        *reinterpret_cast<f32x4*>(Bs + (tid >> 1) * LDS_S + (tid & 1) * 8 + 4) = xb1;
                    } else *reinterpret_cast<f32x4*>(Bs + (tid >> 2) * LDS_S + (tid & 3) * 4) = xb0;
                } else if (tid < 256) {
                    const int kp = tid & 7, r0 = ((tid >> 3) & 31) * 4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) *reinterpret_cast<f32x2*>(Bs + (r0 + e) * LDS_S + 2 * kp) = f32x2{xb0[e], xb1[e]};
                }
                __syncthreads();
#pragma unroll
                for (int g8 = 0; g8 < 2; ++g8) {
                    f32x4 a[TM], b[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f32x4*>(As + (wm * WM + i * 32 + li) * LDS_S + 8 * g8 + 4 * lh);
#pragma unroll
                    for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const f32x4*>(Bs + (wn * WN + j * 32 + li) * LDS_S + 8 * g8 + 4 * lh);
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][q], b[j][q], acc[i][j], 0, 0, 0);
                }
                __syncthreads();
            }
            so_a = 1.f; so_b = 1.f;
        }
    }

    if (!AK && do_colsum) {          // reduce the 8 k-pair groups through LDS (the stages are free now)
        float* cs = reinterpret_cast<float*>(smem_raw);
        if (A_FULL) *reinterpret_cast<f32x4*>(cs + (tid & 7) * BM + (tid >> 3) * 4) = csum;
        else if (tid < 128) *reinterpret_cast<f32x4*>(cs + (tid >> 4) * BM + (tid & 15) * 4) = csum;
        __syncthreads();
        if (tid < BM && m0 + tid < p.M) {
            float s = 0.f;
#pragma unroll
            for (int g = 0; g < 8; ++g) s += cs[g * BM + tid];
            if (p.split_k > 1) p.ws[(long)p.split_k * p.M * p.N + (long)zs * p.M + m0 + tid] = s;
            else p.colsum[m0 + tid] = s;
        }
        __syncthreads();
    }
    if (ABL & 4) { if (acc[0][0][0] + acc[1][1][3] + acc[0][1][5] + acc[1][0][7] == 123.456f) p.C[tid] = 1.f; __syncthreads(); return; }
    if (BKM && p.act == GAOT_ACT_SWIGLU) epilogue_swiglu<TM, TN, WM, WN>(p, reinterpret_cast<float*>(smem_raw), acc, m0, n0, wm, wn, wave, lane, so_a, so_b);
    else epilogue_vec<TM, TN, WM, WN>(p, reinterpret_cast<float*>(smem_raw), acc, m0, n0, wm, wn, wave, lane, zs, so_a, so_b);
    __syncthreads();          // the epilogue's LDS slabs alias the stages the next tile is about to fill
}

template <bool AK, bool BKM, int BM = 128, int ABL = 0, int NP = 3, bool BPL = false, bool BREAL = BKM>
__global__ __launch_bounds__(BM == 256 ? 512 : 256, BM == 256 ? 1 : ((NP == 2 || NP >= 4) ? GAOT_SPLIT2_WG_PER_CU : 2)) void gemm_split_kernel(const GemmArgs p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem_raw[SplitGeom<BM, NP>::SMEM_BYTES];
    if ((ABL & 128) && (blockIdx.x & 8)) {      // tuning: de-phase half of the workgroups by ~p.ablate x 3.4 us at start
        for (int i = 0; i < p.ablate; ++i) __builtin_amdgcn_s_sleep(127);
    }
    const int tiles = p.tiles_m * p.tiles_n;
    {
        const int vb = blockIdx.x;
        // XCD-aware tile order (as gemm.hip)
        const int q = tiles >> 3, r = tiles & 7, x = vb & 7, slot = vb >> 3;
        const int logical = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + slot;
        split_tile<AK, BKM, BM, ABL, NP, 0, BPL, BREAL>(p, smem_raw, logical, blockIdx.z);
    }
}

// ---- grouped weight-gradient products: ONE launch over many C_i[M_i,N_i] = A_i[K_i,M_i]^T B_i[K_i,N_i] (both operands row-contiguous,
// the long reduction runs over the rows: dW = dY^T X of every Linear of a backward pass).  A workgroup finds its product in a
// prefix table carried in the kernel arguments, computes one 128x128 tile of one K slab with split_tile<TN>, and -- when the
// product is cut into K slabs (at most 1 024 values of k per workgroup: the bf16 MFMA accumulation cap, gemm.hip) -- the LAST
// workgroup to finish a tile sums that tile's slabs in slab order and writes C (and the fused column sums): deterministic, no
// atomics on data, no separate reduce launch.  Cross-workgroup visibility follows cdna_hip_programming.md guideline 16: plain
// slab stores -> every wave drains vmcnt -> barrier -> lane 0: agent-scope release, asm vmcnt(0), relaxed agent-scope ticket ->
// the last arriver: agent-scope acquire (invalidates this CU's L1) -> barrier -> plain loads.  The ticket counters return to
// zero (the last arriver resets its counter), so the same zero-initialised buffer serves every launch.
constexpr int TNG_MAX = TN_GROUP_MAX;
struct TnProb {
    const float* A; const float* B; float* C; float* colsum;
    int M, N, K; int lda, ldb, ldc;
    int tiles_m, tiles_n, split, kt_per_split;     // kt_per_split in 32-wide k-tiles (GemmArgs convention)
    int wg_end;                                    // first workgroup index past this product
    int cnt_off;                                   // first ticket counter of this product
    long ws_off;                                   // float offset of this product's slabs in the workspace
    const float* a_amax; const float* b_amax;      // fp16 pieces (NP = 4): the operands' magnitude words
};
struct TnGroupArgs { int n; float* ws; int* counters; TnProb p[TNG_MAX]; };

// ABL: tuning builds only (the ablation bits of split_tile).  Measured on the 15 products of the bench step (K = 8 192, K slabs of
// 4 096, 364 us): without the split arithmetic 345, without MFMAs 269, without LDS fragment reads 297, without in-loop global loads
// 304, without LDS plane writes 282, without the barrier 351 -- every part costs 20-100 us and the parts ADD UP (the k-loop's
// load -> split -> plane write -> barrier -> fragment read -> MFMA chain is not overlapped across the two resident workgroups of a CU).
// BM = 256 [r6]: 256 x 128 tiles on eight waves, ONE workgroup per CU (the same eight waves and register budget as two 128 x 128
// workgroups): 384 operand rows per k-tile and CU instead of 512 for the same products
template <int ABL = 0, int NP = 3, int BM = 128>
__global__ __launch_bounds__(BM == 256 ? 512 : 256, BM == 256 ? 1 : 2) void gemm_tn_grouped_kernel(const TnGroupArgs g) {      // (three per CU: 168 registers, 94 spilled)
    __shared__ __attribute__((aligned(16))) unsigned char smem_raw[SplitGeom<BM, NP>::SMEM_BYTES];
    // XCD-aware order (workgroup b runs on XCD b % 8): every XCD takes a CONTIGUOUS range of the logical work list, and inside
    // a product the list runs K slab by K slab, tile row by tile row -- so the workgroups an XCD's L2 serves at the same time
    // share the A panel of one (tile row, K slab) and walk the B panels of neighbouring tile columns
    int b;
    {
        const int total = gridDim.x, q = total >> 3, r = total & 7, x = blockIdx.x & 7, slot = blockIdx.x >> 3;
        b = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + slot;
    }
    int i = 0;
    while (i + 1 < g.n && b >= g.p[i].wg_end) ++i;                  // wave-uniform scan of <= 24 entries
    const int first = i > 0 ? g.p[i - 1].wg_end : 0;
    const int split = g.p[i].split;
    const int local = b - first;
    const int ntile = g.p[i].tiles_m * g.p[i].tiles_n;
    const int z = local / ntile, tile = local - z * ntile;
    GemmArgs a;
    a.M = g.p[i].M; a.N = g.p[i].N; a.K = g.p[i].K;
    a.A = g.p[i].A; a.lda = g.p[i].lda; a.A2 = nullptr; a.lda2 = 0; a.k_split = 0;
    a.B = g.p[i].B; a.ldb = g.p[i].ldb; a.C = g.p[i].C; a.ldc = g.p[i].ldc;
    a.bias = nullptr; a.rowbias = nullptr; a.rb_period = 0; a.ld_rb = 0; a.rowscale = nullptr; a.act = GAOT_ACT_NONE;
    a.aux_in = nullptr; a.aux_out = nullptr; a.ld_aux = 0; a.residual = nullptr; a.ldr = 0;
    a.split_k = split; a.ktiles_per_split = g.p[i].kt_per_split; a.ws = g.ws + g.p[i].ws_off;
    a.colsum = g.p[i].colsum; a.tiles_m = g.p[i].tiles_m; a.tiles_n = g.p[i].tiles_n; a.vec_epi = 1; a.ablate = 0;
    a.a_amax = g.p[i].a_amax; a.b_amax = g.p[i].b_amax; a.c_amax = nullptr; a.a2_amax = nullptr;
    a.Bpl = nullptr; a.ld_bpl = 0; a.bpl_stride = 0;
    a.bpl_flag = 0;
    split_tile<false, false, BM, ABL, NP, 64, false>(a, smem_raw, tile, z);
    if (split <= 1) return;

    const int tid = threadIdx.x;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // this wave's slab stores have left
    __syncthreads();
    int* flag = reinterpret_cast<int*>(smem_raw);
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the write-back must not be overtaken by the ticket (ROCm 7.2 drops the fence's own wait)
        int* cnt = g.counters + g.p[i].cnt_off + tile;
        const int t = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = t == split - 1;
        if (last) {
            __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        *flag = last;
    }
    __syncthreads();
    if (!*flag) return;
    const int m0 = (tile / a.tiles_n) * BM, n0 = (tile % a.tiles_n) * 128;
    const long slab = (long)a.M * a.N;
#pragma unroll 4
    for (int idx = tid; idx < BM * 32; idx += (BM == 256 ? 512 : 256)) {
        const int m = m0 + (idx >> 5), n = n0 + (idx & 31) * 4;
        if (m < a.M && n < a.N) {
            const float* src = a.ws + (long)m * a.N + n;
            f32x4 s = *reinterpret_cast<const f32x4*>(src);
            for (int zz = 1; zz < split; ++zz) s += *reinterpret_cast<const f32x4*>(src + zz * slab);
            *reinterpret_cast<f32x4*>(a.C + (long)m * a.ldc + n) = s;
        }
    }
    if (a.colsum != nullptr && (tile % a.tiles_n) == 0 && tid < BM && m0 + tid < a.M) {
        const float* src = a.ws + (long)split * slab + m0 + tid;
        float s = src[0];
        for (int zz = 1; zz < split; ++zz) s += src[(long)zz * a.M];
        a.colsum[m0 + tid] = s;
    }
}

static int g_tn_bm = 128;            // tile rows of the grouped launch: 128, or 256 when every product's M is a multiple of 256 (tuning hook)
void set_tn_bm(int bm) { g_tn_bm = bm == 256 ? 256 : 128; }
static int tn_tile_rows(const gaot_wgrad_item* items, int n) {
    if (g_tn_bm != 256) return 128;
    for (int i = 0; i < n; ++i) if (items[i].M % 256 != 0) return 128;
    return 256;
}
static int g_tn_rule = 0;            // automatic K slabs: 0 = the rule of rounds 3-5 (default), 1 = the cost model below [r6: measured, not kept] (tuning hook)
void set_tn_rule(int r) { g_tn_rule = r; }
static int g_tn_kslab = 0;           // 0 = automatic (below); otherwise a fixed K slab (tuning hook)
static int g_tn_cap_long = 4096;     // the same for node-level products (K = batch x nodes > 16 384)
static int g_tn_cap = 4096;          // longest K slab of a product whose reduction is much longer than the rest (tuning hook: a negative argument sets it)
void set_tn_kslab(int k) {
    if (k <= -100000) { g_tn_cap_long = ((-k - 100000) < 256 ? 256 : ((-k - 100000) / 32) * 32); g_tn_kslab = 0; return; }      // -(100000 + cap): node-level products only
    if (k < 0) { g_tn_cap = (-k < 256 ? 256 : (-k / 32) * 32); g_tn_kslab = 0; return; }
    g_tn_kslab = k == 0 ? 0 : (k < 256 ? 256 : (k / 32) * 32);
}
// host side of the grouped launch: items -> prefix table; returns the workspace floats / counters it needs when `args` is null
long plan_tn_grouped(const gaot_wgrad_item* items, int n, TnGroupArgs* args, int* n_counters, int* n_wg, int force_bm) {
    long ws = 0; int cnt = 0, wg = 0;
    // K slab per workgroup: the accumulators are flushed to the vector pipe every 1 024 values of k inside the kernel, so the slab
    // length is a load-balance / slab-traffic choice.  A launch lasts as long as one workgroup's K loop, whatever the number of work
    // items, until the items outnumber the workgroups the chip holds: so the slab count is what brings tiles x slabs to a bit more
    // than one workgroup per CU (272).  Measured (tools/wgrad_kslab_sweep.py, K = 8 192): the 14-15 products of a whole backward pass
    // (204 tiles): 314 / 340 / 387 us at slabs of 4 096 / 2 048 / 1 024 -> 2 slabs; the 4 products of ONE phase of a staged backward
    // (68 tiles): 160 / 108 / 123 us -> 4 slabs.
    const int BMr = force_bm ? force_bm : tn_tile_rows(items, n);
    long tiles_total = 0;
    for (int i = 0; i < n; ++i) tiles_total += (long)cdiv(items[i].M, BMr) * cdiv(items[i].N, 128);
    const int want = BMr == 256 ? (int)(256 / (tiles_total > 0 ? tiles_total : 1))          // one workgroup per CU: not more than one round
                                : (int)((272 + tiles_total - 1) / (tiles_total > 0 ? tiles_total : 1));
    // ONE slab length for the whole launch: the launch lasts as long as its longest K loop times the rounds, so a product with a longer
    // reduction than the rest (the patch-level layers: K = 4 x tokens; node-level ones: batch x nodes) is cut to the slab of the shortest
    // one, not to a fixed 4 096 -- 4 096-token batch, same box (tools/step_ab.py gaot_debug_set_wgrad_kslab ... --c4), slabs of
    // token-level / longer products: 2 048 / 4 096 (the former rule) 1.5777 ms, 4 096 / 4 096 1.5332; on two other boxes 4 096 / 4 096
    // 1.5421, 1.6240 against 2 048 / 2 048 1.5118, 1.5938.  K = 8 192 tokens keeps its 4 096 / 4 096 (2 048 / 2 048: 2.1076 -> 2.1505)
    long kmin = 0;
    for (int i = 0; i < n; ++i) kmin = (kmin == 0 || items[i].K < kmin) ? items[i].K : kmin;
    int base_slab = 4096;
    {
        int s0 = want < 1 ? 1 : (want > 16 ? 16 : want);
        if (s0 > kmin / 512) s0 = kmin / 512 > 0 ? (int)(kmin / 512) : 1;
        const long b = (kmin + s0 - 1) / s0;
        base_slab = b < 1024 ? 1024 : (b > 4096 ? 4096 : (int)b);
    }
    // [r6, measured and NOT kept: g_tn_rule = 1 only] same-box steps, rule -> model: C2 2.1152 -> 2.1242 ms, C3 4.763 -> 4.804, C4 / C5
    // within noise -- a round that is 80 % full is not 20 % wasted (the CUs that hold one workgroup instead of two run it faster), so
    // slab counts that land just under a whole number of rounds buy nothing and pay their extra slabs.
    // the slab length by a cost model instead of the rule above: a launch lasts  rounds x (longest K loop) + a x slabs,
    // rounds = ceil(workgroups / resident slots) -- 512 slots at two 128-row workgroups per CU -- and a ~ 185 values of k per slab (the slab
    // stores and the last arriver's sum; from the sweep above: 314 / 340 us at 2 / 4 slabs of the 204-tile launch).  The model reproduces
    // that sweep's 387 us at 8 slabs and the 68-tile launch's 160 / 108 / 123 us, and finds what the rule misses: slab counts that land
    // just under a whole number of rounds (204 tiles x 5 slabs = 1 020 workgroups = two full rounds of 1 664 instead of one 80 %-full
    // round of 4 096; K = 16 384: 5 x 3 296 instead of 4 x 4 096).
    int model_slab = 0;
    if (g_tn_rule == 1 && g_tn_kslab == 0 && kmin >= 1024) {
        const long slots = BMr == 256 ? 256 : 512;
        double best = 1e30;
        for (int L = 512; L <= 4096; L += 32) {
            long total = 0, lmax = 0; int smin_k = 1;
            for (int i = 0; i < n; ++i) {
                const int kt32 = items[i].K / 32;
                int sp = (int)((items[i].K + L - 1) / L);
                if (sp > 16 && items[i].K > 4 * kmin) sp = 16;          // (node-level reductions: many slabs would queue on one tile's last arriver)
                const int per = (kt32 + sp - 1) / sp;
                sp = (kt32 + per - 1) / per;
                total += (long)cdiv(items[i].M, BMr) * cdiv(items[i].N, 128) * sp;
                lmax = per * 32L > lmax ? per * 32L : lmax;
                if (items[i].K == kmin) smin_k = sp;
            }
            const long rounds = (total + slots - 1) / slots;
            const double cost = (double)rounds * lmax + 185.0 * smin_k;
            if (cost < best - 1e-9 || (cost < best + 1e-9 && L > model_slab)) { best = cost; model_slab = L; }
        }
    }
    for (int i = 0; i < n; ++i) {
        const gaot_wgrad_item& it = items[i];
        const int kt32 = it.K / 32;
        int kslab = g_tn_kslab;
        if (kslab == 0 && model_slab > 0) {
            kslab = model_slab;
            if ((it.K + kslab - 1) / kslab > 16 && it.K > 4 * kmin) kslab = (int)((it.K + 15) / 16);
        } else
        if (kslab == 0) {
            int s_auto = want < 1 ? 1 : (want > 16 ? 16 : want);      // the last workgroup of a tile sums the slabs alone: keep them few
            if (s_auto > it.K / 512) s_auto = it.K / 512 > 0 ? it.K / 512 : 1;      // slabs of at least 512
            kslab = (it.K + s_auto - 1) / s_auto;
            const int cap0 = it.K > 16384 ? g_tn_cap_long : g_tn_cap;
            const int cap = cap0 < base_slab ? cap0 : base_slab;
            if (kslab > cap) kslab = cap;               // products with a much longer reduction than the rest (node-level layers: K = batch x nodes)
        }
        int split = (it.K + kslab - 1) / kslab;
        int per = (kt32 + split - 1) / split;
        split = (kt32 + per - 1) / per;
        const int tm = cdiv(it.M, BMr), tn = cdiv(it.N, 128);
        if (args) {
            TnProb& q = args->p[i];
            q.A = it.g; q.B = it.x; q.C = it.out; q.colsum = it.colsum;
            q.M = it.M; q.N = it.N; q.K = it.K; q.lda = (int)it.ldg; q.ldb = (int)it.ldx; q.ldc = (int)it.ldo;
            q.tiles_m = tm; q.tiles_n = tn; q.split = split; q.kt_per_split = per;
            q.wg_end = wg + tm * tn * split; q.cnt_off = cnt; q.ws_off = ws;
            q.a_amax = it.g_absmax; q.b_amax = it.x_absmax;
        }
        wg += tm * tn * split;
        if (split > 1) { ws += (long)split * ((long)it.M * it.N + it.M); ws = (ws + 3) & ~3L; cnt += tm * tn; }
    }
    if (n_counters) *n_counters = cnt;
    if (n_wg) *n_wg = wg;
    return ws;
}

// pieces per operand: 3 = exact (six piece products), 2 = two rounded pieces (three piece products)
void launch_tn_grouped(const gaot_wgrad_item* items, int n, float* ws, int* counters, int pieces, hipStream_t st) {
    TnGroupArgs args;
    args.n = n; args.ws = ws; args.counters = counters;
    int wg = 0;
    const int bm = pieces == 4 ? tn_tile_rows(items, n) : 128;          // (the 256-row tiles exist for the fp16 pieces only)
    plan_tn_grouped(items, n, &args, nullptr, &wg, bm);
    if (bm == 256) hipLaunchKernelGGL((gemm_tn_grouped_kernel<0, 4, 256>), dim3(wg), dim3(512), 0, st, args);
    else if (pieces == 4) hipLaunchKernelGGL((gemm_tn_grouped_kernel<0, 4>), dim3(wg), dim3(256), 0, st, args);
    else if (pieces == 2) hipLaunchKernelGGL((gemm_tn_grouped_kernel<0, 2>), dim3(wg), dim3(256), 0, st, args);
    else hipLaunchKernelGGL((gemm_tn_grouped_kernel<0, 3>), dim3(wg), dim3(256), 0, st, args);
}

template <int BM, int NP>
static void launch_split_bm(GemmArgs& a, bool ak, bool bk, hipStream_t st) {
    a.tiles_m = cdiv(a.M, BM);
    a.tiles_n = cdiv(a.N, S_BN);
    const int gx = a.tiles_m * a.tiles_n;
    const int z = a.split_k > 1 ? a.split_k : 1;
    dim3 grid(gx, 1, z);
    dim3 block(BM == 256 ? 512 : 256);
    a.bpl_flag = bk ? 1 : 2;                    // the weight word's verdict on groups along the rows (NT) / columns (NN) of the matrix as stored
    if (NP == 4 && a.Bpl != nullptr) {          // pre-split B (weights): the planes are k-contiguous whatever B's own layout
        if (ak && bk)       hipLaunchKernelGGL((gemm_split_kernel<true, true, BM, 0, 4, true, true>), grid, block, 0, st, a);
        else if (ak)        hipLaunchKernelGGL((gemm_split_kernel<true, true, BM, 0, 4, true, false>), grid, block, 0, st, a);
        else if (bk)        hipLaunchKernelGGL((gemm_split_kernel<false, true, BM, 0, 4, true, true>), grid, block, 0, st, a);
        else                hipLaunchKernelGGL((gemm_split_kernel<false, true, BM, 0, 4, true, false>), grid, block, 0, st, a);
    }
    else if (ak && bk)   hipLaunchKernelGGL((gemm_split_kernel<true, true, BM, 0, NP>), grid, block, 0, st, a);
    else if (ak && !bk)  hipLaunchKernelGGL((gemm_split_kernel<true, false, BM, 0, NP>), grid, block, 0, st, a);
    else if (!ak && !bk) hipLaunchKernelGGL((gemm_split_kernel<false, false, BM, 0, NP>), grid, block, 0, st, a);
    else                 hipLaunchKernelGGL((gemm_split_kernel<false, true, BM, 0, NP>), grid, block, 0, st, a);
}

unsigned split_redo_count(bool reset) {
    unsigned v = 0;
    hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_split_redo_tiles), sizeof(v));
    if (reset) { const unsigned z = 0; hipMemcpyToSymbol(HIP_SYMBOL(g_split_redo_tiles), &z, sizeof(z)); }
    return v;
}

void launch_split(GemmArgs& a, bool ak, bool bk, hipStream_t st, int bm, int pieces) {
    if (pieces == 1) {           // plain bf16 operands (bench variant): 128- and 64-row tiles
        if (bm == 64) launch_split_bm<64, 1>(a, ak, bk, st); else launch_split_bm<128, 1>(a, ak, bk, st);
        return;
    }
    if (pieces == 4) {           // two fp16 pieces per scaled operand (three piece products on the f16 MFMA)
        if (bm == 64) launch_split_bm<64, 4>(a, ak, bk, st); else if (bm == 256) launch_split_bm<256, 4>(a, ak, bk, st); else launch_split_bm<128, 4>(a, ak, bk, st);
        return;
    }
    if (pieces == 2) {           // two rounded pieces per operand (three piece products)
        if (bm == 64) launch_split_bm<64, 2>(a, ak, bk, st);
        else if (bm == 256) launch_split_bm<256, 2>(a, ak, bk, st);
        else launch_split_bm<128, 2>(a, ak, bk, st);
        return;
    }
    if (bm == 64)       launch_split_bm<64, 3>(a, ak, bk, st);
    else if (bm == 256) launch_split_bm<256, 3>(a, ak, bk, st);
    else                launch_split_bm<128, 3>(a, ak, bk, st);
}

}  // namespace gaot
