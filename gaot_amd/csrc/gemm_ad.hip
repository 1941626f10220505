// fp32-level GEMM on two fp16 pieces, "all DMA" form: C[M,N] = A[M,K] W^T with A fp32, k-contiguous, and W PRE-SPLIT into the two
// fp16 planes of the scaled weight (gaot_split_f16_planes_grouped) -- the shape of every forward product x W^T and every
// input-gradient product dY W of the processor.  Same arithmetic as gemm_split.hip's tiles (bit-identical: the same piece products in
// the same order, the same epilogues); what differs is how the operands reach the matrix pipe:
//   * BOTH operands go global -> LDS with global_load_lds_dwordx4: no staging registers, no ds_write, and -- the point -- every wave
//     instruction fetches 8 rows x 128 contiguous bytes.  The staged tiles load A with two lanes per row (64 separate 16-byte pieces
//     per wave instruction: the texture addresser takes them one at a time, and at K = 256 that address path, not the matrix pipe or
//     the LDS, is what a workgroup waits for: tools/ad_bench.hip, ablations);
//   * A sits in LDS as fp32, [row][32 k], the 16-byte chunks of a row XOR-swizzled on the SOURCE side (gemm_glds.hip's scheme:
//     conflict-free ds_read_b128); a wave reads its own rows (lane = row, 8 consecutive k = two 16-byte reads), splits them in
//     registers and feeds the MFMA -- the split arithmetic runs beside the MFMAs of the previous k-step;
//   * k-tile 32 (two MFMA k-steps), an NS-slot ring, ONE s_barrier and ONE counted s_waitcnt vmcnt per k-tile.
// Wave layouts: 128 x 128 tiles as 4 x 1 waves of 32 x 128 outputs (every wave splits its own rows once: 32 split instructions per
// 12 MFMAs), two ring slots of 32 KB, two workgroups per CU; 64 x 128 tiles as 2 x 2 waves of 32 x 64, three slots of 24 KB.
// The fragment reads and the waits are inline asm ON PURPOSE: the compiler cannot tell which ring slot a global_load_lds is filling
// and puts `s_waitcnt vmcnt(0)` in front of every LDS read it knows about while one is in flight.  The price: it does not know these
// reads are pending either, so every consumer waits explicitly with the registers pinned behind the wait.
// The range watch of the fp16 pieces (gemm_split.hip) is kept: per thread the geometric mean of its groups' maxima against the
// tensor's maximum, a vote, and a second pass on the fp32 MFMA -- here straight from global memory (no staging at all: rare and slow).
#include "gemm_common.h"

namespace gaot {

constexpr int AD_BK = 32;

__device__ unsigned g_ad_redo_tiles = 0;      // tiles that took the second pass (summed into gaot_debug_split_redo_count)

__device__ __forceinline__ f32x16 ad_mfma(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

template <int TM> struct AdRaw { f32x4 v[TM][2]; };            // [i][half]: 8 consecutive k of row i (one k-step)
template <int TM> struct AdPieces { u32x4 h[TM], m[TM]; };
template <int TN> struct AdFrag { u32x4 h[TN], m[TN]; };

template <int N> __device__ __forceinline__ void ad_lgkm_wait() {
    static_assert(N >= 0 && N <= 15, "lgkmcnt is a 4-bit counter");
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}
// `asm volatile("" : "+v"(x))`: the value cannot be read (or moved) before this point
template <int TM> __device__ __forceinline__ void ad_pin(AdRaw<TM>& r) {
#pragma unroll
    for (int i = 0; i < TM; ++i) { asm volatile("" : "+v"(r.v[i][0])); asm volatile("" : "+v"(r.v[i][1])); }
}
template <int TN> __device__ __forceinline__ void ad_pin(AdFrag<TN>& f) {
#pragma unroll
    for (int j = 0; j < TN; ++j) { asm volatile("" : "+v"(f.h[j])); asm volatile("" : "+v"(f.m[j])); }
}

// BMT x BNT outputs per workgroup, 4 waves as WAVES_M x (4 / WAVES_M), NS ring slots (BNT = 64: 64 x 64 tiles, twice the workgroups for
// outputs two 128-wide tiles across that would leave every CU with one wave per SIMD).  BREAL: is W itself k-contiguous as stored (NT: the
// planes of W; NN: the planes of W^T) -- only the second pass reads it.
// FL [r6]: reductions longer than the accumulation cap of the f16 / bf16 MFMA (1 024 values of k: its accumulator does not round to nearest)
// in ONE pass: every 32 k-tiles the accumulators join running sums on the vector pipe and start from zero again -- what K slabs + a reduce
// launch do, without the slabs (the 64 x 64 tiles only: 16 more registers).
template <int BMT, int WAVES_M, int NS, bool BREAL, int ABL = 0, int BNT = 128, bool FL = false>
__global__ __launch_bounds__(256, 2) void gemm_ad_kernel(const GemmArgs p) {
    constexpr int WAVES_N = 4 / WAVES_M, WM = BMT / WAVES_M, WN = BNT / WAVES_N, TM = WM / 32, TN = WN / 32;
    constexpr int A_ST = BMT * 128, B_ST = BNT * 128, STAGE = A_ST + B_ST;          // bytes
    constexpr int LA = BMT / 32, LB = BNT / 32;                                     // 1-KB DMA pieces per wave and tile
    constexpr int EPI_BYTES = 4 * 32 * (WN + 4) * 4;
    constexpr int SMEM = NS * STAGE > EPI_BYTES ? NS * STAGE : EPI_BYTES;
    static_assert(TM >= 1 && TN >= 1 && (NS == 2 || NS == 3), "layout");
    __shared__ __attribute__((aligned(16))) unsigned char smem_raw[SMEM];
#ifndef GAOT_NO_DETECT
    constexpr bool DETECT = (ABL & 32) == 0;
#else
    constexpr bool DETECT = false;
#endif

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    const int tiles = p.tiles_m * p.tiles_n;
    int logical;
    {   // XCD-aware tile order (as gemm.hip)
        const int vb = blockIdx.x, q = tiles >> 3, r = tiles & 7, x = vb & 7, slot = vb >> 3;
        logical = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + slot;
    }
    const int m0 = (logical / p.tiles_n) * BMT;
    const int n0 = (logical % p.tiles_n) * BNT;
    const int zs = blockIdx.z;

    const int nkt = p.K / AD_BK;
    int kt_begin = 0, kt_end = nkt;
    if (p.split_k > 1) { kt_begin = zs * p.ktiles_per_split; kt_end = min(nkt, kt_begin + p.ktiles_per_split); }

    // global row of W behind row r (0..127) of the staged tile (SwiGLU: band layout [u1 cols | u3 cols] per wave band, epilogue_swiglu)
    auto brow = [&](int r) -> int {
        if (p.act == GAOT_ACT_SWIGLU) {
            const int F = p.N >> 1, within = r % WN;
            const int gcol = (n0 >> 1) + (r / WN) * (WN / 2) + within % (WN / 2);
            return (within / (WN / 2)) * F + min(gcol, F - 1);
        }
        return min(n0 + r, p.N - 1);
    };

    // ---- DMA sources: piece q of this wave covers 8 rows x 128 B of a slot; lane -> (row, physical chunk), the chunk it FETCHES is the
    // swizzled one
    const float* a_src[LA];
    const float* a2_src[LA];
    const unsigned short* b_src[LB];
#pragma unroll
    for (int q = 0; q < LA; ++q) {
        const int t = (q * 4 + wave) * 64 + lane;
        const int row = t >> 3, pc = t & 7, lc = pc ^ ((row >> 1) & 7);
        const long grow = min(m0 + row, p.M - 1);
        a_src[q] = p.A + grow * p.lda + lc * 4;
        a2_src[q] = p.A2 != nullptr ? p.A2 + grow * p.lda2 + lc * 4 - p.k_split : nullptr;
    }
#pragma unroll
    for (int q = 0; q < LB; ++q) {
        const int t = (q * 4 + wave) * 64 + lane;
        const int row = t >> 3, pc = t & 7, lc = pc ^ ((row >> 1) & 7);
        b_src[q] = p.Bpl + (long)brow(row) * p.ld_bpl + lc * 8;
    }
    auto issue = [&](int kt, int slot) {
        const long k0 = (long)min(kt, kt_end - 1) * AD_BK;          // past the end: the last tile again (never read)
        const bool second = p.A2 != nullptr && k0 >= p.k_split;
        unsigned char* sa = smem_raw + slot * STAGE;
        if ((ABL & 8) && kt > kt_begin + 1) return;          // tuning: no operand traffic in the loop (the counted waits pass at once)
#pragma unroll
        for (int q = 0; q < LA; ++q)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((second ? a2_src[q] : a_src[q]) + k0),
                                             (__attribute__((address_space(3))) void*)(sa + (q * 4 + wave) * 1024), 16, 0, 0);
#pragma unroll
        for (int q = 0; q < LB; ++q)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[q] + k0 * 2),
                                             (__attribute__((address_space(3))) void*)(sa + A_ST + (q * 4 + wave) * 1024), 16, 0, 0);
    };

    // ---- fragment read offsets inside a slot: row * 128 + ((chunk ^ swizzle) << 4)
    //   A (fp32): chunk = k-step * 4 + lh * 2 + c (c = 0, 1: the lane's 8 consecutive k);  W planes: chunk = k-step * 4 + piece * 2 + lh
    const unsigned lds_base = (unsigned)reinterpret_cast<unsigned long>((__attribute__((address_space(3))) void*)smem_raw);
    unsigned a_rd[TM][2][2], b_rd[TN][2][2];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = wm * WM + i * 32 + li, sw = (row >> 1) & 7;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int c = 0; c < 2; ++c) a_rd[i][ks][c] = lds_base + row * 128 + (((ks * 4 + lh * 2 + c) ^ sw) << 4);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = wn * WN + j * 32 + li, sw = (row >> 1) & 7;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) b_rd[j][ks][pl] = lds_base + A_ST + row * 128 + (((ks * 4 + pl * 2 + lh) ^ sw) << 4);
    }
    auto read_a = [&](int slot, int ks, AdRaw<TM>& r) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            asm volatile("ds_read_b128 %0, %1" : "=v"(r.v[i][0]) : "v"(a_rd[i][ks][0] + slot * STAGE));
            asm volatile("ds_read_b128 %0, %1" : "=v"(r.v[i][1]) : "v"(a_rd[i][ks][1] + slot * STAGE));
        }
    };
    auto read_b = [&](int slot, int ks, AdFrag<TN>& f) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            asm volatile("ds_read_b128 %0, %1" : "=v"(f.h[j]) : "v"(b_rd[j][ks][0] + slot * STAGE));
            asm volatile("ds_read_b128 %0, %1" : "=v"(f.m[j]) : "v"(b_rd[j][ks][1] + slot * STAGE));
        }
    };

    float sc_a = 1.f, so_a = 1.f, sc_b = 1.f, so_b = 1.f;
    unsigned ec_a = 0u;          // range watch: sum of the biased exponents of this thread's non-zero group maxima (low 20 bits) and their count
    auto split_a = [&](const AdRaw<TM>& r, AdPieces<TM>& o, bool track) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    unsigned h_, m_;
                    if (ABL & 1) { h_ = __float_as_uint(r.v[i][c][2 * e]); m_ = __float_as_uint(r.v[i][c][2 * e + 1]); }
                    else split2h_pair_gemm(r.v[i][c][2 * e], r.v[i][c][2 * e + 1], sc_a, h_, m_);
                    o.h[i][2 * c + e] = h_; o.m[i][2 * c + e] = m_;
                }
            if (DETECT && track) {          // (every other group of 8: the mean over half the groups is as good a typical magnitude)
                const f32x4 x = r.v[i][0], y = r.v[i][1];
                const float t = fmaxf(fmaxf(fmaxf(fabsf(x[0]), fabsf(x[1])), fmaxf(fabsf(x[2]), fabsf(x[3]))),
                                      fmaxf(fmaxf(fabsf(y[0]), fabsf(y[1])), fmaxf(fabsf(y[2]), fabsf(y[3]))));
                const unsigned e = __float_as_uint(t) >> 23;          // 0 for a zero (or denormal) group: exempt
                ec_a += e + (e != 0u ? (1u << 20) : 0u);
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // MFMAs [g0, g1) of one k-step's TM * TN * 3 (order: i, j, then m h / h m / h h -- gemm_split.hip's)
    auto mfmas = [&](const AdPieces<TM>& a, const AdFrag<TN>& b, int g0, int g1) {
        if (ABL & 2) return;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int g = (i * TN + j) * 3;
                if (g >= g0 && g < g1) acc[i][j] = ad_mfma(a.m[i], b.h[j], acc[i][j]);
                if (g + 1 >= g0 && g + 1 < g1) acc[i][j] = ad_mfma(a.h[i], b.m[j], acc[i][j]);
                if (g + 2 >= g0 && g + 2 < g1) acc[i][j] = ad_mfma(a.h[i], b.h[j], acc[i][j]);
            }
    };
    constexpr int NMF = TM * TN * 3;          // MFMAs per k-step

    AdRaw<TM> ra0, ra1;
    AdPieces<TM> pc0, pc1;
    AdFrag<TN> bf0, bf1;

    // ---- prologue
    {
        unsigned wa = amax_peek(p.a_amax);
        if (p.A2 != nullptr) { const unsigned w2 = amax_peek(p.a2_amax); wa = w2 > wa ? w2 : wa; }
        const unsigned wb = amax_peek(p.b_amax);
        issue(kt_begin, 0);
        if (NS == 3) issue(kt_begin + 1, 1);
        amax_finish(wa, sc_a, so_a);
        amax_finish(wb, sc_b, so_b);
    }

    f32x16 tot[FL ? TM : 1][FL ? TN : 1];
    if (FL) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) tot[i][j][r] = 0.f;
    }
    int slot = 0;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        // tile kt has landed (this wave's pieces; with three slots tile kt + 1 may still be in flight)
        if (NS == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LA + LB) : "memory");
        else         asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();          // everyone's pieces have landed; nobody reads the slot refilled next any more
        asm volatile("" ::: "memory");
        const int nslot = NS == 3 ? (slot == 0 ? 2 : slot - 1) : (slot ^ 1);          // (slot + NS - 1) % NS
        // k-step 0: its rows and fragments; the first split is exposed (the other workgroup's MFMAs run meanwhile)
        read_a(slot, 0, ra0);
        read_a(slot, 1, ra1);
        read_b(slot, 0, bf0);
        ad_lgkm_wait<2 * TM + 2 * TN>(); ad_pin(ra0);
        split_a(ra0, pc0, true);
        __builtin_amdgcn_sched_barrier(0);
        read_b(slot, 1, bf1);
        ad_lgkm_wait<2 * TN>(); ad_pin(ra1); ad_pin(bf0);
        // beside k-step 0's MFMAs: k-step 1's pieces and the DMA issue of the tile that refills the slot everyone has left (an issue
        // costs the wave 60-100 cycles: hidden under an MFMA each instead of exposed at the head of the tile)
        issue(kt + NS - 1, nslot);
        split_a(ra1, pc1, false);
        mfmas(pc0, bf0, 0, NMF);
#pragma unroll
        for (int g = 0; g < NMF; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, (TM * 36 + NMF - 1) / NMF, 0);
            if (g < LA + LB) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        ad_lgkm_wait<0>(); ad_pin(bf1);
        mfmas(pc1, bf1, 0, NMF);
        __builtin_amdgcn_sched_barrier(0);
        slot = slot == NS - 1 ? 0 : slot + 1;
        if (FL && ((kt - kt_begin) & 31) == 31) {          // wave-uniform: 32 k-tiles of 32 = the accumulation cap
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { tot[i][j][r] += acc[i][j][r]; acc[i][j][r] = 0.f; }
        }
    }
    if (FL) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] += tot[i][j][r];
    }
    // The range watch's verdict rides on the barrier the epilogue needs anyway: every wave leaves its count of threads that saw the
    // spread (and whether any saw a stale word) in LDS BEFORE it, everyone reads the four entries after it (__syncthreads_count +
    // __syncthreads_or are six barriers: ~1 us per workgroup, 8 us on a launch of four workgroup rounds)
    __shared__ int s_vote[4];
    if (DETECT) {
        const unsigned es = ec_a & 0xfffffu, cn = ec_a >> 20;
        int la = 0;
        if (cn != 0u) {
            const float mean_e = (float)es / (float)cn - 127.f + (float)((int)(__float_as_uint(sc_a) >> 23) - 127);
            la = (int)((13.5f - mean_e) * 16.f);
        }
        const int lb_w = (int)(fminf(fmaxf(__uint_as_float(reinterpret_cast<const unsigned*>(p.b_amax)[p.bpl_flag]), 0.f), 64.f) * 16.f);      // (clamped: a NaN or a garbage verdict is a number of binades, never an int overflow)
        const bool bad = la + lb_w > 17 * 16;
        bool stale = la < -24;          // a mean 1.5 binades above the claimed maximum: a stale word
        if ((kt_end - kt_begin) * TM >= 4096) stale = true;          // (a k range too long for the packed counters: play safe)
        const int nb = __popcll(__ballot(bad)), ns = __ballot(stale) != 0ull ? 1 : 0;
        if (lane == 0) s_vote[wave] = nb | (ns << 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the ring's last (unused) pieces have landed: the epilogue reuses the LDS
    __syncthreads();

    if (DETECT) {
        const int v = s_vote[0] + s_vote[1] + s_vote[2] + s_vote[3];
        if (__builtin_expect((v >> 16) != 0 || (v & 0xffff) * 8 >= 256, 0)) {
            // ---- the second pass: fp32 operands straight from memory into v_mfma_f32_32x32x2_f32 (lane (i, h) supplies k = 4 h + q of
            // each group of 8 for both operands) -- no pieces, no scales, nothing that depends on the operands' range
            if (tid == 0) atomicAdd(&g_ad_redo_tiles, 1u);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            long a_off[TM], b_off[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a_off[i] = (long)min(m0 + wm * WM + i * 32 + li, p.M - 1);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int r = wn * WN + j * 32 + li;
                b_off[j] = BREAL ? (long)brow(r) * p.ldb + 4 * lh : (long)(4 * lh) * p.ldb + min(n0 + r, p.N - 1);
            }
            for (int kt = kt_begin; kt < kt_end; ++kt) {
                const long k0 = (long)kt * AD_BK;
                const bool second = p.A2 != nullptr && k0 >= p.k_split;
#pragma unroll
                for (int g8 = 0; g8 < 4; ++g8) {
                    f32x4 a[TM], b[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        a[i] = *reinterpret_cast<const f32x4*>((second ? p.A2 + a_off[i] * p.lda2 - p.k_split : p.A + a_off[i] * p.lda) + k0 + 8 * g8 + 4 * lh);
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        if (BREAL) b[j] = *reinterpret_cast<const f32x4*>(p.B + b_off[j] + k0 + 8 * g8);
                        else {
#pragma unroll
                            for (int q = 0; q < 4; ++q) b[j][q] = p.B[b_off[j] + (k0 + 8 * g8 + q) * p.ldb];
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][q], b[j][q], acc[i][j], 0, 0, 0);
                }
            }
            so_a = 1.f; so_b = 1.f;
        }
    }

    if (ABL & 4) { if (acc[0][0][0] + acc[TM - 1][TN - 1][3] == 123.456f) p.C[tid] = 1.f; return; }
    if (p.act == GAOT_ACT_SWIGLU) epilogue_swiglu<TM, TN, WM, WN>(p, reinterpret_cast<float*>(smem_raw), acc, m0, n0, wm, wn, wave, lane, so_a, so_b);
    else epilogue_vec<TM, TN, WM, WN>(p, reinterpret_cast<float*>(smem_raw), acc, m0, n0, wm, wn, wave, lane, zs, so_a, so_b);
}

unsigned ad_redo_count(bool reset) {
    unsigned v = 0;
    hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_ad_redo_tiles), sizeof(v));
    if (reset) { const unsigned z = 0; hipMemcpyToSymbol(HIP_SYMBOL(g_ad_redo_tiles), &z, sizeof(z)); }
    return v;
}

// bm: 128 or 64; bn: 128, or 64 with bm = 64.  Needs: A k-contiguous (16-byte aligned rows), pre-split planes, K % 32 == 0, the vector epilogue
void launch_ad(GemmArgs& a, bool b_kmajor, hipStream_t st, int bm, int bn) {
    a.tiles_m = cdiv(a.M, bm);
    a.tiles_n = cdiv(a.N, bn);
    a.bpl_flag = b_kmajor ? 1 : 2;
    dim3 grid(a.tiles_m * a.tiles_n, 1, a.split_k > 1 ? a.split_k : 1), block(256);
    const long k_per_wg = a.split_k > 1 ? (long)a.ktiles_per_split * AD_BK : a.K;
    if (bm == 64 && bn == 64 && k_per_wg > 1024) {          // (the dispatcher sends reductions past the cap here only: gemm.hip ad_narrow)
        if (b_kmajor) hipLaunchKernelGGL((gemm_ad_kernel<64, 2, 3, true, 0, 64, true>), grid, block, 0, st, a);
        else          hipLaunchKernelGGL((gemm_ad_kernel<64, 2, 3, false, 0, 64, true>), grid, block, 0, st, a);
    } else
    if (bm == 64 && bn == 64) {
        if (b_kmajor) hipLaunchKernelGGL((gemm_ad_kernel<64, 2, 3, true, 0, 64>), grid, block, 0, st, a);
        else          hipLaunchKernelGGL((gemm_ad_kernel<64, 2, 3, false, 0, 64>), grid, block, 0, st, a);
    } else if (bm == 64) {
        if (b_kmajor) hipLaunchKernelGGL((gemm_ad_kernel<64, 2, 3, true>), grid, block, 0, st, a);
        else          hipLaunchKernelGGL((gemm_ad_kernel<64, 2, 3, false>), grid, block, 0, st, a);
    } else {
        if (b_kmajor) hipLaunchKernelGGL((gemm_ad_kernel<128, 4, 2, true>), grid, block, 0, st, a);
        else          hipLaunchKernelGGL((gemm_ad_kernel<128, 4, 2, false>), grid, block, 0, st, a);
    }
}

}  // namespace gaot
