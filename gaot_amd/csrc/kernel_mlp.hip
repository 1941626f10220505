// Fused kernel MLP of the graph neural operator (reference mlp.py:307-337 via agno.py:229-231): per edge
//     k_e = W_L ... gelu(W_2 gelu(W_1 [y_j, x_i] + b_1) + b_2) ... + b_L          (every width 64, c_in <= 16)
// as ONE kernel forward and ONE kernel backward instead of a chain of skinny GEMM launches over the E ~ 5*10^4 edge rows.
//
// Orientation: activations are kept TRANSPOSED, [feature][edge], so that a layer is  Z^T = W * H^T  on
// v_mfma_f32_32x32x2_f32 with A = W (lane -> output feature) and B = H^T (lane -> edge).  The MFMA result fragment
// (lane = edge column, 16 registers = features crow(r, hi)) then IS the B operand of the next layer: the k-slot of
// half-wave hi is defined as feature crow(t, hi), the weights are read to match, and no activation ever moves between
// lanes or through LDS on the forward chain / the input-gradient chain (same trick as P in the attention kernels).
// A wave owns 32 edges; weights sit in LDS once per workgroup as plain [64][68] images that serve both W (forward,
// ds_read_b128 along k) and W^T (backward, ds_read_b32 along the output feature).
//
// Backward recomputes the forward chain from the edge coordinates (keeps the pre-activations in registers), then per
// layer: G and H go through a wave-private LDS tile [feature][edge] (the weight gradient reduces over EDGES, so edges
// must become the k index), dW_m += G_m H_{m-1}^T on MFMA into per-wave accumulators that live for the whole kernel
// (192 registers for 3 layers: the accumulator half of the register file), db_m and dW_1 (c_in columns) from the same
// operand registers on the VALU, dH = W^T G from registers again.  Workgroups are persistent; per-workgroup partial
// gradients go to a workspace and the short-matrix column sum (pointwise.hip) adds them in fixed order (deterministic, no atomics).
#include "common.h"

namespace gaot {

constexpr int KM_WLD = 68;     // floats per LDS weight row (16-byte aligned rows, conflict-free b128 / b32 reads)
constexpr int KM_TLD = 36;     // floats per LDS [feature][32 edges] tile row
constexpr int KM_MAXC = 16;    // c_in of the first layer

struct KMArgs {
    const float* x; int cin; int E;
    const float* w1; const float* b1;
    const float* w[3]; const float* b[3];
    float* out;            // forward: k [E, 64]
    const float* dk;       // backward: upstream gradient [E, 64]
    float* ws;             // backward: per-workgroup partial gradients [grid][psize]
    int psize;
    int ntiles;            // 128-edge tiles
    int abl;               // tuning only (gaot_debug_set_kernel_mlp_ablate): forward 1 = no stores, 2 = no GELU, 4 = no MFMA layers, 8 = no weight staging
    int cout;              // width of the LAST layer (<= 64, multiple of 4): only the first cout columns are stored / read
    int ldw[4];            // row stride of layer i's weight matrix (>= its input width: a column block of a wider matrix is fine)
    int wo[4];             // output width of layer i (<= 64, multiples of 4; layer i + 1 reads that many inputs): weights are [wo[i]][wo[i-1]]
                           // row-major; rows / columns past the widths are staged as ZEROS, so the chain runs at width 64 throughout
                           // (act(0) = 0 for GELU and ReLU: padded features carry nothing forward, and their gradients come out zero)
};

__device__ __forceinline__ void gelu_both(float z, float& h, float& d) {
    const float cdf = 0.5f * (1.0f + erf_nb(z * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __builtin_amdgcn_exp2f(-0.72134752044448170368f * z * z);
    h = z * cdf;
    d = cdf + z * pdf;
}

// hidden activation: ACT = GAOT_ACT_GELU (kernel MLP of the integral transform) or GAOT_ACT_RELU (geometry embedding)
template <int ACT> __device__ __forceinline__ float act_f(float z) { return ACT == GAOT_ACT_RELU ? fmaxf(z, 0.f) : gelu_f(z); }
template <int ACT> __device__ __forceinline__ void act_both(float z, float& h, float& d) {
    if (ACT == GAOT_ACT_RELU) { h = fmaxf(z, 0.f); d = z > 0.f ? 1.f : 0.f; }
    else gelu_both(z, h, d);
}

// crow(r, hi) = crow0(r) + 4 hi: addressing through (base + 4 hi) [crow0(r) * stride] keeps ONE address register and constant offsets
// (written as crow(r, hi) the compiler turns the sum into an OR it cannot fold and hoists one address per r out of the tile loop)
__device__ __forceinline__ constexpr int crow0(int r) { return (r & 3) + 8 * (r >> 2); }

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// stage the weights of the workgroup: Ws[m] = W_{m+1} as [64][KM_WLD], W1s [64][cin], biases
template <int NL, int CM>
__device__ __forceinline__ void km_stage_weights(const KMArgs& p, float* Ws, float* W1s, float* Bs, int tid) {
#pragma unroll
    for (int m = 0; m < NL; ++m)
        for (int i = tid; i < 64 * 16; i += 256) {          // 16 float4 per row
            const int r = i >> 4, c4 = (i & 15) * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (r < p.wo[m + 1] && c4 < p.wo[m]) v = *reinterpret_cast<const f32x4*>(p.w[m] + r * p.ldw[m + 1] + c4);
            *reinterpret_cast<f32x4*>(Ws + m * 64 * KM_WLD + r * KM_WLD + c4) = v;
        }
    for (int i = tid; i < 64 * CM; i += 256) { const int f = i / CM, c = i % CM; W1s[i] = (c < p.cin && f < p.wo[0]) ? p.w1[f * p.ldw[0] + c] : 0.f; }   // rows padded to CM
    if (tid < 64) {
        Bs[tid] = tid < p.wo[0] ? p.b1[tid] : 0.f;
#pragma unroll
        for (int m = 0; m < NL; ++m) Bs[64 * (m + 1) + tid] = tid < p.wo[m + 1] ? p.b[m][tid] : 0.f;
    }
}

// first layer on the VALU: z0[kt][t] = b1[f] + sum_c W1[f][c] x[c], f = kt*32 + crow(t, hi)
template <int CM>
__device__ __forceinline__ void km_layer0(const float* W1s, const float* Bs, const float (&xr)[CM], int cin, int hi, f32x16 (&z)[2]) {
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int f0 = kt * 32 + crow0(t);           // feature f0 + 4 hi: one address register, constant offsets
            float v = (Bs + 4 * hi)[f0];
            const float* wr = (W1s + 4 * hi * CM) + f0 * CM;      // zero-padded row: whole 16-byte chunks, no per-column predicate
#pragma unroll
            for (int c4 = 0; c4 < CM / 4; ++c4) {
                const f32x4 w4 = *reinterpret_cast<const f32x4*>(wr + 4 * c4);
                v += (w4[0] * xr[4 * c4] + w4[1] * xr[4 * c4 + 1]) + (w4[2] * xr[4 * c4 + 2] + w4[3] * xr[4 * c4 + 3]);
            }
            z[kt][t] = v;
            if ((t & 3) == 3) __builtin_amdgcn_sched_barrier(0);     // keep the scheduler from hoisting all 32 weight-row reads
        }
}

// one MFMA layer: acc[io] = b + W * h   (h: [kt][t] fragment registers of this lane's edge)
__device__ __forceinline__ void km_layer(const float* W, const float* bias, const f32x16 (&h)[2], int li, int hi, f32x16 (&acc)[2]) {
#pragma unroll
    for (int io = 0; io < 2; ++io)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[io][r] = bias[io * 32 + crow(r, hi)];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 a[2];
#pragma unroll
            for (int io = 0; io < 2; ++io)
                a[io] = *reinterpret_cast<const f32x4*>(W + (io * 32 + li) * KM_WLD + kt * 32 + 8 * q + 4 * hi);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int io = 0; io < 2; ++io)
                    acc[io] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[io][s], h[kt][4 * q + s], acc[io], 0, 0, 0);
        }
}

template <int CM>
__device__ __forceinline__ void km_load_x(const KMArgs& p, int e, float (&xr)[CM]) {
    const float* src = p.x + (long)e * p.cin;
    if (CM == 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src);
        xr[0] = v[0]; xr[1] = v[1]; xr[2] = v[2]; xr[3] = v[3];
    } else {
#pragma unroll
        for (int c = 0; c < CM; ++c) xr[c] = c < p.cin ? src[c] : 0.f;
    }
}

// ------------------------------------------------------------------------------------------ forward
template <int NL, int CM, int ACT>
__global__ __launch_bounds__(256) void kernel_mlp_fwd_kernel(const KMArgs p) {
    __shared__ __attribute__((aligned(16))) float Ws[NL * 64 * KM_WLD];
    __shared__ __attribute__((aligned(16))) float W1s[64 * KM_MAXC];
    __shared__ float Bs[64 * (NL + 1)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, hi = lane >> 5;
    if (!(p.abl & 8)) km_stage_weights<NL, CM>(p, Ws, W1s, Bs, tid);
    __syncthreads();
    const int e0 = (blockIdx.x * 4 + wave) * 32;
    if (e0 >= p.E) return;
    const int e = min(e0 + li, p.E - 1);
    float xr[CM];
    km_load_x<CM>(p, e, xr);
    f32x16 z[2], h[2];
    km_layer0<CM>(W1s, Bs, xr, p.cin, hi, z);
#pragma unroll
    for (int m = 0; m < NL; ++m) {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int t = 0; t < 16; ++t) h[kt][t] = (p.abl & 2) ? z[kt][t] : act_f<ACT>(z[kt][t]);
        if (p.abl & 4) { z[0] = h[0]; z[1] = h[1]; }
        else km_layer(Ws + m * 64 * KM_WLD, Bs + 64 * (m + 1), h, li, hi, z);
    }
    if (e0 + li < p.E && (!(p.abl & 1) || z[0][0] == 123.456f)) {
        float* dst = p.out + (long)(e0 + li) * p.cout;
#pragma unroll
        for (int io = 0; io < 2; ++io)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (io * 32 + 8 * q + 4 * hi < p.cout)
                    *reinterpret_cast<f32x4*>(dst + io * 32 + 8 * q + 4 * hi) = f32x4{z[io][4 * q], z[io][4 * q + 1], z[io][4 * q + 2], z[io][4 * q + 3]};
    }
}

// ------------------------------------------------------------------------------------------ backward
// partial-gradient vector of a workgroup: [m = 0..NL-1] dW_{m+1} [64][64] | dW_1 [64][cin] | db_1 [64] | db_{m+1} [64]
//
// A workgroup walks 128-edge tiles; wave w owns edges [32w, 32w+32) of the tile for the recompute and the input-gradient
// chain (registers only), and QUADRANT (io_w, kt_w) = (w >> 1, w & 1) of every layer's weight gradient over all 128
// edges (operands from the workgroup's [feature][128 edges] LDS tiles), so its accumulators are 16 registers per layer.
constexpr int KM_TLD128 = 132;   // floats per row of a [64 features][128 edges] tile
template <int NL, int CM, int ACT>
__global__ __launch_bounds__(256, 1) void kernel_mlp_bwd_kernel(const KMArgs p) {
    constexpr int WS_FLOATS = NL * 64 * KM_WLD;
    constexpr int TILE = 64 * KM_TLD128;
    __shared__ __attribute__((aligned(16))) float Ws[WS_FLOATS];
    __shared__ __attribute__((aligned(16))) float W1s[64 * KM_MAXC];
    __shared__ float Bs[64 * (NL + 1)];
    __shared__ __attribute__((aligned(16))) float Gt[TILE];
    __shared__ __attribute__((aligned(16))) float Ht[TILE];
    __shared__ __attribute__((aligned(16))) float Xt[128 * KM_MAXC];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, hi = lane >> 5;
    const int io_w = wave >> 1, kt_w = wave & 1;
    const int cin = p.cin;
    km_stage_weights<NL, CM>(p, Ws, W1s, Bs, tid);
    __syncthreads();

    f32x16 dW[NL];                       // quadrant (io_w, kt_w): rows = output feature crow(r, hi), column = input feature li
#pragma unroll
    for (int m = 0; m < NL; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) dW[m][r] = 0.f;
    float db[NL + 1];                    // feature io_w*32 + li, this half-wave's edges (MFMA layers: kept by kt_w == 0 waves)
    float dw1[CM];                       // feature io_w*32 + li, edges of half (kt_w) of the tile
#pragma unroll
    for (int m = 0; m <= NL; ++m) db[m] = 0.f;
#pragma unroll
    for (int c = 0; c < CM; ++c) dw1[c] = 0.f;

    for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        asm volatile("" ::: "memory");      // the weight images are loop-invariant: stop LICM from parking them in 400 registers
        const int e0 = tile * 128 + wave * 32;
        const bool valid = e0 + li < p.E;
        const int e = min(e0 + li, p.E - 1);
        float xr[CM];
        km_load_x<CM>(p, e, xr);
        if (hi == 0) {
#pragma unroll
            for (int c = 0; c < CM; ++c) Xt[(wave * 32 + li) * KM_MAXC + c] = valid ? xr[c] : 0.f;          // xr is zero beyond c_in
        }
        // recompute the forward chain, keeping the pre-activations of the GELU layers
        f32x16 z[NL][2];
        km_layer0<CM>(W1s, Bs, xr, cin, hi, z[0]);
#pragma unroll
        for (int m = 1; m < NL; ++m) {
            f32x16 h[2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    h[kt][t] = act_f<ACT>(z[m - 1][kt][t]);
                    if ((t & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                }
            km_layer(Ws + (m - 1) * 64 * KM_WLD, Bs + 64 * m, h, li, hi, z[m]);
        }
        // upstream gradient in fragment layout (zero for edges past the end: they add nothing to any gradient)
        f32x16 g[2];
        {
            const float* src = p.dk + (long)e * p.cout;
#pragma unroll
            for (int io = 0; io < 2; ++io)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (valid && io * 32 + 8 * q + 4 * hi < p.cout) v = *reinterpret_cast<const f32x4*>(src + io * 32 + 8 * q + 4 * hi);
                    g[io][4 * q] = v[0]; g[io][4 * q + 1] = v[1]; g[io][4 * q + 2] = v[2]; g[io][4 * q + 3] = v[3];
                }
        }
        // layers NL .. 1 (layer m: weights Ws[m-1], input H_{m-1} = gelu(Z_{m-1}))
#pragma unroll
        for (int m = NL; m >= 1; --m) {
            // (1) G_m -> workgroup tile.  (2) dH_{m-1} = W_m^T G_m straight from registers: 64 MFMAs with nothing but weight
            // reads around them, so the H_{m-1} = gelu(Z_{m-1}) / gelu'(Z_{m-1}) evaluation (VALU, independent) is issued in
            // their shadow; Z_{m-1} is replaced in place by gelu'.  (3) H tile, workgroup sync, dW_m quadrant on MFMA.
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int t = 0; t < 16; ++t) Gt[(kt * 32 + crow(t, hi)) * KM_TLD128 + wave * 32 + li] = g[kt][t];
            f32x16 dh[2];
#pragma unroll
            for (int io = 0; io < 2; ++io)
#pragma unroll
                for (int r = 0; r < 16; ++r) dh[io][r] = 0.f;
            const float* W = Ws + (m - 1) * 64 * KM_WLD;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const float* wr = W + (kt * 32 + crow(t, hi)) * KM_WLD + li;
#pragma unroll
                    for (int io = 0; io < 2; ++io)
                        dh[io] = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[io * 32], g[kt][t], dh[io], 0, 0, 0);
                    // one GELU pair per MFMA pair (same element index: its Z is dead for the MFMAs)
                    float hv, dv;
                    act_both<ACT>(z[m - 1][kt][t], hv, dv);
                    z[m - 1][kt][t] = dv;
                    Ht[(kt * 32 + crow(t, hi)) * KM_TLD128 + wave * 32 + li] = hv;
                    if ((t & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                }
            __syncthreads();
            // dW_m quadrant += G_m[io_w rows] H_{m-1}[kt_w rows]^T over the 128 edges (k = 8q + 4hi + s)
            {
                const float* ga = Gt + (io_w * 32 + li) * KM_TLD128 + 4 * hi;
                const float* hb = Ht + (kt_w * 32 + li) * KM_TLD128 + 4 * hi;
                float dsum = 0.f;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(ga + 8 * q);
                    const f32x4 b = *reinterpret_cast<const f32x4*>(hb + 8 * q);
                    dsum += (a[0] + a[1]) + (a[2] + a[3]);
#pragma unroll
                    for (int s = 0; s < 4; ++s) dW[m - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], dW[m - 1], 0, 0, 0);
                }
                db[m] += dsum;
            }
#pragma unroll
            for (int io = 0; io < 2; ++io)
#pragma unroll
                for (int r = 0; r < 16; ++r) g[io][r] = dh[io][r] * z[m - 1][io][r];
            __syncthreads();          // the tiles are rewritten by the next layer
        }
        // first layer: dW_1 = G_0 x^T, db_1: wave (io_w, kt_w) takes feature tile io_w over edge half kt_w of the tile
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int t = 0; t < 16; ++t) Gt[(kt * 32 + crow(t, hi)) * KM_TLD128 + wave * 32 + li] = g[kt][t];
        __syncthreads();
        {
            const float* ga = Gt + (io_w * 32 + li) * KM_TLD128 + kt_w * 64 + 4 * hi;
            float dsum = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(ga + 8 * q);
                dsum += (a[0] + a[1]) + (a[2] + a[3]);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const float* xe = Xt + (kt_w * 64 + 8 * q + 4 * hi + s) * KM_MAXC;
#pragma unroll
                    for (int c4 = 0; c4 < CM / 4; ++c4) {             // x rows are zero beyond c_in
                        const f32x4 xv = *reinterpret_cast<const f32x4*>(xe + 4 * c4);
                        dw1[4 * c4] += a[s] * xv[0]; dw1[4 * c4 + 1] += a[s] * xv[1]; dw1[4 * c4 + 2] += a[s] * xv[2]; dw1[4 * c4 + 3] += a[s] * xv[3];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            db[0] += dsum;
        }
        __syncthreads();
    }

    // ---- the workgroup's partial: weight-gradient quadrants straight from their owners; the small vectors through LDS
    float* dst = p.ws + (long)blockIdx.x * p.psize;
#pragma unroll
    for (int m = 0; m < NL; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[m * 4096 + (io_w * 32 + crow(r, hi)) * 64 + kt_w * 32 + li] = dW[m][r];
    float* R = Gt;                                   // [64][cin] dW_1 | db_1 [64] | db_m [64] ...
    const int off_b = 64 * cin, nsmall = 64 * cin + 64 * (NL + 1);
#pragma unroll
    for (int c = 0; c < CM; ++c) dw1[c] += __shfl_xor(dw1[c], 32, 64);       // fold the two half-waves (different edges)
#pragma unroll
    for (int m = 0; m <= NL; ++m) db[m] += __shfl_xor(db[m], 32, 64);
    for (int pass = 0; pass < 2; ++pass) {            // edge half 0 stores, edge half 1 adds
        if (kt_w == pass && hi == 0) {
#pragma unroll
            for (int c = 0; c < CM; ++c)
                if (c < cin) { float* d = R + (io_w * 32 + li) * cin + c; *d = pass == 0 ? dw1[c] : *d + dw1[c]; }
            float* d0 = R + off_b + io_w * 32 + li;
            *d0 = pass == 0 ? db[0] : *d0 + db[0];
            if (pass == 0) {
#pragma unroll
                for (int m = 1; m <= NL; ++m) R[off_b + m * 64 + io_w * 32 + li] = db[m];
            }
        }
        __syncthreads();
    }
    for (int i = tid; i < nsmall; i += 256) dst[NL * 4096 + i] = R[i];
}

// =========================================================================================== bf16-split variants (default)
// The same two kernels with the 64-wide layers of the forward chain, of the recompute and of the input-gradient chain on
// v_mfma_f32_32x32x16_bf16: each fp32 operand is split exactly into three bf16 pieces (common.h), six piece products per k-step give
// the fp32-level product at 3 / 8 of the matrix-pipe time of the fp32 MFMA (6 x 8 passes per 16 k against 8 x 16 passes).  The
// orientation trick of the fp32 kernels carries over unchanged: k-slot e of half-wave hi in k-step j is feature
// 16 j + 8 (e >> 2) + 4 hi + (e & 3), i.e. fragment registers 8 (j & 1) .. + 7 of tile j >> 1, so the B operand is split straight out
// of the previous layer's result registers.  Weights sit in LDS as three bf16 planes [64][64] per layer (split once per workgroup);
// the A operand is two 8-byte reads per plane (forward) or two TRANSPOSING reads (ds_read_b64_tr_b16: a lane supplies the address of 4
// consecutive columns of one row and receives 4 consecutive rows of one column; tools/probe/tr_read.hip prints the lane map) for W^T
// in the input-gradient chain.  The weight gradient stays on the fp32 MFMA with fp32 [feature][edge] tiles: it reduces over the
// EDGES, with cancellation, so its operands need all 24 bits, and six bf16 planes of a 128-edge tile do not fit next to the weight
// planes (two ROUNDED pieces per operand did fit, were 12 % faster and cost 7.6e-6 of relative error on every dW: not taken).
constexpr int KS_LDB = 136;                   // bytes per row of a 64-column bf16 plane (+8: conflict-free 8-byte row reads and edge-row stores)
constexpr int KS_PLANE = 64 * KS_LDB;         // one weight plane
__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// the six piece products, smallest terms first
__device__ __forceinline__ f32x16 mfma_split6(const u32x4 (&a)[3], const u32x4 (&b)[3], f32x16 c) {
    c = mfma_bf16(a[2], b[0], c);
    c = mfma_bf16(a[0], b[2], c);
    c = mfma_bf16(a[1], b[1], c);
    c = mfma_bf16(a[1], b[0], c);
    c = mfma_bf16(a[0], b[1], c);
    return mfma_bf16(a[0], b[0], c);
}
// NP pieces per operand: 3 = exact, 2 = two rounded pieces (common.h split2_pair); layer m's planes start at m * NP * KS_PLANE
template <int NP>
__device__ __forceinline__ void km_split(float x0, float x1, unsigned& a, unsigned& b, unsigned& c) {
    if (NP == 3) split3_pair(x0, x1, a, b, c);
    else { split2_pair(x0, x1, a, b); c = 0u; }
}

template <int NL, int CM, int NP = 3>
__device__ __forceinline__ void km_stage_weights_split(const KMArgs& p, unsigned char* Wp, float* W1s, float* Bs, int tid) {
#pragma unroll
    for (int m = 0; m < NL; ++m)
        for (int i = tid; i < 64 * 16; i += 256) {
            const int r = i >> 4, c4 = (i & 15) * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (r < p.wo[m + 1] && c4 < p.wo[m]) v = *reinterpret_cast<const f32x4*>(p.w[m] + r * p.ldw[m + 1] + c4);
            unsigned h0, m0, l0, h1, m1, l1;
            km_split<NP>(v[0], v[1], h0, m0, l0);
            km_split<NP>(v[2], v[3], h1, m1, l1);
            unsigned char* d = Wp + m * NP * KS_PLANE + r * KS_LDB + c4 * 2;
            *reinterpret_cast<u32x2*>(d) = u32x2{h0, h1};
            *reinterpret_cast<u32x2*>(d + KS_PLANE) = u32x2{m0, m1};
            if (NP == 3) *reinterpret_cast<u32x2*>(d + 2 * KS_PLANE) = u32x2{l0, l1};
        }
    for (int i = tid; i < 64 * CM; i += 256) { const int f = i / CM, c = i % CM; W1s[i] = (c < p.cin && f < p.wo[0]) ? p.w1[f * p.ldw[0] + c] : 0.f; }
    if (tid < 64) {
        Bs[tid] = tid < p.wo[0] ? p.b1[tid] : 0.f;
#pragma unroll
        for (int m = 0; m < NL; ++m) Bs[64 * (m + 1) + tid] = tid < p.wo[m + 1] ? p.b[m][tid] : 0.f;
    }
}

// the piece products of one k-step, smallest terms first: six with three pieces, three with two (second x second <= 2^-18 of the term)
template <int NP>
__device__ __forceinline__ f32x16 mfma_pieces(const u32x4 (&a)[3], const u32x4 (&b)[3], f32x16 c) {
    if (NP == 3) return mfma_split6(a, b, c);
    c = mfma_bf16(a[1], b[0], c);
    c = mfma_bf16(a[0], b[1], c);
    return mfma_bf16(a[0], b[0], c);
}

// one layer: acc[io] = b + W h, W as NP planes at Wp
template <int NP = 3>
__device__ __forceinline__ void km_layer_split(const unsigned char* Wp, const float* bias, const f32x16 (&h)[2], int li, int hi, f32x16 (&acc)[2]) {
#pragma unroll
    for (int io = 0; io < 2; ++io)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[io][r] = (bias + 4 * hi)[io * 32 + crow0(r)];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int kt = j >> 1, r0 = 8 * (j & 1);
        unsigned q0[4], q1[4], q2[4];
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) km_split<NP>(h[kt][r0 + 2 * e2], h[kt][r0 + 2 * e2 + 1], q0[e2], q1[e2], q2[e2]);
        const u32x4 b[3] = {u32x4{q0[0], q0[1], q0[2], q0[3]}, u32x4{q1[0], q1[1], q1[2], q1[3]}, u32x4{q2[0], q2[1], q2[2], q2[3]}};
#pragma unroll
        for (int io = 0; io < 2; ++io) {
            const unsigned char* ap = Wp + (io * 32 + li) * KS_LDB + (16 * j + 4 * hi) * 2;
            u32x4 a[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                a[pl] = pl < NP ? join8(*reinterpret_cast<const u32x2*>(ap + pl * KS_PLANE), *reinterpret_cast<const u32x2*>(ap + pl * KS_PLANE + 16))
                                : u32x4{0u, 0u, 0u, 0u};
            acc[io] = mfma_pieces<NP>(a, b, acc[io]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// (two workgroups per CU: 9 planes + the first layer's [64][CM] weights + biases are 80 384 B at c_in <= 4, and 256 registers)
// The body takes its workgroup index and its LDS from the caller: the single-problem kernel and the PAIR kernel below (two chains in one
// launch, the workgroups of the second behind those of the first) share it.
template <int NL, int CM, int NP>
struct KmFwdSmem {
    static constexpr int WP = NL * NP * KS_PLANE, W1 = 64 * CM * 4, BS = 64 * (NL + 1) * 4;
    static constexpr int BYTES = WP + W1 + BS;
};
template <int NL, int CM, int ACT, int NP = 3>
__device__ __forceinline__ void km_fwd_split_body(const KMArgs& p, const int block, const int nblocks, unsigned char* smem) {
    using S = KmFwdSmem<NL, CM, NP>;
    unsigned char* Wp = smem;
    float* W1s = reinterpret_cast<float*>(smem + S::WP);
    float* Bs = reinterpret_cast<float*>(smem + S::WP + S::W1);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, hi = lane >> 5;
    // Workgroup `block` of `nblocks` walks the 128-edge tiles block, block + nblocks, ...: the weights are staged (read, split into planes,
    // written to LDS) ONCE per workgroup.  With a workgroup per tile the staging -- 48 KB of fp32 weights through the L2, 12 288 splits -- was
    // paid for every 128 edges: 3 578 times on the 458 k-edge unions of a vx batch, seven rounds of workgroups on the chip.
    // The edge coordinates are requested before the weights are staged (and a tile ahead after that): their latency hides behind the work.
    int e0 = (block * 4 + wave) * 32;
    float xr[CM];
    km_load_x<CM>(p, min(e0 + li, p.E - 1), xr);
    km_stage_weights_split<NL, CM, NP>(p, Wp, W1s, Bs, tid);
    __syncthreads();
    for (int tile = block; tile < p.ntiles; tile += nblocks) {
        e0 = (tile * 4 + wave) * 32;
        float xn[CM];
        const int en = ((tile + nblocks) * 4 + wave) * 32 + li;
        km_load_x<CM>(p, min(en, p.E - 1), xn);
        if (e0 < p.E) {
            f32x16 z[2], h[2];
            km_layer0<CM>(W1s, Bs, xr, p.cin, hi, z);
#pragma unroll
            for (int m = 0; m < NL; ++m) {
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int t = 0; t < 16; ++t) h[kt][t] = act_f<ACT>(z[kt][t]);
                km_layer_split<NP>(Wp + m * NP * KS_PLANE, Bs + 64 * (m + 1), h, li, hi, z);
            }
            if (e0 + li < p.E) {
                float* dst = p.out + (long)(e0 + li) * p.cout;
#pragma unroll
                for (int io = 0; io < 2; ++io)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (io * 32 + 8 * q + 4 * hi < p.cout)
                            *reinterpret_cast<f32x4*>(dst + io * 32 + 8 * q + 4 * hi) = f32x4{z[io][4 * q], z[io][4 * q + 1], z[io][4 * q + 2], z[io][4 * q + 3]};
            }
        }
#pragma unroll
        for (int c = 0; c < CM; ++c) xr[c] = xn[c];
    }
}
template <int NL, int CM, int ACT, int NP = 3>
__global__ __launch_bounds__(256, 2) void kernel_mlp_fwd_split_kernel(const KMArgs p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[KmFwdSmem<NL, CM, NP>::BYTES];
    km_fwd_split_body<NL, CM, ACT, NP>(p, blockIdx.x, gridDim.x, smem);
}
// TWO chains in one launch (exact products): the kernel MLP of an integral transform (GELU, over the E edge rows: 435 workgroups at the
// bench configuration) and the geometry-embedding chain of the same transform (ReLU, over the Q query rows: 32 .. 128 workgroups) --
// neither depends on the other, and alone the second is a 12-14 us launch of a few dozen workgroups on 256 CUs.  Workgroups 0 .. na - 1
// walk chain A's tiles, the nb behind them chain B's.
template <int NLA, int CMA, int NLB, int CMB>
__global__ __launch_bounds__(256, 2) void kernel_mlp_fwd_pair_kernel(const KMArgs a, const KMArgs b, const int na, const int nb) {
    constexpr int BYTES = KmFwdSmem<NLA, CMA, 3>::BYTES > KmFwdSmem<NLB, CMB, 3>::BYTES ? KmFwdSmem<NLA, CMA, 3>::BYTES : KmFwdSmem<NLB, CMB, 3>::BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[BYTES];
    if ((int)blockIdx.x < na) km_fwd_split_body<NLA, CMA, GAOT_ACT_GELU, 3>(a, blockIdx.x, na, smem);
    else km_fwd_split_body<NLB, CMB, GAOT_ACT_RELU, 3>(b, (int)blockIdx.x - na, nb, smem);
}

template <int NL>
struct KmBwdSmem {
    static constexpr int TILE = 64 * KM_TLD128;
    static constexpr int WP = NL * 3 * KS_PLANE, W1 = 64 * KM_MAXC * 4, BS = 64 * (NL + 1) * 4, GT = TILE * 4, XT = 128 * KM_MAXC * 4;
    static constexpr int BYTES = WP + W1 + BS + 2 * GT + XT;
};
// (body: workgroup `block` of `nblocks` walks the tiles block, block + nblocks, ...; see km_fwd_split_body)
template <int NL, int CM, int ACT>
__device__ __forceinline__ void km_bwd_split_body(const KMArgs& p, const int block, const int nblocks, unsigned char* smem) {
    using S = KmBwdSmem<NL>;
    unsigned char* Wp = smem;
    float* W1s = reinterpret_cast<float*>(smem + S::WP);
    float* Bs = reinterpret_cast<float*>(smem + S::WP + S::W1);
    float* Gt = reinterpret_cast<float*>(smem + S::WP + S::W1 + S::BS);
    float* Ht = Gt + S::TILE;
    float* Xt = Ht + S::TILE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, hi = lane >> 5;
    const int io_w = wave >> 1, kt_w = wave & 1;
    const int cin = p.cin;
    // transposing read: lane t of 16-lane group g addresses row 4 (g >> 1) + (t >> 2), columns 16 (g & 1) + 4 (t & 3) .. + 3 of a
    // [16 rows][32 columns] block and receives rows 4 (g >> 1) .. + 3 of column 16 (g & 1) + t = li: k-slots 0..3 of half-wave hi = g >> 1
    // (k-slots 4..7: the same, eight rows further down)
    const int tr_off = (4 * (lane >> 5) + ((lane & 15) >> 2)) * KS_LDB + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    km_stage_weights_split<NL, CM>(p, Wp, W1s, Bs, tid);
    __syncthreads();

    f32x16 dW[NL];
#pragma unroll
    for (int m = 0; m < NL; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) dW[m][r] = 0.f;
    float db[NL + 1];
    float dw1[CM];
#pragma unroll
    for (int m = 0; m <= NL; ++m) db[m] = 0.f;
#pragma unroll
    for (int c = 0; c < CM; ++c) dw1[c] = 0.f;

    for (int tile = block; tile < p.ntiles; tile += nblocks) {
        asm volatile("" ::: "memory");
        const int e0 = tile * 128 + wave * 32;
        const bool valid = e0 + li < p.E;
        const int e = min(e0 + li, p.E - 1);
        float xr[CM];
        km_load_x<CM>(p, e, xr);
        // upstream gradient: requested now, consumed after the recompute (one wave per SIMD: nothing else hides the HBM latency)
        f32x16 g[2];
        {
            const float* src = p.dk + (long)e * p.cout;
#pragma unroll
            for (int io = 0; io < 2; ++io)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (valid && io * 32 + 8 * q + 4 * hi < p.cout) v = *reinterpret_cast<const f32x4*>(src + io * 32 + 8 * q + 4 * hi);
                    g[io][4 * q] = v[0]; g[io][4 * q + 1] = v[1]; g[io][4 * q + 2] = v[2]; g[io][4 * q + 3] = v[3];
                }
        }
        if (hi == 0) {
#pragma unroll
            for (int c = 0; c < CM; ++c) Xt[(wave * 32 + li) * KM_MAXC + c] = valid ? xr[c] : 0.f;
        }
        // recompute the chain; every hidden activation is evaluated ONCE, together with its derivative: z[m] becomes act'(z[m]),
        // hk[m] = act(z[m]).  The LAST hidden activation is not needed by the recompute: it is evaluated inside its layer's
        // backward, eight values at a time.
        // (KEEP_H: with three MFMA layers the two kept activation fragments push the kernel past 512 registers, and the spills cost
        // more than evaluating those activations a second time)
        constexpr bool KEEP_H = true;
        f32x16 z[NL][2], hk[KEEP_H && NL > 1 ? NL - 1 : 1][2];
        km_layer0<CM>(W1s, Bs, xr, cin, hi, z[0]);
#pragma unroll
        for (int m = 0; m + 1 < NL; ++m) {
            if (KEEP_H) {
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int t = 0; t < 16; ++t) {
                        float hv, dv;
                        act_both<ACT>(z[m][kt][t], hv, dv);
                        hk[m][kt][t] = hv; z[m][kt][t] = dv;
                    }
                km_layer_split(Wp + m * 3 * KS_PLANE, Bs + 64 * (m + 1), hk[m], li, hi, z[m + 1]);
            } else {
                f32x16 h[2];
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int t = 0; t < 16; ++t) h[kt][t] = act_f<ACT>(z[m][kt][t]);
                km_layer_split(Wp + m * 3 * KS_PLANE, Bs + 64 * (m + 1), h, li, hi, z[m + 1]);
            }
        }
        // The bf16 MFMA does not round its accumulator to nearest: every result carries a small bias of ONE sign, whatever the sign
        // of the value (measured: db_1 = sum over edges of G_0 was 2.8e-6 off at 5.6e4 edges and 8.8e-6 at 4e5, growing like sqrt(E),
        // where the fp32 MFMA gives 3e-7).  Odd edges therefore run the input-gradient chain on -G: their bias comes back with the
        // opposite sign and cancels in every sum over edges (the sign is undone in the act' multiply below; G itself goes to the
        // tiles unflipped).
        const unsigned sbit = (unsigned)(li & 1) << 31;
        auto flip = [sbit](float v) { return __uint_as_float(__float_as_uint(v) ^ sbit); };
        float* gcol = Gt + (4 * hi) * KM_TLD128 + wave * 32 + li;       // this lane's edge column, feature row 4 hi
        float* hcol = Ht + (4 * hi) * KM_TLD128 + wave * 32 + li;
#pragma unroll
        for (int m = NL; m >= 1; --m) {
            // dH_{m-1} = W_m^T G_m on the bf16 pipe (W^T fragments by transposing reads, G split out of the registers), G_m and
            // H_{m-1} to the workgroup's fp32 [feature][edge] tiles on the way; then dW_m quadrant on the fp32 MFMA (its operands
            // are summed over ~10^5 edges with cancellation: two-piece operands there cost 7e-6 of relative error, measured)
            const unsigned char* W = Wp + (m - 1) * 3 * KS_PLANE;
            f32x16 dh[2];
#pragma unroll
            for (int io = 0; io < 2; ++io)
#pragma unroll
                for (int r = 0; r < 16; ++r) dh[io][r] = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int kt = j >> 1, r0 = 8 * (j & 1);
                u32x4 a[2][3];
#pragma unroll
                for (int io = 0; io < 2; ++io) {
                    const unsigned char* ap = W + (16 * j) * KS_LDB + (io * 32) * 2 + tr_off;
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) a[io][pl] = join8(lds_tr(ap + pl * KS_PLANE), lds_tr(ap + pl * KS_PLANE + 8 * KS_LDB));
                }
                unsigned q0[4], q1[4], q2[4];
#pragma unroll
                for (int e2 = 0; e2 < 4; ++e2) split3_pair(flip(g[kt][r0 + 2 * e2]), flip(g[kt][r0 + 2 * e2 + 1]), q0[e2], q1[e2], q2[e2]);
                const u32x4 b[3] = {u32x4{q0[0], q0[1], q0[2], q0[3]}, u32x4{q1[0], q1[1], q1[2], q1[3]}, u32x4{q2[0], q2[1], q2[2], q2[3]}};
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    float hv;
                    if (m == NL || !KEEP_H) { float dv; act_both<ACT>(z[m - 1][kt][r0 + u], hv, dv); z[m - 1][kt][r0 + u] = dv; }
                    else hv = hk[m == NL ? 0 : m - 1][kt][r0 + u];
                    gcol[(kt * 32 + crow0(r0 + u)) * KM_TLD128] = g[kt][r0 + u];
                    hcol[(kt * 32 + crow0(r0 + u)) * KM_TLD128] = hv;
                }
#pragma unroll
                for (int io = 0; io < 2; ++io) dh[io] = mfma_split6(a[io], b, dh[io]);
                __builtin_amdgcn_sched_barrier(0);          // one k-step at a time: left alone, the scheduler hoists fragment reads until it spills
            }
            __syncthreads();
            {
                const float* ga = Gt + (io_w * 32 + li) * KM_TLD128 + 4 * hi;
                const float* hb = Ht + (kt_w * 32 + li) * KM_TLD128 + 4 * hi;
                float dsum = 0.f;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const f32x4 av = *reinterpret_cast<const f32x4*>(ga + 8 * q);
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(hb + 8 * q);
                    dsum += (av[0] + av[1]) + (av[2] + av[3]);
#pragma unroll
                    for (int s = 0; s < 4; ++s) dW[m - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[s], dW[m - 1], 0, 0, 0);
                }
                db[m] += dsum;
            }
#pragma unroll
            for (int io = 0; io < 2; ++io)
#pragma unroll
                for (int r = 0; r < 16; ++r) g[io][r] = dh[io][r] * flip(z[m - 1][io][r]);
            __syncthreads();
        }
        // first layer: dW_1 = G_0 x^T, db_1 on the VALU
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int t = 0; t < 16; ++t) gcol[(kt * 32 + crow0(t)) * KM_TLD128] = g[kt][t];
        __syncthreads();
        {
            const float* ga = Gt + (io_w * 32 + li) * KM_TLD128 + kt_w * 64 + 4 * hi;
            float dsum = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(ga + 8 * q);
                dsum += (av[0] + av[1]) + (av[2] + av[3]);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const float* xe = Xt + (kt_w * 64 + 8 * q + 4 * hi + s) * KM_MAXC;
#pragma unroll
                    for (int c4 = 0; c4 < CM / 4; ++c4) {
                        const f32x4 xv = *reinterpret_cast<const f32x4*>(xe + 4 * c4);
                        dw1[4 * c4] += av[s] * xv[0]; dw1[4 * c4 + 1] += av[s] * xv[1]; dw1[4 * c4 + 2] += av[s] * xv[2]; dw1[4 * c4 + 3] += av[s] * xv[3];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            db[0] += dsum;
        }
        __syncthreads();
    }

    float* dst = p.ws + (long)block * p.psize;
#pragma unroll
    for (int m = 0; m < NL; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[m * 4096 + (io_w * 32 + crow(r, hi)) * 64 + kt_w * 32 + li] = dW[m][r];
    float* R = Gt;
    const int off_b = 64 * cin, nsmall = 64 * cin + 64 * (NL + 1);
#pragma unroll
    for (int c = 0; c < CM; ++c) dw1[c] += __shfl_xor(dw1[c], 32, 64);
#pragma unroll
    for (int m = 0; m <= NL; ++m) db[m] += __shfl_xor(db[m], 32, 64);
    for (int pass = 0; pass < 2; ++pass) {
        if (kt_w == pass && hi == 0) {
#pragma unroll
            for (int c = 0; c < CM; ++c)
                if (c < cin) { float* d = R + (io_w * 32 + li) * cin + c; *d = pass == 0 ? dw1[c] : *d + dw1[c]; }
            float* d0 = R + off_b + io_w * 32 + li;
            *d0 = pass == 0 ? db[0] : *d0 + db[0];
            if (pass == 0) {
#pragma unroll
                for (int m = 1; m <= NL; ++m) R[off_b + m * 64 + io_w * 32 + li] = db[m];
            }
        }
        __syncthreads();
    }
    for (int i = tid; i < nsmall; i += 256) dst[NL * 4096 + i] = R[i];
}

template <int NL, int CM, int ACT>
__global__ __launch_bounds__(256, 1) void kernel_mlp_bwd_split_kernel(const KMArgs p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[KmBwdSmem<NL>::BYTES];
    km_bwd_split_body<NL, CM, ACT>(p, blockIdx.x, gridDim.x, smem);
}
// the backward of the two chains of kernel_mlp_fwd_pair_kernel in one launch: one workgroup per CU, so the second chain's few workgroups
// start as the first ones of chain A retire (its 435 tiles over 256 workgroups leave a third of the CUs idle during their second round)
template <int NLA, int CMA, int NLB, int CMB>
__global__ __launch_bounds__(256, 1) void kernel_mlp_bwd_pair_kernel(const KMArgs a, const KMArgs b, const int na, const int nb) {
    constexpr int BYTES = KmBwdSmem<NLA>::BYTES > KmBwdSmem<NLB>::BYTES ? KmBwdSmem<NLA>::BYTES : KmBwdSmem<NLB>::BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[BYTES];
    if ((int)blockIdx.x < na) km_bwd_split_body<NLA, CMA, GAOT_ACT_GELU>(a, blockIdx.x, na, smem);
    else km_bwd_split_body<NLB, CMB, GAOT_ACT_RELU>(b, (int)blockIdx.x - na, nb, smem);
}

// ------------------------------------------------------------------------------------------ backward, two rounded pieces everywhere
// With two pieces per operand the weight planes take a third less LDS, and the weight gradient moves onto the bf16 pipe as well:
// G_m and H_{m-1} go to LDS as their two bf16 pieces, [128 edges][64 features] (a lane's four consecutive features are one 8-byte
// store; G's pieces are the ones the input-gradient chain has just formed), and come back edge-major through the transposing reads;
// three piece products per 16-edge k-step.  db rides on the A fragments (v_dot2c_f32_bf16).  The sign flip of odd edges (see the
// three-piece kernel) covers G AND H here, so their products -- and dW -- are unchanged; db undoes it with (+1, -1) dot factors.
// The bf16 accumulators of dW run over at most KS_FLUSH_TILES tiles before they are added to the workgroup's partial row.
constexpr int KS_TPLANE = 128 * KS_LDB;
constexpr int KS_FLUSH_TILES = 16;
__device__ __forceinline__ float dot2_pm(unsigned w, float acc) {          // acc + lo(w) - hi(w) of a packed bf16 pair
    asm("v_dot2c_f32_bf16 %0, 0xbf803f80, %1" : "+v"(acc) : "v"(w));       // (inline asm: see the note on __builtin_amdgcn_fdot2_f32_bf16 in DESIGN.md 6)
    return acc;
}
template <int NL, int CM, int ACT>
__global__ __launch_bounds__(256, 1) void kernel_mlp_bwd_split2_kernel(const KMArgs p) {
    __shared__ __attribute__((aligned(16))) unsigned char Wp[NL * 2 * KS_PLANE];
    __shared__ __attribute__((aligned(16))) float W1s[64 * KM_MAXC];
    __shared__ float Bs[64 * (NL + 1)];
    __shared__ __attribute__((aligned(16))) unsigned char T[4 * KS_TPLANE];     // G piece 1 | G piece 2 | H piece 1 | H piece 2; first layer: fp32 [64][132]
    __shared__ __attribute__((aligned(16))) float Xt[128 * KM_MAXC];
    static_assert(4 * KS_TPLANE >= 64 * KM_TLD128 * 4, "the first layer's fp32 tile lives in the plane region");
    float* Gt = reinterpret_cast<float*>(T);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, hi = lane >> 5;
    const int io_w = wave >> 1, kt_w = wave & 1;
    const int cin = p.cin;
    const int tr_off = (4 * (lane >> 5) + ((lane & 15) >> 2)) * KS_LDB + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    // edge coordinates run one tile ahead (the first request goes out before the weights are staged): with one wave per SIMD nothing
    // else hides the latency of the load the whole tile starts with
    float xn[CM];
    km_load_x<CM>(p, min((int)blockIdx.x * 128 + wave * 32 + li, p.E - 1), xn);
    km_stage_weights_split<NL, CM, 2>(p, Wp, W1s, Bs, tid);
    __syncthreads();

    f32x16 dW[NL];
#pragma unroll
    for (int m = 0; m < NL; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) dW[m][r] = 0.f;
    float db[NL + 1];
    float dw1[CM];
#pragma unroll
    for (int m = 0; m <= NL; ++m) db[m] = 0.f;
#pragma unroll
    for (int c = 0; c < CM; ++c) dw1[c] = 0.f;
    float* dst = p.ws + (long)blockIdx.x * p.psize;
    float* dquad = dst + (io_w * 32 + 4 * hi) * 64 + kt_w * 32 + li;      // this lane's column of its dW quadrant, row 4 hi (one address register)
    int since_flush = 0;
    bool flushed = false;

    for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        asm volatile("" ::: "memory");
        if (since_flush == KS_FLUSH_TILES) {
            float* dq2 = dquad;
            asm volatile("" : "+v"(dq2));           // keep the 48 store addresses of this cold block out of the loop's registers
#pragma unroll
            for (int m = 0; m < NL; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float* d = dq2 + m * 4096 + crow0(r) * 64;
                    *d = flushed ? *d + dW[m][r] : dW[m][r];
                    dW[m][r] = 0.f;
                }
            flushed = true;
            since_flush = 0;
        }
        ++since_flush;
        const int e0 = tile * 128 + wave * 32;
        const bool valid = e0 + li < p.E;
        const int e = min(e0 + li, p.E - 1);
        float xr[CM];
#pragma unroll
        for (int c = 0; c < CM; ++c) xr[c] = xn[c];
        if (tile + (int)gridDim.x < p.ntiles) km_load_x<CM>(p, min((tile + (int)gridDim.x) * 128 + wave * 32 + li, p.E - 1), xn);
        f32x16 g[2];
        {
            const float* src = p.dk + (long)e * p.cout;
#pragma unroll
            for (int io = 0; io < 2; ++io)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (valid && io * 32 + 8 * q + 4 * hi < p.cout) v = *reinterpret_cast<const f32x4*>(src + io * 32 + 8 * q + 4 * hi);
                    g[io][4 * q] = v[0]; g[io][4 * q + 1] = v[1]; g[io][4 * q + 2] = v[2]; g[io][4 * q + 3] = v[3];
                }
        }
        if (hi == 0) {
#pragma unroll
            for (int c = 0; c < CM; ++c) Xt[(wave * 32 + li) * KM_MAXC + c] = valid ? xr[c] : 0.f;
        }
        f32x16 z[NL][2], hk[NL > 1 ? NL - 1 : 1][2];
        km_layer0<CM>(W1s, Bs, xr, cin, hi, z[0]);
#pragma unroll
        for (int m = 0; m + 1 < NL; ++m) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    float hv, dv;
                    act_both<ACT>(z[m][kt][t], hv, dv);
                    hk[m][kt][t] = hv; z[m][kt][t] = dv;
                }
            km_layer_split<2>(Wp + m * 2 * KS_PLANE, Bs + 64 * (m + 1), hk[m], li, hi, z[m + 1]);
        }
        const unsigned sbit = (unsigned)(li & 1) << 31;
        auto flip = [sbit](float v) { return __uint_as_float(__float_as_uint(v) ^ sbit); };
        unsigned char* trow = T + (wave * 32 + li) * KS_LDB + (4 * hi) * 2;      // this lane's edge row, its feature quad
        float* gcol = Gt + (4 * hi) * KM_TLD128 + wave * 32 + li;
#pragma unroll
        for (int m = NL; m >= 1; --m) {
            const unsigned char* W = Wp + (m - 1) * 2 * KS_PLANE;
            f32x16 dh[2];
#pragma unroll
            for (int io = 0; io < 2; ++io)
#pragma unroll
                for (int r = 0; r < 16; ++r) dh[io][r] = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int kt = j >> 1, r0 = 8 * (j & 1);
                u32x4 a[2][2];
#pragma unroll
                for (int io = 0; io < 2; ++io) {
                    const unsigned char* ap = W + (16 * j) * KS_LDB + (io * 32) * 2 + tr_off;
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) a[io][pl] = join8(lds_tr(ap + pl * KS_PLANE), lds_tr(ap + pl * KS_PLANE + 8 * KS_LDB));
                }
                unsigned g1[4], g2[4], h1[4], h2[4];
                float hv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (m == NL) { float dv; act_both<ACT>(z[m - 1][kt][r0 + u], hv[u], dv); z[m - 1][kt][r0 + u] = dv; }
                    else hv[u] = hk[m == NL ? 0 : m - 1][kt][r0 + u];
                }
#pragma unroll
                for (int e2 = 0; e2 < 4; ++e2) {
                    split2_pair(flip(g[kt][r0 + 2 * e2]), flip(g[kt][r0 + 2 * e2 + 1]), g1[e2], g2[e2]);
                    split2_pair(flip(hv[2 * e2]), flip(hv[2 * e2 + 1]), h1[e2], h2[e2]);
                }
                // registers r0 .. r0 + 3 are features 16 j + 4 hi + 0..3, r0 + 4 .. r0 + 7 the same, eight features up
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    unsigned char* d = trow + (16 * j + 8 * s2) * 2;
                    *reinterpret_cast<u32x2*>(d) = u32x2{g1[2 * s2], g1[2 * s2 + 1]};
                    *reinterpret_cast<u32x2*>(d + KS_TPLANE) = u32x2{g2[2 * s2], g2[2 * s2 + 1]};
                    *reinterpret_cast<u32x2*>(d + 2 * KS_TPLANE) = u32x2{h1[2 * s2], h1[2 * s2 + 1]};
                    *reinterpret_cast<u32x2*>(d + 3 * KS_TPLANE) = u32x2{h2[2 * s2], h2[2 * s2 + 1]};
                }
                const u32x4 b0 = {g1[0], g1[1], g1[2], g1[3]}, b1 = {g2[0], g2[1], g2[2], g2[3]};
#pragma unroll
                for (int io = 0; io < 2; ++io) {
                    dh[io] = mfma_bf16(a[io][1], b0, dh[io]);
                    dh[io] = mfma_bf16(a[io][0], b1, dh[io]);
                    dh[io] = mfma_bf16(a[io][0], b0, dh[io]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
            // dW_m quadrant (io_w, kt_w) += G_m[io_w features] H_{m-1}[kt_w features]^T over the tile's 128 edges
            {
                const unsigned char* ga = T + (io_w * 32) * 2 + tr_off;
                const unsigned char* hb = T + 2 * KS_TPLANE + (kt_w * 32) * 2 + tr_off;
                float dsum = 0.f;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    u32x4 a[2], b[2];
#pragma unroll
                    for (int pc = 0; pc < 2; ++pc) {
                        a[pc] = join8(lds_tr(ga + pc * KS_TPLANE + (16 * ks) * KS_LDB), lds_tr(ga + pc * KS_TPLANE + (16 * ks + 8) * KS_LDB));
                        b[pc] = join8(lds_tr(hb + pc * KS_TPLANE + (16 * ks) * KS_LDB), lds_tr(hb + pc * KS_TPLANE + (16 * ks + 8) * KS_LDB));
                    }
                    dW[m - 1] = mfma_bf16(a[1], b[0], dW[m - 1]);
                    dW[m - 1] = mfma_bf16(a[0], b[1], dW[m - 1]);
                    dW[m - 1] = mfma_bf16(a[0], b[0], dW[m - 1]);
                    if (kt_w == 0) {
#pragma unroll
                        for (int pc = 1; pc >= 0; --pc)
#pragma unroll
                            for (int u = 0; u < 4; ++u) dsum = dot2_pm(a[pc][u], dsum);
                    }
                }
                db[m] += dsum;
            }
#pragma unroll
            for (int io = 0; io < 2; ++io)
#pragma unroll
                for (int r = 0; r < 16; ++r) g[io][r] = dh[io][r] * flip(z[m - 1][io][r]);
            __syncthreads();
        }
        // first layer: dW_1 = G_0 x^T, db_1 on the VALU (fp32 tile in the plane region)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int t = 0; t < 16; ++t) gcol[(kt * 32 + crow0(t)) * KM_TLD128] = g[kt][t];
        __syncthreads();
        {
            const float* ga = Gt + (io_w * 32 + li) * KM_TLD128 + kt_w * 64 + 4 * hi;
            float dsum = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(ga + 8 * q);
                dsum += (av[0] + av[1]) + (av[2] + av[3]);
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) {
                    const float* xe = Xt + (kt_w * 64 + 8 * q + 4 * hi + s2) * KM_MAXC;
#pragma unroll
                    for (int c4 = 0; c4 < CM / 4; ++c4) {
                        const f32x4 xv = *reinterpret_cast<const f32x4*>(xe + 4 * c4);
                        dw1[4 * c4] += av[s2] * xv[0]; dw1[4 * c4 + 1] += av[s2] * xv[1]; dw1[4 * c4 + 2] += av[s2] * xv[2]; dw1[4 * c4 + 3] += av[s2] * xv[3];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            db[0] += dsum;
        }
        __syncthreads();
    }

    asm volatile("" : "+v"(dquad));
#pragma unroll
    for (int m = 0; m < NL; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float* d = dquad + m * 4096 + crow0(r) * 64;
            *d = flushed ? *d + dW[m][r] : dW[m][r];
        }
    float* R = Gt;
    const int off_b = 64 * cin, nsmall = 64 * cin + 64 * (NL + 1);
#pragma unroll
    for (int c = 0; c < CM; ++c) dw1[c] += __shfl_xor(dw1[c], 32, 64);
#pragma unroll
    for (int m = 0; m <= NL; ++m) db[m] += __shfl_xor(db[m], 32, 64);
    for (int pass = 0; pass < 2; ++pass) {
        if (kt_w == pass && hi == 0) {
#pragma unroll
            for (int c = 0; c < CM; ++c)
                if (c < cin) { float* d = R + (io_w * 32 + li) * cin + c; *d = pass == 0 ? dw1[c] : *d + dw1[c]; }
            float* d0 = R + off_b + io_w * 32 + li;
            *d0 = pass == 0 ? db[0] : *d0 + db[0];
            if (pass == 0) {
#pragma unroll
                for (int m = 1; m <= NL; ++m) R[off_b + m * 64 + io_w * 32 + li] = db[m];
            }
        }
        __syncthreads();
    }
    for (int i = tid; i < nsmall; i += 256) dst[NL * 4096 + i] = R[i];
}

static int km_check(const float* x, int E, int cin, int n_layers, const float* const* w, const float* const* b) {
    GAOT_REQUIRE(x && E > 0 && cin >= 1 && cin <= KM_MAXC, "kernel_mlp: need x, E > 0 and 1 <= c_in <= %d (got %d)", KM_MAXC, cin);
    GAOT_REQUIRE(n_layers >= 2 && n_layers <= 4, "kernel_mlp: 2..4 layers (got %d)", n_layers);
    for (int i = 0; i < n_layers; ++i) GAOT_REQUIRE(w[i] && b[i] && aligned16(w[i]), "kernel_mlp: weights / biases must be non-null, weights 16-byte aligned");
    if (cin == 4) GAOT_REQUIRE(aligned16(x) && aligned16(w[0]), "kernel_mlp: x and W_1 must be 16-byte aligned for c_in = 4");
    return GAOT_OK;
}

static int g_km_abl = 0;
constexpr int KM_FWD_WGS = 512;     // resident workgroups of the split forward kernels: two per CU on 256 CUs
static int g_km_split = 1;      // 1: bf16-split kernels (default), 0: the fp32-MFMA kernels (gaot_debug_set_kernel_mlp_split)
static void km_fill(KMArgs& a, const float* x, int E, int cin, int n_layers, const float* const* w, const float* const* b, const int* widths,
                    const int* ldw = nullptr) {
    a.x = x; a.cin = cin; a.E = E; a.w1 = w[0]; a.b1 = b[0]; a.cout = widths[n_layers - 1];
    for (int i = 0; i < 4; ++i) a.wo[i] = i < n_layers ? widths[i] : 64;
    for (int i = 0; i < 4; ++i) a.ldw[i] = (ldw && i < n_layers && ldw[i] > 0) ? ldw[i] : (i == 0 ? cin : a.wo[i - 1]);
    for (int m = 0; m < 3; ++m) { a.w[m] = m + 1 < n_layers ? w[m + 1] : nullptr; a.b[m] = m + 1 < n_layers ? b[m + 1] : nullptr; }
    a.ntiles = cdiv(E, 128);
    a.abl = g_km_abl;
    a.psize = (n_layers - 1) * 4096 + 64 * cin + 64 * n_layers;
}

}  // namespace gaot

using namespace gaot;

extern "C" int gaot_debug_set_kernel_mlp_ablate(int bits) { const int old = g_km_abl; g_km_abl = bits; return old; }
extern "C" int gaot_debug_set_kernel_mlp_split(int on) { const int old = g_km_split; g_km_split = on ? 1 : 0; return old; }

static int km_widths_ok(const int32_t* widths, int n_layers) {
    GAOT_REQUIRE(widths, "kernel_mlp: widths must be non-null");
    for (int i = 0; i < n_layers; ++i)
        GAOT_REQUIRE(widths[i] >= 4 && widths[i] <= 64 && widths[i] % 4 == 0, "kernel_mlp: layer width %d must be a multiple of 4 in 4..64", widths[i]);
    return GAOT_OK;
}
static const int32_t KM_W64[4] = {64, 64, 64, 64};

extern "C" int gaot_kernel_mlp_fwd_w(const float* x, int32_t E, int32_t cin, int32_t n_layers, const float* const* w,
                                     const float* const* b, int32_t act, const int32_t* widths, const int32_t* ldw, int32_t pieces, float* out,
                                     gaot_stream_t stream);
extern "C" int gaot_kernel_mlp_fwd(const float* x, int32_t E, int32_t cin, int32_t n_layers, const float* const* w,
                                   const float* const* b, int32_t act, float* out, gaot_stream_t stream) {
    return gaot_kernel_mlp_fwd_w(x, E, cin, n_layers, w, b, act, KM_W64, nullptr, 0, out, stream);
}

static int km_ldw_ok(const int32_t* ldw, const int32_t* widths, int cin, int n_layers) {
    if (!ldw) return GAOT_OK;
    for (int i = 0; i < n_layers; ++i) {
        const int in_w = i == 0 ? cin : widths[i - 1];
        GAOT_REQUIRE(ldw[i] == 0 || (ldw[i] >= in_w && (i == 0 || ldw[i] % 4 == 0)), "kernel_mlp: ldw[%d] = %d must be 0 (dense) or >= %d%s", i, ldw[i], in_w,
                     i ? " and a multiple of 4" : "");
    }
    return GAOT_OK;
}

extern "C" int gaot_kernel_mlp_fwd_w(const float* x, int32_t E, int32_t cin, int32_t n_layers, const float* const* w,
                                     const float* const* b, int32_t act, const int32_t* widths, const int32_t* ldw, int32_t pieces, float* out,
                                     gaot_stream_t stream) {
    GAOT_REQUIRE(pieces == 0 || pieces == 2 || pieces == 3, "kernel_mlp: pieces must be 0 / 3 (exact) or 2 (two rounded pieces), got %d", pieces);
    if (int rc = km_check(x, E, cin, n_layers, w, b)) return rc;
    if (int rc = km_widths_ok(widths, n_layers)) return rc;
    if (int rc = km_ldw_ok(ldw, widths, cin, n_layers)) return rc;
    GAOT_REQUIRE(act == GAOT_ACT_GELU || act == GAOT_ACT_RELU, "kernel_mlp: hidden activation must be GAOT_ACT_GELU or GAOT_ACT_RELU (got %d)", act);
    GAOT_REQUIRE(out && aligned16(out), "kernel_mlp_fwd: out must be non-null and 16-byte aligned");
    KMArgs a{}; km_fill(a, x, E, cin, n_layers, w, b, widths, ldw); a.out = out;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    // split kernels: persistent workgroups, two per CU (the weights are staged once per workgroup); the fp32-MFMA kernel: one per tile
    const dim3 grid((g_km_split && !a.abl && a.ntiles > KM_FWD_WGS) ? KM_FWD_WGS : a.ntiles), block(256);
#define KM_FWD3(K, NL, ...) do { if (cin == 4) hipLaunchKernelGGL((K<NL, 4, __VA_ARGS__>), grid, block, 0, st, a); \
                                 else if (cin <= 8) hipLaunchKernelGGL((K<NL, 8, __VA_ARGS__>), grid, block, 0, st, a); \
                                 else hipLaunchKernelGGL((K<NL, KM_MAXC, __VA_ARGS__>), grid, block, 0, st, a); } while (0)
#define KM_FWD2(NL, A) do { if (g_km_split && !a.abl) { if (pieces == 2) KM_FWD3(kernel_mlp_fwd_split_kernel, NL, A, 2); \
                                                        else KM_FWD3(kernel_mlp_fwd_split_kernel, NL, A, 3); } \
                            else KM_FWD3(kernel_mlp_fwd_kernel, NL, A); } while (0)
#define KM_FWD(NL) do { if (act == GAOT_ACT_RELU) KM_FWD2(NL, GAOT_ACT_RELU); else KM_FWD2(NL, GAOT_ACT_GELU); } while (0)
    if (n_layers == 2) KM_FWD(1); else if (n_layers == 3) KM_FWD(2); else KM_FWD(3);
#undef KM_FWD
#undef KM_FWD2
#undef KM_FWD3
    GAOT_CHECK_LAUNCH("gaot_kernel_mlp_fwd");
    return GAOT_OK;
}

extern "C" int64_t gaot_kernel_mlp_bwd_workspace(int32_t E, int32_t cin, int32_t n_layers) {
    const int64_t psize = (int64_t)(n_layers - 1) * 4096 + 64 * cin + 64 * n_layers;
    int grid = cdiv(E, 128); if (grid > 256) grid = 256;
    return (int64_t)grid * psize;
}

extern "C" int gaot_kernel_mlp_bwd_w(const float* x, int32_t E, int32_t cin, int32_t n_layers, const float* const* w,
                                     const float* const* b, int32_t act, const int32_t* widths, const int32_t* ldw, int32_t pieces, const float* dk,
                                     float* grads, float* workspace, gaot_stream_t stream);
extern "C" int gaot_kernel_mlp_bwd(const float* x, int32_t E, int32_t cin, int32_t n_layers, const float* const* w,
                                   const float* const* b, int32_t act, const float* dk, float* grads, float* workspace,
                                   gaot_stream_t stream) {
    return gaot_kernel_mlp_bwd_w(x, E, cin, n_layers, w, b, act, KM_W64, nullptr, 0, dk, grads, workspace, stream);
}

extern "C" int gaot_kernel_mlp_bwd_w(const float* x, int32_t E, int32_t cin, int32_t n_layers, const float* const* w,
                                     const float* const* b, int32_t act, const int32_t* widths, const int32_t* ldw, int32_t pieces, const float* dk,
                                     float* grads, float* workspace, gaot_stream_t stream) {
    GAOT_REQUIRE(pieces == 0 || pieces == 2 || pieces == 3, "kernel_mlp: pieces must be 0 / 3 (exact) or 2 (two rounded pieces), got %d", pieces);
    if (int rc = km_check(x, E, cin, n_layers, w, b)) return rc;
    if (int rc = km_widths_ok(widths, n_layers)) return rc;
    if (int rc = km_ldw_ok(ldw, widths, cin, n_layers)) return rc;
    GAOT_REQUIRE(act == GAOT_ACT_GELU || act == GAOT_ACT_RELU, "kernel_mlp: hidden activation must be GAOT_ACT_GELU or GAOT_ACT_RELU (got %d)", act);
    GAOT_REQUIRE(dk && grads && workspace && aligned16(dk), "kernel_mlp_bwd: dk (16-byte aligned), grads, workspace must be non-null");
    KMArgs a{}; km_fill(a, x, E, cin, n_layers, w, b, widths, ldw); a.dk = dk; a.ws = workspace;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int grid = a.ntiles > 256 ? 256 : a.ntiles;
#define KM_BWD3(K, NL, A) do { if (cin == 4) hipLaunchKernelGGL((K<NL, 4, A>), dim3(grid), dim3(256), 0, st, a); \
                               else if (cin <= 8) hipLaunchKernelGGL((K<NL, 8, A>), dim3(grid), dim3(256), 0, st, a); \
                               else hipLaunchKernelGGL((K<NL, KM_MAXC, A>), dim3(grid), dim3(256), 0, st, a); } while (0)
#define KM_BWD2(NL, A) do { if (g_km_split && pieces == 2) KM_BWD3(kernel_mlp_bwd_split2_kernel, NL, A); \
                            else if (g_km_split) KM_BWD3(kernel_mlp_bwd_split_kernel, NL, A); else KM_BWD3(kernel_mlp_bwd_kernel, NL, A); } while (0)
#define KM_BWD(NL) do { if (act == GAOT_ACT_RELU) KM_BWD2(NL, GAOT_ACT_RELU); else KM_BWD2(NL, GAOT_ACT_GELU); } while (0)
    if (n_layers == 2) KM_BWD(1); else if (n_layers == 3) KM_BWD(2); else KM_BWD(3);
#undef KM_BWD
#undef KM_BWD2
#undef KM_BWD3
    GAOT_CHECK_LAUNCH("gaot_kernel_mlp_bwd");
    // fixed-order sum of the per-workgroup partial rows: the short-matrix column sum of pointwise.hip (<= 256 rows).
    // grads == workspace: the caller sums the rows itself (gaot_colsum_grouped over parameter-sized column blocks, at the end of
    // the backward pass): rows = gaot_kernel_mlp_bwd_rows(E), row stride = the parameter block size.
    if (grads != workspace)
        if (int rc = gaot_colsum(workspace, a.psize, grid, a.psize, grads, workspace, stream)) return rc;
    return GAOT_OK;
}

extern "C" int32_t gaot_kernel_mlp_bwd_rows(int32_t E) { int grid = cdiv(E, 128); return grid > 256 ? 256 : grid; }

// ---- two chains, one launch (gaot_kernel_mlp_fwd_pair / _bwd_pair): chain A = a GELU chain of 4 layers (the kernel MLP of the integral
// transform), chain B = a ReLU chain of 3 layers (the geometry-embedding chain incl. its half of the recovery block), exact products.
// Any other pair -- or the fp32-MFMA / two-piece modes -- runs as the two single launches.
static bool km_pair_ok(const gaot_kmlp_desc* a, const gaot_kmlp_desc* b) {
    return g_km_split && !g_km_abl && a->n_layers == 4 && b->n_layers == 3 && a->act == GAOT_ACT_GELU && b->act == GAOT_ACT_RELU &&
           (a->pieces == 0 || a->pieces == 3) && (b->pieces == 0 || b->pieces == 3) && a->cin <= 8 && b->cin > 4 && a->E > 0 && b->E > 0;
}
static int km_desc_fill(KMArgs& k, const gaot_kmlp_desc* d, bool bwd) {
    if (int rc = km_check(d->x, d->E, d->cin, d->n_layers, d->w, d->b)) return rc;
    if (int rc = km_widths_ok(d->widths, d->n_layers)) return rc;
    if (int rc = km_ldw_ok(d->ldw, d->widths, d->cin, d->n_layers)) return rc;
    km_fill(k, d->x, d->E, d->cin, d->n_layers, d->w, d->b, d->widths, d->ldw);
    if (!bwd) { GAOT_REQUIRE(d->out && aligned16(d->out), "kernel_mlp_fwd_pair: out must be non-null and 16-byte aligned"); k.out = d->out; }
    else { GAOT_REQUIRE(d->dk && d->grads && d->workspace && aligned16(d->dk), "kernel_mlp_bwd_pair: dk (16-byte aligned), grads, workspace must be non-null");
           k.dk = d->dk; k.ws = d->workspace; }
    return GAOT_OK;
}

extern "C" int gaot_kernel_mlp_fwd_pair(const gaot_kmlp_desc* a, const gaot_kmlp_desc* b, gaot_stream_t stream) {
    GAOT_REQUIRE(a && b, "kernel_mlp_fwd_pair: null descriptor");
    if (!km_pair_ok(a, b)) {
        if (int rc = gaot_kernel_mlp_fwd_w(a->x, a->E, a->cin, a->n_layers, a->w, a->b, a->act, a->widths, a->ldw, a->pieces, a->out, stream)) return rc;
        return gaot_kernel_mlp_fwd_w(b->x, b->E, b->cin, b->n_layers, b->w, b->b, b->act, b->widths, b->ldw, b->pieces, b->out, stream);
    }
    KMArgs ka{}, kb{};
    if (int rc = km_desc_fill(ka, a, false)) return rc;
    if (int rc = km_desc_fill(kb, b, false)) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    // one resident round (two workgroups per CU) shared in proportion to the chains' tiles; fewer tiles than that: a workgroup per tile
    int na = ka.ntiles, nb = kb.ntiles;
    if (na + nb > KM_FWD_WGS) {
        na = (int)((long)KM_FWD_WGS * ka.ntiles / (ka.ntiles + kb.ntiles));
        na = na < 1 ? 1 : (na > KM_FWD_WGS - 1 ? KM_FWD_WGS - 1 : na);
        nb = KM_FWD_WGS - na;
        if (na > ka.ntiles) na = ka.ntiles;
        if (nb > kb.ntiles) nb = kb.ntiles;
    }
    const dim3 grid(na + nb), block(256);
#define KM_PAIR(CA, CB) hipLaunchKernelGGL((kernel_mlp_fwd_pair_kernel<3, CA, 2, CB>), grid, block, 0, st, ka, kb, na, nb)
    if (a->cin == 4) { if (b->cin <= 8) KM_PAIR(4, 8); else KM_PAIR(4, KM_MAXC); }
    else             { if (b->cin <= 8) KM_PAIR(8, 8); else KM_PAIR(8, KM_MAXC); }
#undef KM_PAIR
    GAOT_CHECK_LAUNCH("gaot_kernel_mlp_fwd_pair");
    return GAOT_OK;
}

extern "C" int gaot_kernel_mlp_bwd_pair(const gaot_kmlp_desc* a, const gaot_kmlp_desc* b, gaot_stream_t stream) {
    GAOT_REQUIRE(a && b, "kernel_mlp_bwd_pair: null descriptor");
    if (!km_pair_ok(a, b)) {
        if (int rc = gaot_kernel_mlp_bwd_w(a->x, a->E, a->cin, a->n_layers, a->w, a->b, a->act, a->widths, a->ldw, a->pieces, a->dk, a->grads, a->workspace, stream)) return rc;
        return gaot_kernel_mlp_bwd_w(b->x, b->E, b->cin, b->n_layers, b->w, b->b, b->act, b->widths, b->ldw, b->pieces, b->dk, b->grads, b->workspace, stream);
    }
    KMArgs ka{}, kb{};
    if (int rc = km_desc_fill(ka, a, true)) return rc;
    if (int rc = km_desc_fill(kb, b, true)) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int na = ka.ntiles > 256 ? 256 : ka.ntiles, nb = kb.ntiles > 256 ? 256 : kb.ntiles;
#define KM_PAIR(CA, CB) hipLaunchKernelGGL((kernel_mlp_bwd_pair_kernel<3, CA, 2, CB>), dim3(na + nb), dim3(256), 0, st, ka, kb, na, nb)
    if (a->cin == 4) { if (b->cin <= 8) KM_PAIR(4, 8); else KM_PAIR(4, KM_MAXC); }
    else             { if (b->cin <= 8) KM_PAIR(8, 8); else KM_PAIR(8, KM_MAXC); }
#undef KM_PAIR
    GAOT_CHECK_LAUNCH("gaot_kernel_mlp_bwd_pair");
    // the per-workgroup partial rows: summed here unless the caller does it (grads == workspace, see gaot_kernel_mlp_bwd_w)
    if (a->grads != a->workspace) if (int rc = gaot_colsum(a->workspace, ka.psize, na, ka.psize, a->grads, a->workspace, stream)) return rc;
    if (b->grads != b->workspace) if (int rc = gaot_colsum(b->workspace, kb.psize, nb, kb.psize, b->grads, b->workspace, stream)) return rc;
    return GAOT_OK;
}
