// Skinny products that do not belong on 32x32 MFMA tiles (they are HBM-bound, a few flops per byte):
//   A) tiny reduction  K <= 16        : lifting (K = in_channels), first kernel-MLP layer (K = 2d), geoembed (K = 3+2d)
//   B) tiny output     N <= 4         : output projection (N = out_channels) and its relatives
//   C) tiny weight gradient           : out[M,N] = sum_r A[r,m] B[r,n] with min(M,N) <= 8 over a LONG r (tens of
//                                       thousands of rows) -- a handful of scaled column sums
// All three keep the GEMM descriptor semantics (same epilogue) so callers never see the difference.
#include "gemm_common.h"

namespace gaot {

// ---- A: thread per output element, lanes along n (coalesced stores; A row is a broadcast load)
template <bool BKM>
__global__ __launch_bounds__(256) void skinny_k_kernel(const GemmArgs p) {
    const long total = (long)p.M * p.N;
    for (long gid = (long)blockIdx.x * blockDim.x + threadIdx.x; gid < total; gid += (long)gridDim.x * blockDim.x) {
        const int m = (int)(gid / p.N), n = (int)(gid - (long)m * p.N);
        const float* a = p.A + (long)m * p.lda;
        float acc = 0.f;
        if (BKM) { const float* b = p.B + (long)n * p.ldb; for (int k = 0; k < p.K; ++k) acc = fmaf(a[k], b[k], acc); }
        else     { for (int k = 0; k < p.K; ++k) acc = fmaf(a[k], p.B[(long)k * p.ldb + n], acc); }
        epilogue_store(p, m, n, acc);
    }
}

// ---- B: LPR lanes per output row, each lane strides over k, shuffle-reduce, lane 0 of the group finishes
template <bool BKM>
__global__ __launch_bounds__(256) void skinny_n_kernel(const GemmArgs p, int lpr) {
    const int rows_per_block = 256 / lpr;
    const int sub = threadIdx.x % lpr;
    const int m = blockIdx.x * rows_per_block + threadIdx.x / lpr;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (m < p.M) {
        const float* a = p.A + (long)m * p.lda;
        for (int k = sub; k < p.K; k += lpr) {
            const float av = a[k];
#pragma unroll
            for (int n = 0; n < 4; ++n)
                if (n < p.N) acc[n] = fmaf(av, BKM ? p.B[(long)n * p.ldb + k] : p.B[(long)k * p.ldb + n], acc[n]);
        }
    }
    for (int off = lpr >> 1; off > 0; off >>= 1)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[n] += __shfl_xor(acc[n], off, 64);
    if (m < p.M && sub == 0)
#pragma unroll
        for (int n = 0; n < 4; ++n)
            if (n < p.N) epilogue_store(p, m, n, acc[n]);
}

// ---- C: out[m,n] = sum_r A[r*lda + m] * B[r*ldb + n]; lanes along the WIDE output dim, J <= JM (8 or 16) accumulators along the
// narrow one; grid (wide tiles of 64, row chunks); 4 waves interleave rows; partials -> ws[chunk][M*N]
template <bool WIDE_IS_M, int JM = 8>
__global__ __launch_bounds__(256) void skinny_tn_partial_kernel(const GemmArgs p, int chunks) {
    __shared__ float red[4][JM][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int W = WIDE_IS_M ? p.M : p.N, J = WIDE_IS_M ? p.N : p.M;
    const int w = blockIdx.x * 64 + lane;
    const int rows = (p.K + chunks - 1) / chunks;
    const int r0 = blockIdx.y * rows, r1 = min(p.K, r0 + rows);
    const float* wide = WIDE_IS_M ? p.A : p.B;
    const float* nar = WIDE_IS_M ? p.B : p.A;
    const long ldw = WIDE_IS_M ? p.lda : p.ldb, ldn = WIDE_IS_M ? p.ldb : p.lda;
    float acc[JM], cs[JM];     // cs: fused column sum of the A operand (WIDE_IS_M: cs[0] per lane; else per narrow index)
#pragma unroll
    for (int j = 0; j < JM; ++j) { acc[j] = 0.f; cs[j] = 0.f; }
    if (w < W) {
        int r = r0 + wave;
        for (; r + 12 < r1; r += 16) {             // four rows in flight per wave (HBM latency, not bandwidth, limits here)
            float v[4], nn[4][JM];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[u] = wide[(long)(r + 4 * u) * ldw + w];
#pragma unroll
                for (int j = 0; j < JM; ++j) nn[u][j] = (j < J) ? nar[(long)(r + 4 * u) * ldn + j] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (WIDE_IS_M) cs[0] += v[u];
#pragma unroll
                for (int j = 0; j < JM; ++j) {
                    acc[j] = fmaf(v[u], nn[u][j], acc[j]);
                    if (!WIDE_IS_M) cs[j] += nn[u][j];
                }
            }
        }
        for (; r < r1; r += 4) {
            const float v0 = wide[(long)r * ldw + w];
            if (WIDE_IS_M) cs[0] += v0;
#pragma unroll
            for (int j = 0; j < JM; ++j)
                if (j < J) {
                    const float n0 = nar[(long)r * ldn + j];
                    acc[j] = fmaf(v0, n0, acc[j]);
                    if (!WIDE_IS_M) cs[j] += n0;
                }
        }
    }
#pragma unroll
    for (int j = 0; j < JM; ++j) red[wave][j][lane] = acc[j];
    __syncthreads();
    if (wave == 0 && w < W)
        for (int j = 0; j < J; ++j) {
            const float v = red[0][j][lane] + red[1][j][lane] + red[2][j][lane] + red[3][j][lane];
            const long o = WIDE_IS_M ? (long)w * p.N + j : (long)j * p.N + w;      // [m][n] within the chunk slab
            p.ws[(long)blockIdx.y * p.M * p.N + o] = v;
        }
    if (p.colsum) {             // uniform over the block
        __syncthreads();
#pragma unroll
        for (int j = 0; j < JM; ++j) red[wave][j][lane] = cs[j];
        __syncthreads();
        float* slab = p.ws + (long)chunks * p.M * p.N + (long)blockIdx.y * p.M;
        if (WIDE_IS_M) {
            if (wave == 0 && w < W) slab[w] = red[0][0][lane] + red[1][0][lane] + red[2][0][lane] + red[3][0][lane];
        } else if (blockIdx.x == 0 && threadIdx.x < J) {     // every lane of a wave holds the same narrow sums: take lane 0
            const int j = threadIdx.x;
            slab[j] = red[0][j][0] + red[1][j][0] + red[2][j][0] + red[3][j][0];
        }
    }
}
__global__ __launch_bounds__(256) void skinny_tn_final_kernel(const GemmArgs p, int chunks) {
    __shared__ float red[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long total = (long)p.M * p.N;
    const long o = (long)blockIdx.x * 64 + lane;
    float s = 0.f;
    if (o < total) {
        int c = wave;
        for (; c + 60 < chunks; c += 64) {            // 16 loads in flight per lane
            float t[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) t[u] = p.ws[(long)(c + 4 * u) * total + o];
#pragma unroll
            for (int u = 0; u < 16; ++u) s += t[u];
        }
        for (; c < chunks; c += 4) s += p.ws[(long)c * total + o];
    }
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && o < total) epilogue_store(p, (int)(o / p.N), (int)(o % p.N), red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]);
    if (p.colsum && blockIdx.x == 0) {          // [chunks][M] slab: waves split the chunks, lanes the rows
        const float* slab = p.ws + (long)chunks * total;
        for (int mb = 0; mb < p.M; mb += 64) {
            const int m = mb + lane;
            float v = 0.f;
            if (m < p.M) {
                int c = wave;
                for (; c + 60 < chunks; c += 64) {
                    float t[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) t[u] = slab[(long)(c + 4 * u) * p.M + m];
#pragma unroll
                    for (int u = 0; u < 16; ++u) v += t[u];
                }
                for (; c < chunks; c += 4) v += slab[(long)c * p.M + m];
            }
            __syncthreads();
            red[wave][lane] = v;
            __syncthreads();
            if (wave == 0 && m < p.M) p.colsum[m] = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
        }
    }
}

// the same decision without a launch (gaot_gemm_path)
bool skinny_would(const GemmArgs& a, bool ak, bool bk) {
    if (a.A2 != nullptr) return false;
    if (!ak && !bk && a.split_k > 1 && a.ws != nullptr && (a.N <= 16 || a.M <= 16) && (long)a.M * a.N <= 4096 && a.K >= 1024) return true;
    if (!ak || a.split_k > 1 || a.colsum != nullptr) return false;
    return a.K <= 16 || a.N <= 4;
}

bool launch_skinny(const GemmArgs& a, bool ak, bool bk, hipStream_t st) {
    if (a.A2 != nullptr) return false;
    // C: transposed-A weight gradient with a tiny side and a long reduction (needs the split-K workspace)
    if (!ak && !bk && a.split_k > 1 && a.ws != nullptr && (a.N <= 16 || a.M <= 16) && (long)a.M * a.N <= 4096 && a.K >= 1024) {
        const int chunks = a.split_k;
        // narrow side up to 8 (2-D geometry statistics: 7 columns) or up to 16 (3-D: 9 columns, 3-channel inputs of wide layers)
        if (a.N <= 8)       hipLaunchKernelGGL((skinny_tn_partial_kernel<true, 8>), dim3(cdiv(a.M, 64), chunks), dim3(256), 0, st, a, chunks);
        else if (a.M <= 8)  hipLaunchKernelGGL((skinny_tn_partial_kernel<false, 8>), dim3(cdiv(a.N, 64), chunks), dim3(256), 0, st, a, chunks);
        else if (a.N <= 16) hipLaunchKernelGGL((skinny_tn_partial_kernel<true, 16>), dim3(cdiv(a.M, 64), chunks), dim3(256), 0, st, a, chunks);
        else                hipLaunchKernelGGL((skinny_tn_partial_kernel<false, 16>), dim3(cdiv(a.N, 64), chunks), dim3(256), 0, st, a, chunks);
        hipLaunchKernelGGL(skinny_tn_final_kernel, dim3(cdiv((long)a.M * a.N, 64)), dim3(256), 0, st, a, chunks);
        return true;
    }
    if (!ak || a.split_k > 1 || a.colsum != nullptr) return false;
    if (a.K <= 16) {                                   // A
        long nb = ((long)a.M * a.N + 255) / 256;
        if (nb > 16384) nb = 16384;
        if (bk) hipLaunchKernelGGL(skinny_k_kernel<true>, dim3((unsigned)nb), dim3(256), 0, st, a);
        else    hipLaunchKernelGGL(skinny_k_kernel<false>, dim3((unsigned)nb), dim3(256), 0, st, a);
        return true;
    }
    if (a.N <= 4) {                                    // B
        int lpr = 1;
        while (lpr < 64 && lpr * 4 < a.K) lpr <<= 1;   // ~4 k per lane, at most one wave per row
        const int rpb = 256 / lpr;
        if (bk) hipLaunchKernelGGL(skinny_n_kernel<true>, dim3(cdiv(a.M, rpb)), dim3(256), 0, st, a, lpr);
        else    hipLaunchKernelGGL(skinny_n_kernel<false>, dim3(cdiv(a.M, rpb)), dim3(256), 0, st, a, lpr);
        return true;
    }
    return false;
}

}  // namespace gaot
