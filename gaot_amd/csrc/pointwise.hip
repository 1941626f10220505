// HBM-bound row/elementwise kernels of the processor: RMSNorm fwd/bwd, SwiGLU gate fwd/bwd, column and
// batch reductions (bias gradients), patchify / unpatchify.  One wave per row, lanes along the feature
// dim (coalesced 256 B per wave-load), wave reductions by DPP shuffles.
#include "common.h"

namespace gaot {

// ---------------------------------------------------------------- RMSNorm (attn.py:161-172)
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          int M, int D, float eps, float* __restrict__ y,
                                                          float* __restrict__ rstd, float* __restrict__ y_amax) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= M) return;
    const float* xr = x + (long)row * D;
    float ss = 0.f;
    for (int d = lane; d < D; d += 64) { const float v = xr[d]; ss += v * v; }
    ss = wave_sum(ss);
    const float r = rsqrtf(ss / (float)D + eps);
    if (lane == 0) rstd[row] = r;
    float* yr = y + (long)row * D;
    float am = 0.f;
    for (int d = lane; d < D; d += 64) { const float o = xr[d] * r * w[d]; yr[d] = o; am = fmaxf(am, fabsf(o)); }
    if (y_amax) amax_publish(y_amax, am, lane, row);
}

constexpr int RMS_ROWS_PER_BLOCK = 8;
constexpr int RMS_MAX_D = 2048;

// dx = r*(w*dy) - x*r^3*mean(w*dy*x) (+dx_add) ; dw partial per block = sum_rows dy*x*r
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ rstd, const float* __restrict__ dy,
                                                          const float* __restrict__ dx_add, const float* __restrict__ dx_add2, int M, int D,
                                                          float* __restrict__ dx, float* __restrict__ dwp, float* __restrict__ dx_amax) {
    __shared__ float red[4][RMS_MAX_D];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float am = 0.f;
    float dwacc[RMS_MAX_D / 64];
#pragma unroll
    for (int i = 0; i < RMS_MAX_D / 64; ++i) dwacc[i] = 0.f;
    const int row_base = blockIdx.x * RMS_ROWS_PER_BLOCK;
    for (int rr = wave; rr < RMS_ROWS_PER_BLOCK; rr += 4) {
        const int row = row_base + rr;
        if (row >= M) break;
        const float* xr = x + (long)row * D;
        const float* gr = dy + (long)row * D;
        const float r = rstd[row];
        float dot = 0.f;
        for (int d = lane; d < D; d += 64) dot += w[d] * gr[d] * xr[d];
        dot = wave_sum(dot);
        const float coef = r * r * r * dot / (float)D;
        float* or_ = dx + (long)row * D;
        const float* ar = dx_add ? dx_add + (long)row * D : nullptr;
        const float* ar2 = dx_add2 ? dx_add2 + (long)row * D : nullptr;
#pragma unroll
        for (int i = 0; i < RMS_MAX_D / 64; ++i) {
            const int d = lane + i * 64;
            if (d < D) {
                const float xv = xr[d], gv = gr[d];
                float v = r * w[d] * gv - xv * coef;
                if (ar) v += ar[d];
                if (ar2) v += ar2[d];
                or_[d] = v;
                am = fmaxf(am, fabsf(v));
                dwacc[i] += gv * xv * r;
            }
        }
    }
    if (dx_amax) amax_publish(dx_amax, am, lane, (int)blockIdx.x * 4 + wave);
#pragma unroll
    for (int i = 0; i < RMS_MAX_D / 64; ++i) {
        const int d = lane + i * 64;
        if (d < D) red[wave][d] = dwacc[i];
    }
    __syncthreads();
    for (int d = threadIdx.x; d < D; d += 256)
        dwp[(long)blockIdx.x * D + d] = red[0][d] + red[1][d] + red[2][d] + red[3][d];
}

// 16-byte variants for D <= 256 * NV, D % 4 == 0 (256, 384, 512): a lane owns float4 columns lane*4 + 256*v where those lie inside the
// row; one pass over x / dy.
template <int NV, int D = 256 * NV>
__global__ __launch_bounds__(256) void rmsnorm_fwd_vec_kernel(const float* __restrict__ x, const float* __restrict__ w, int M,
                                                              float eps, float* __restrict__ y, float* __restrict__ rstd,
                                                              float* __restrict__ y_amax) {
    static_assert(D % 4 == 0 && D > 256 * (NV - 1) && D <= 256 * NV, "row width");
    __shared__ float amred[4];
    const int row_ = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const bool live = row_ < M;
    const int row = live ? row_ : M - 1;             // past the end: recompute the last row, store nothing (the block-wide publish needs every wave)
    f32x4 xv[NV];
    float ss = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const bool on = v * 256 + lane * 4 < D;
        xv[v] = on ? *reinterpret_cast<const f32x4*>(x + (long)row * D + v * 256 + lane * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        ss += xv[v][0] * xv[v][0] + xv[v][1] * xv[v][1] + xv[v][2] * xv[v][2] + xv[v][3] * xv[v][3];
    }
    ss = wave_sum(ss);
    const float r = rsqrtf(ss / (float)D + eps);
    if (lane == 0 && live) rstd[row] = r;
    float am = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const bool on = v * 256 + lane * 4 < D;
        if (!on) continue;
        const f32x4 wv = *reinterpret_cast<const f32x4*>(w + v * 256 + lane * 4);
        const f32x4 o = xv[v] * r * wv;
        if (live) *reinterpret_cast<f32x4*>(y + (long)row * D + v * 256 + lane * 4) = o;
        am = fmaxf(am, fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
    }
    if (y_amax) amax_publish_block<4>(y_amax, am, amred);
}
template <int NV, int D = 256 * NV>
__global__ __launch_bounds__(256) void rmsnorm_bwd_vec_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ rstd, const float* __restrict__ dy,
                                                              const float* __restrict__ dx_add, const float* __restrict__ dx_add2, int M,
                                                              float* __restrict__ dx, float* __restrict__ dwp, float* __restrict__ dx_amax,
                                                              int nz, long zstride, const float* __restrict__ dy_add) {
    // nz > 1: dy arrives as the K slabs of the split-K product that computed it (zstride apart; gaot_gemm_desc.raw_slabs), summed here in
    // slab order instead of by a reduce launch of its own; dy_add: a further addend of dy (that product's fused residual)
    static_assert(D % 4 == 0 && D > 256 * (NV - 1) && D <= 256 * NV, "row width");
    __shared__ f32x4 red[4][NV][64];
    __shared__ float amred[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float am = 0.f;
    f32x4 wv[NV], dwacc[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        wv[v] = v * 256 + lane * 4 < D ? *reinterpret_cast<const f32x4*>(w + v * 256 + lane * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        dwacc[v] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int row_base = blockIdx.x * RMS_ROWS_PER_BLOCK;
    for (int rr = wave; rr < RMS_ROWS_PER_BLOCK; rr += 4) {
        const int row = row_base + rr;
        if (row >= M) break;
        f32x4 xv[NV], gv[NV];
        float dot = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            if (v * 256 + lane * 4 >= D) { xv[v] = gv[v] = f32x4{0.f, 0.f, 0.f, 0.f}; continue; }
            xv[v] = *reinterpret_cast<const f32x4*>(x + (long)row * D + v * 256 + lane * 4);
            gv[v] = *reinterpret_cast<const f32x4*>(dy + (long)row * D + v * 256 + lane * 4);
            for (int z = 1; z < nz; ++z) gv[v] += *reinterpret_cast<const f32x4*>(dy + z * zstride + (long)row * D + v * 256 + lane * 4);
            if (dy_add) gv[v] += *reinterpret_cast<const f32x4*>(dy_add + (long)row * D + v * 256 + lane * 4);
            const f32x4 t = wv[v] * gv[v] * xv[v];
            dot += t[0] + t[1] + t[2] + t[3];
        }
        dot = wave_sum(dot);
        const float r = rstd[row];
        const float coef = r * r * r * dot / (float)D;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            if (v * 256 + lane * 4 >= D) continue;
            f32x4 o = wv[v] * gv[v] * r - xv[v] * coef;
            if (dx_add) o += *reinterpret_cast<const f32x4*>(dx_add + (long)row * D + v * 256 + lane * 4);
            if (dx_add2) o += *reinterpret_cast<const f32x4*>(dx_add2 + (long)row * D + v * 256 + lane * 4);
            *reinterpret_cast<f32x4*>(dx + (long)row * D + v * 256 + lane * 4) = o;
            am = fmaxf(am, fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
            dwacc[v] += gv[v] * xv[v] * r;
        }
    }
    if (dx_amax) amax_publish_block<4>(dx_amax, am, amred);
#pragma unroll
    for (int v = 0; v < NV; ++v) red[wave][v][lane] = dwacc[v];
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int v = 0; v < NV; ++v)
            if (v * 256 + lane * 4 < D)
                *reinterpret_cast<f32x4*>(dwp + (long)blockIdx.x * D + v * 256 + lane * 4) =
                    red[0][v][lane] + red[1][v][lane] + red[2][v][lane] + red[3][v][lane];
    }
}

// ---------------------------------------------------------------- SwiGLU gate (attn.py:151)
__global__ void swiglu_fwd_kernel(const float* __restrict__ u, long M, int F, float* __restrict__ g) {
    const long total4 = M * F / 4;
    const int F4 = F / 4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
        const long m = i / F4;
        const int f = (int)(i % F4) * 4;
        const f32x4 a = *reinterpret_cast<const f32x4*>(u + m * 2 * F + f);
        const f32x4 b = *reinterpret_cast<const f32x4*>(u + m * 2 * F + F + f);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (a[j] / (1.0f + expf(-a[j]))) * b[j];
        *reinterpret_cast<f32x4*>(g + m * F + f) = o;
    }
}
__global__ void swiglu_bwd_kernel(const float* __restrict__ u, const float* __restrict__ dg, long M, int F,
                                  float* __restrict__ du) {
    const long total4 = M * F / 4;
    const int F4 = F / 4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
        const long m = i / F4;
        const int f = (int)(i % F4) * 4;
        const f32x4 a = *reinterpret_cast<const f32x4*>(u + m * 2 * F + f);
        const f32x4 b = *reinterpret_cast<const f32x4*>(u + m * 2 * F + F + f);
        const f32x4 g = *reinterpret_cast<const f32x4*>(dg + m * F + f);
        f32x4 d1, d3;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float sg = 1.0f / (1.0f + expf(-a[j]));
            d1[j] = g[j] * b[j] * sg * (1.0f + a[j] * (1.0f - sg));
            d3[j] = g[j] * a[j] * sg;
        }
        *reinterpret_cast<f32x4*>(du + m * 2 * F + f) = d1;
        *reinterpret_cast<f32x4*>(du + m * 2 * F + F + f) = d3;
    }
}

// ---------------------------------------------------------------- reductions
// Column sums out[n] = sum_m x[m,n] for tall matrices (bias / norm-weight gradients).
// Stage 1: grid (column tiles of 64, R row chunks); a wave's lanes run along columns (coalesced 256 B rows) or,
// for narrow matrices (N <= 32), along (row, column) pairs so all 64 lanes stay busy; waves interleave rows.
// Stage 2: one block per 64 columns sums the R partial rows (R <= 256) with all 4 waves.
constexpr int COLSUM_MAX_CHUNKS = 256;
__host__ __device__ inline int colsum_chunks(int M, int N) {
    const int ctiles = (N + 63) / 64;
    int r = 1024 / ctiles;                       // ~1024 blocks in flight
    const int by_rows = (M + 63) / 64;           // at least 64 rows per chunk
    if (r > by_rows) r = by_rows;
    if (r > COLSUM_MAX_CHUNKS) r = COLSUM_MAX_CHUNKS;
    return r < 1 ? 1 : r;
}
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, long ld, int M, int N, int R,
                                                             float* __restrict__ part) {
    __shared__ float red[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int rows_per_chunk = (M + R - 1) / R;
    const int r0 = blockIdx.y * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
    float s = 0.f;
    if (N > 32) {
        const int n = blockIdx.x * 64 + lane;
        if (n < N)
            for (int r = r0 + wave; r < r1; r += 4) s += x[(long)r * ld + n];
        red[wave][lane] = s;
        __syncthreads();
        if (wave == 0 && n < N) part[(long)blockIdx.y * N + n] = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
    } else {
        int lpr = 1;
        while (lpr < N) lpr <<= 1;               // lanes per row (power of two <= 32)
        const int rpw = 64 / lpr;                // rows per wave-step
        const int n = lane % lpr, sub = lane / lpr;
        if (n < N)
            for (int r = r0 + wave * rpw + sub; r < r1; r += 4 * rpw) s += x[(long)r * ld + n];
        for (int off = 32; off >= lpr; off >>= 1) s += __shfl_xor(s, off, 64);
        red[wave][lane] = s;
        __syncthreads();
        if (wave == 0 && lane < N) part[(long)blockIdx.y * N + lane] = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
    }
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ part, int P, int N, float* __restrict__ out) {
    __shared__ float red[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (n < N)
        for (int p = wave; p < P; p += 4) s += part[(long)p * N + n];
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && n < N) out[n] = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
}
// short matrices (M <= 1024 rows, e.g. per-workgroup partial rows): ONE launch; 16 columns per workgroup as 4 float4
// lanes x 64 row groups (N / 16 workgroups: the matrix is tiny, parallelism comes from splitting the columns finely),
// four independent 16-byte loads in flight per thread, fixed-order LDS sum
__global__ __launch_bounds__(256) void colsum_small_kernel(const float* __restrict__ x, int M, int N, float* __restrict__ out) {
    __shared__ f32x4 red[64][4];
    const int c4 = threadIdx.x & 3, rg = threadIdx.x >> 2;
    const int n = blockIdx.x * 16 + c4 * 4;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    if (n < N) {
        int r = rg;
        for (; r + 192 < M; r += 256) {
            s0 += *reinterpret_cast<const f32x4*>(x + (long)r * N + n);
            s1 += *reinterpret_cast<const f32x4*>(x + (long)(r + 64) * N + n);
            s2 += *reinterpret_cast<const f32x4*>(x + (long)(r + 128) * N + n);
            s3 += *reinterpret_cast<const f32x4*>(x + (long)(r + 192) * N + n);
        }
        for (; r < M; r += 64) s0 += *reinterpret_cast<const f32x4*>(x + (long)r * N + n);
    }
    red[rg][c4] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rg < 4) {                      // 4 threads per float4 column: 16 row groups each, then a 2-step exchange
        f32x4 t = red[rg][c4];
#pragma unroll
        for (int g = 1; g < 16; ++g) t += red[rg + 4 * g][c4];
        red[rg][c4] = t;
    }
    __syncthreads();
    if (rg == 0 && n < N) *reinterpret_cast<f32x4*>(out + n) = (red[0][c4] + red[1][c4]) + (red[2][c4] + red[3][c4]);
}
// ---- grouped column sums: ONE launch over many small partial-row matrices (the per-workgroup partial rows of the norm-weight
// and lifting gradients: each used to cost its own 5 us launch at the end of its backward node).  Same tiling as colsum_small_kernel
// (16 columns per workgroup as 4 float4 lanes x 64 row groups, fixed-order LDS sum); the item table travels in the kernel arguments.
constexpr int COLSUM_GROUP_MAX = 64;       // 64 x 56 B of table in the kernel arguments
struct ColsumItem { const float* x; float* out; long ld; int M, N, wg_end, out_cols; long out_ld; };
struct ColsumGroupArgs { int n; ColsumItem it[COLSUM_GROUP_MAX]; };
__global__ __launch_bounds__(256) void colsum_grouped_kernel(const ColsumGroupArgs g) {
    __shared__ f32x4 red[64][4];
    int i = 0;
    while (i + 1 < g.n && (int)blockIdx.x >= g.it[i].wg_end) ++i;
    const int first = i > 0 ? g.it[i - 1].wg_end : 0;
    const float* __restrict__ x = g.it[i].x;
    const long ld = g.it[i].ld;
    const int M = g.it[i].M, N = g.it[i].N;
    const int c4 = threadIdx.x & 3, rg = threadIdx.x >> 2;
    const int n = ((int)blockIdx.x - first) * 16 + c4 * 4;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    if (n < N) {
        int r = rg;
        for (; r + 192 < M; r += 256) {
            s0 += *reinterpret_cast<const f32x4*>(x + (long)r * ld + n);
            s1 += *reinterpret_cast<const f32x4*>(x + (long)(r + 64) * ld + n);
            s2 += *reinterpret_cast<const f32x4*>(x + (long)(r + 128) * ld + n);
            s3 += *reinterpret_cast<const f32x4*>(x + (long)(r + 192) * ld + n);
        }
        for (; r < M; r += 64) s0 += *reinterpret_cast<const f32x4*>(x + (long)r * ld + n);
    }
    red[rg][c4] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rg < 4) {
        f32x4 t = red[rg][c4];
#pragma unroll
        for (int q = 1; q < 16; ++q) t += red[rg + 4 * q][c4];
        red[rg][c4] = t;
    }
    __syncthreads();
    if (rg == 0 && n < N) {
        const int oc = g.it[i].out_cols;          // the sums form an [N / oc, oc] matrix stored with row stride out_ld (oc = N: one row)
        float* dst = g.it[i].out + (long)(n / oc) * g.it[i].out_ld + (n % oc);
        *reinterpret_cast<f32x4*>(dst) = (red[0][c4] + red[1][c4]) + (red[2][c4] + red[3][c4]);
    }
}
__global__ void batchsum_kernel(const float* __restrict__ x, int B, long RN, float* __restrict__ out) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < RN; i += (long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += x[(long)b * RN + i];
        out[i] = s;
    }
}

// ---------------------------------------------------------------- patchify (gaot.py:182-185,202-205,224-231)
// tokens[b, s, k]: s = patch id (row-major over patch grid), k = (within-patch offsets row-major, c)
__global__ void patchify_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int H, int W, int Dz,
                                int P, int C, int inverse) {
    const int dim = Dz > 0 ? 3 : 2;
    const int D1 = Dz > 0 ? Dz : 1;
    const long nodes = (long)H * W * D1;
    const long total = (long)B * nodes * C;
    const int pw = W / P, pd = D1 > 1 ? D1 / P : 1;
    const int pvol = dim == 3 ? P * P * P : P * P;
    for (long gid = (long)blockIdx.x * blockDim.x + threadIdx.x; gid < total; gid += (long)gridDim.x * blockDim.x) {
        // gid enumerates the GRID layout [b, node, c]
        const int c = (int)(gid % C);
        const long node = (gid / C) % nodes;
        const long b = gid / (C * nodes);
        int h, w_, z = 0;
        if (dim == 3) { z = (int)(node % D1); w_ = (int)((node / D1) % W); h = (int)(node / ((long)D1 * W)); }
        else { w_ = (int)(node % W); h = (int)(node / W); }
        const int ph = h / P, i = h % P, pwi = w_ / P, j = w_ % P;
        long s, k;
        if (dim == 3) {
            const int pz = z / P, l = z % P;
            s = ((long)ph * pw + pwi) * pd + pz;
            k = (((long)i * P + j) * P + l) * C + c;
        } else {
            s = (long)ph * pw + pwi;
            k = ((long)i * P + j) * C + c;
        }
        const long tok = (b * (nodes / pvol) + s) * ((long)pvol * C) + k;
        if (inverse) out[gid] = in[tok]; else out[tok] = in[gid];
    }
}

// ---------------------------------------------------------------- AdamW over flat buffers (torch.optim.AdamW semantics,
// the optimizer at optimizers.py:196): decoupled weight decay, bias-corrected moments, eps added after the bias correction
// of sqrt(v).  `step` lives on the device so the launch is hipGraph-replayable; a 1-thread kernel advances it first.
__global__ void adamw_tick_kernel(float* step) { step[0] += 1.0f; }
// hyper (optional, DEVICE): {lr, beta1, beta2, eps, weight_decay} read at run time, so a captured launch follows a learning-rate
// schedule without re-capture (the scalar arguments are ignored then)
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, long n, float lr, float b1, float b2, float eps,
                                                    float wd, const float* __restrict__ step, const float* __restrict__ hyper) {
    if (hyper) { lr = hyper[0]; b1 = hyper[1]; b2 = hyper[2]; eps = hyper[3]; wd = hyper[4]; }
    const float t = step[0];
    const float bc1 = 1.0f - powf(b1, t);
    const float bc2 = 1.0f - powf(b2, t);
    const float step_size = lr / bc1;
    const float inv_bc2_sqrt = 1.0f / sqrtf(bc2);
    const long n4 = n / 4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        f32x4 pv = reinterpret_cast<f32x4*>(p)[i];
        const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
        f32x4 mv = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            pv[q] *= 1.0f - lr * wd;
            mv[q] = mv[q] + (gv[q] - mv[q]) * (1.0f - b1);          // lerp form, as torch's fused kernel
            vv[q] = b2 * vv[q] + (1.0f - b2) * gv[q] * gv[q];
            const float denom = sqrtf(vv[q]) * inv_bc2_sqrt + eps;
            pv[q] -= step_size * (mv[q] / denom);
        }
        reinterpret_cast<f32x4*>(p)[i] = pv;
        reinterpret_cast<f32x4*>(m)[i] = mv;
        reinterpret_cast<f32x4*>(v)[i] = vv;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {      // scalar tail
        const long i = n4 * 4 + threadIdx.x;
        float pv = p[i] * (1.0f - lr * wd);
        const float mv = m[i] + (g[i] - m[i]) * (1.0f - b1);
        const float vv = b2 * v[i] + (1.0f - b2) * g[i] * g[i];
        pv -= step_size * (mv / (sqrtf(vv) * inv_bc2_sqrt + eps));
        p[i] = pv; m[i] = mv; v[i] = vv;
    }
}

// 16-byte patchify: one thread per 4 channels of a grid node (C % 4 == 0); a node's C channels stay contiguous on both sides
// IDX = unsigned (everything below 2^31 elements: the index arithmetic is five divisions per 16 bytes moved, and 64-bit ones cost four
// times the 32-bit ones) or long
template <typename IDX>
__global__ void patchify_vec_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int H, int W, int Dz, int P,
                                    int C, int inverse, float* __restrict__ out_amax) {
    __shared__ float amred[4];
    float am = 0.f;
    const int dim = Dz > 0 ? 3 : 2;
    const int D1 = Dz > 0 ? Dz : 1;
    const int C4 = C / 4;
    const IDX nodes = (IDX)H * W * D1;
    const IDX total = (IDX)B * nodes * C4;
    const int pw = W / P, pd = D1 > 1 ? D1 / P : 1;
    const int pvol = dim == 3 ? P * P * P : P * P;
    const IDX toks = nodes / pvol;
    for (IDX gid = (IDX)blockIdx.x * blockDim.x + threadIdx.x; gid < total; gid += (IDX)gridDim.x * blockDim.x) {
        const int c = (int)(gid % C4) * 4;
        const IDX nc = gid / C4;
        const IDX node = nc % nodes;
        const IDX b = nc / nodes;
        int h, w_, z = 0;
        if (dim == 3) { z = (int)(node % D1); w_ = (int)((node / D1) % W); h = (int)(node / ((IDX)D1 * W)); }
        else { w_ = (int)(node % W); h = (int)(node / W); }
        const int ph = h / P, i = h % P, pwi = w_ / P, j = w_ % P;
        IDX s, k;
        if (dim == 3) {
            const int pz = z / P, l = z % P;
            s = ((IDX)ph * pw + pwi) * pd + pz;
            k = (((IDX)i * P + j) * P + l) * C + c;
        } else {
            s = (IDX)ph * pw + pwi;
            k = ((IDX)i * P + j) * C + c;
        }
        const IDX tok = (b * toks + s) * ((IDX)pvol * C) + k;
        const IDX g = (b * nodes + node) * C + c;
        const f32x4 v = *reinterpret_cast<const f32x4*>(inverse ? in + tok : in + g);
        *reinterpret_cast<f32x4*>(inverse ? out + g : out + tok) = v;
        am = fmaxf(am, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    }
    if (out_amax) amax_publish_block<4>(out_amax, am, amred);
}

}  // namespace gaot

using namespace gaot;
#define ST(s) reinterpret_cast<hipStream_t>(s)

extern "C" int gaot_rmsnorm_fwd(const float* x, const float* w, int32_t M, int32_t D, float eps, float* y, float* rstd, float* y_absmax,
                                gaot_stream_t stream) {
    GAOT_REQUIRE(x && w && y && rstd && M > 0 && D > 0, "rmsnorm_fwd: bad arguments");
    const bool v16 = aligned16(x) && aligned16(w) && aligned16(y);
    if (v16 && D == 256)      hipLaunchKernelGGL(rmsnorm_fwd_vec_kernel<1>, dim3(cdiv(M, 4)), dim3(256), 0, ST(stream), x, w, M, eps, y, rstd, y_absmax);
    else if (v16 && D == 512) hipLaunchKernelGGL(rmsnorm_fwd_vec_kernel<2>, dim3(cdiv(M, 4)), dim3(256), 0, ST(stream), x, w, M, eps, y, rstd, y_absmax);
    else if (v16 && D == 384) hipLaunchKernelGGL((rmsnorm_fwd_vec_kernel<2, 384>), dim3(cdiv(M, 4)), dim3(256), 0, ST(stream), x, w, M, eps, y, rstd, y_absmax);
    else hipLaunchKernelGGL(rmsnorm_fwd_kernel, dim3(cdiv(M, 4)), dim3(256), 0, ST(stream), x, w, M, D, eps, y, rstd, y_absmax);
    GAOT_CHECK_LAUNCH("gaot_rmsnorm_fwd");
    return GAOT_OK;
}

extern "C" int gaot_rmsnorm_bwd_partials(int32_t M) { return cdiv(M, RMS_ROWS_PER_BLOCK); }

extern "C" int gaot_rmsnorm_bwd(const float* x, const float* w, const float* rstd, const float* dy, const float* dx_add,
                                const float* dx_add2, int32_t M, int32_t D, float* dx, float* dw_partial, float* dx_absmax,
                                gaot_stream_t stream) {
    GAOT_REQUIRE(x && w && rstd && dy && dx && dw_partial && M > 0 && D > 0, "rmsnorm_bwd: bad arguments");
    GAOT_REQUIRE(D <= RMS_MAX_D, "rmsnorm_bwd: D=%d exceeds %d", D, RMS_MAX_D);
    const bool v16 = aligned16(x) && aligned16(w) && aligned16(dy) && aligned16(dx) && aligned16(dw_partial) && (!dx_add || aligned16(dx_add)) &&
                     (!dx_add2 || aligned16(dx_add2));
    const dim3 grid(cdiv(M, RMS_ROWS_PER_BLOCK));
    if (v16 && D == 256)      hipLaunchKernelGGL(rmsnorm_bwd_vec_kernel<1>, grid, dim3(256), 0, ST(stream), x, w, rstd, dy, dx_add, dx_add2, M, dx, dw_partial, dx_absmax, 1, 0L, (const float*)nullptr);
    else if (v16 && D == 512) hipLaunchKernelGGL(rmsnorm_bwd_vec_kernel<2>, grid, dim3(256), 0, ST(stream), x, w, rstd, dy, dx_add, dx_add2, M, dx, dw_partial, dx_absmax, 1, 0L, (const float*)nullptr);
    else if (v16 && D == 384) hipLaunchKernelGGL((rmsnorm_bwd_vec_kernel<2, 384>), grid, dim3(256), 0, ST(stream), x, w, rstd, dy, dx_add, dx_add2, M, dx, dw_partial, dx_absmax, 1, 0L, (const float*)nullptr);
    else hipLaunchKernelGGL(rmsnorm_bwd_kernel, grid, dim3(256), 0, ST(stream), x, w, rstd, dy, dx_add, dx_add2, M, D, dx, dw_partial, dx_absmax);
    GAOT_CHECK_LAUNCH("gaot_rmsnorm_bwd");
    return GAOT_OK;
}

extern "C" int gaot_rmsnorm_bwd_slabs(const float* x, const float* w, const float* rstd, const float* dy_slabs, int32_t n_slabs, int64_t slab_stride,
                                      const float* dy_add, const float* dx_add, const float* dx_add2, int32_t M, int32_t D, float* dx,
                                      float* dw_partial, float* dx_absmax, gaot_stream_t stream) {
    GAOT_REQUIRE(x && w && rstd && dy_slabs && dx && dw_partial && M > 0 && n_slabs >= 1 && (n_slabs == 1 || slab_stride >= (int64_t)M * D),
                 "rmsnorm_bwd_slabs: bad arguments");
    GAOT_REQUIRE((D == 256 || D == 384 || D == 512) && slab_stride % 4 == 0 && aligned16(x) && aligned16(w) && aligned16(dy_slabs) && aligned16(dx) && aligned16(dw_partial) &&
                 (!dy_add || aligned16(dy_add)) && (!dx_add || aligned16(dx_add)) && (!dx_add2 || aligned16(dx_add2)),
                 "rmsnorm_bwd_slabs: D must be 256, 384 or 512 (got %d) and every pointer 16-byte aligned", D);
    const dim3 grid(cdiv(M, RMS_ROWS_PER_BLOCK));
    if (D == 256) hipLaunchKernelGGL(rmsnorm_bwd_vec_kernel<1>, grid, dim3(256), 0, ST(stream), x, w, rstd, dy_slabs, dx_add, dx_add2, M, dx, dw_partial, dx_absmax, n_slabs, (long)slab_stride, dy_add);
    else if (D == 384) hipLaunchKernelGGL((rmsnorm_bwd_vec_kernel<2, 384>), grid, dim3(256), 0, ST(stream), x, w, rstd, dy_slabs, dx_add, dx_add2, M, dx, dw_partial, dx_absmax, n_slabs, (long)slab_stride, dy_add);
    else          hipLaunchKernelGGL(rmsnorm_bwd_vec_kernel<2>, grid, dim3(256), 0, ST(stream), x, w, rstd, dy_slabs, dx_add, dx_add2, M, dx, dw_partial, dx_absmax, n_slabs, (long)slab_stride, dy_add);
    GAOT_CHECK_LAUNCH("gaot_rmsnorm_bwd_slabs");
    return GAOT_OK;
}

extern "C" int gaot_swiglu_fwd(const float* u, int32_t M, int32_t F, float* g, gaot_stream_t stream) {
    GAOT_REQUIRE(u && g && M > 0 && F > 0 && F % 4 == 0 && aligned16(u) && aligned16(g), "swiglu_fwd: bad arguments (F %% 4 == 0, 16B aligned)");
    hipLaunchKernelGGL(swiglu_fwd_kernel, dim3(cap_blocks((long)M * F / 4, 256, 4096)), dim3(256), 0, ST(stream), u, (long)M, F, g);
    GAOT_CHECK_LAUNCH("gaot_swiglu_fwd");
    return GAOT_OK;
}

// out[i] = g[i] * act'(z[i]): the derivative of a chain's LAST activation (no following product to fuse it into)
__global__ void act_bwd_kernel(const float* __restrict__ g, const float* __restrict__ z, long n, int act, float* __restrict__ out) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float zv = z[i];
        out[i] = act == GAOT_ACT_GELU ? g[i] * gelu_grad_f(zv) : (zv > 0.f ? g[i] : 0.f);
    }
}
extern "C" int gaot_act_bwd(const float* g, const float* z, int64_t n, int32_t act, float* out, gaot_stream_t stream) {
    GAOT_REQUIRE(g && z && out && n > 0 && (act == GAOT_ACT_GELU || act == GAOT_ACT_RELU), "act_bwd: bad arguments");
    hipLaunchKernelGGL(act_bwd_kernel, dim3(cap_blocks(n, 256, 4096)), dim3(256), 0, ST(stream), g, z, (long)n, act, out);
    GAOT_CHECK_LAUNCH("gaot_act_bwd");
    return GAOT_OK;
}
extern "C" int gaot_swiglu_bwd(const float* u, const float* dg, int32_t M, int32_t F, float* du, gaot_stream_t stream) {
    GAOT_REQUIRE(u && dg && du && M > 0 && F > 0 && F % 4 == 0 && aligned16(u) && aligned16(dg) && aligned16(du),
                 "swiglu_bwd: bad arguments (F %% 4 == 0, 16B aligned)");
    hipLaunchKernelGGL(swiglu_bwd_kernel, dim3(cap_blocks((long)M * F / 4, 256, 4096)), dim3(256), 0, ST(stream), u, dg, (long)M, F, du);
    GAOT_CHECK_LAUNCH("gaot_swiglu_bwd");
    return GAOT_OK;
}

extern "C" int64_t gaot_colsum_scratch(int32_t M, int32_t N) { return (int64_t)colsum_chunks(M, N) * N; }

extern "C" int gaot_colsum(const float* x, int64_t ld, int32_t M, int32_t N, float* out, float* scratch,
                           gaot_stream_t stream) {
    GAOT_REQUIRE(x && out && scratch && M > 0 && N > 0, "colsum: bad arguments");
    if (M <= 1024 && ld == N && N % 4 == 0 && aligned16(x) && aligned16(out)) {   // per-workgroup partial rows of the norm-weight gradients
        hipLaunchKernelGGL(colsum_small_kernel, dim3(cdiv(N, 16)), dim3(256), 0, ST(stream), x, M, N, out);
        GAOT_CHECK_LAUNCH("gaot_colsum");
        return GAOT_OK;
    }
    const int P = colsum_chunks(M, N);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(cdiv(N, 64), P), dim3(256), 0, ST(stream), x, (long)ld, M, N, P, scratch);
    hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(N, 64)), dim3(256), 0, ST(stream), scratch, P, N, out);
    GAOT_CHECK_LAUNCH("gaot_colsum");
    return GAOT_OK;
}

extern "C" int gaot_colsum_grouped(const gaot_colsum_item* items, int32_t n, gaot_stream_t stream) {
    GAOT_REQUIRE(items != nullptr && n > 0, "colsum_grouped: no items");
    for (int i = 0; i < n; ++i)
        GAOT_REQUIRE(items[i].x && items[i].out && items[i].M > 0 && items[i].N > 0 && items[i].N % 4 == 0 && items[i].ld % 4 == 0 &&
                     items[i].ld >= items[i].N && aligned16(items[i].x) && aligned16(items[i].out) &&
                     (items[i].out_cols == 0 || (items[i].out_cols % 4 == 0 && items[i].N % items[i].out_cols == 0 && items[i].out_ld % 4 == 0 &&
                                                 items[i].out_ld >= items[i].out_cols)),
                     "colsum_grouped: item %d needs N %% 4 == 0, ld %% 4 == 0, 16-byte aligned pointers (and out_cols | N, out_cols, out_ld %% 4 == 0)", i);
    for (int i0 = 0; i0 < n; i0 += COLSUM_GROUP_MAX) {
        ColsumGroupArgs a;
        a.n = n - i0 < COLSUM_GROUP_MAX ? n - i0 : COLSUM_GROUP_MAX;
        int wg = 0;
        for (int i = 0; i < a.n; ++i) {
            const gaot_colsum_item& it = items[i0 + i];
            wg += cdiv(it.N, 16);
            a.it[i] = ColsumItem{it.x, it.out, (long)it.ld, it.M, it.N, wg, it.out_cols > 0 ? it.out_cols : it.N, it.out_cols > 0 ? (long)it.out_ld : (long)it.N};
        }
        hipLaunchKernelGGL(colsum_grouped_kernel, dim3(wg), dim3(256), 0, ST(stream), a);
        GAOT_CHECK_LAUNCH("gaot_colsum_grouped");
    }
    return GAOT_OK;
}

// ---- fp16 pieces of weight matrices (gaot_gemm_desc.b_planes), many matrices per launch.  64 x 64 tiles through LDS so that both the
// plain and the transposed form are written from coalesced reads.  Layout [row][k / 16][piece][k % 16]: the tile kernels fetch the two
// pieces of a row's 16-wide k group as ONE 64-byte segment (two separate planes were two 32-byte requests: the L2 saw three requests
// per row and k group where one and a half now do -- tools/pmc_cache.sh: its request rate is what the k-loops run into).
constexpr int F16PL_GROUP_MAX = 64;
struct F16PlItem { const float* src; const float* amax; unsigned short* pk; unsigned short* pt; long ld; int rows, cols, tiles_c, wg_end; };
struct F16PlGroupArgs { int n; F16PlItem it[F16PL_GROUP_MAX]; };
__global__ __launch_bounds__(256) void split_f16_planes_kernel(const F16PlGroupArgs g) {
    __shared__ float tile[64][65];
    int i = 0;
    while (i + 1 < g.n && (int)blockIdx.x >= g.it[i].wg_end) ++i;
    const int local = (int)blockIdx.x - (i > 0 ? g.it[i - 1].wg_end : 0);
    const int r0 = (local / g.it[i].tiles_c) * 64, c0 = (local % g.it[i].tiles_c) * 64;
    const int rows = g.it[i].rows, cols = g.it[i].cols;
    const float* __restrict__ src = g.it[i].src;
    const long ld = g.it[i].ld;
    const int tid = threadIdx.x;
    float sc, inv;
    amax_scale(g.it[i].amax, sc, inv);
    for (int e = tid; e < 64 * 64; e += 256) {
        const int r = e >> 6, c = e & 63;
        tile[r][c] = (r0 + r < rows && c0 + c < cols) ? src[(long)(r0 + r) * ld + c0 + c] : 0.f;
    }
    __syncthreads();
    // the matrix's spread for the consumers (gemm_split.hip, "Dynamic range of the fp16 pieces"): L = log2(scaled tensor maximum / geometric
    // mean of the non-zero maxima of the aligned groups of 8) along each row stretch (float 1 of the word's first line) and along each column
    // stretch (float 2) of this tile, the largest over the matrix by an atomic max on the float's bits (L >= 0; zero with every new arena)
    if (tid < 128) {
        const bool along_row = tid < 64;
        const int t = tid & 63;
        unsigned es = 0u, cn = 0u;
#pragma unroll
        for (int grp = 0; grp < 8; ++grp) {
            float mx = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) mx = fmaxf(mx, fabsf(along_row ? tile[t][8 * grp + j] : tile[8 * grp + j][t]));
            const unsigned e = __float_as_uint(mx) >> 23;
            es += e;
            cn += e != 0u ? 1u : 0u;
        }
        float L = 0.f;
        if (cn > 0u) L = fmaxf(0.f, 13.5f - ((float)es / (float)cn - 127.f + (float)((int)(__float_as_uint(sc) >> 23) - 127)));
        // wave 0 holds the 64 row stretches, wave 1 the 64 column stretches of this tile.  What is published is the tile's EIGHTH-largest L
        // (to a binade): a vote, as in the consumers -- one dead output feature among 64 does not send every product of the weight through
        // the slow pass, a row 10^6 times larger than the rest (63 of 64 stretches far below it) does.  One atomic per wave.
        float Lq = 0.f;
#pragma unroll
        for (int cand = 4; cand <= 40; cand += 2) {
            const int n = __builtin_popcountll(__ballot(L >= (float)cand));
            if (n >= 8) Lq = (float)(cand + 1);          // (the eighth-largest lies in [cand, cand + 2))
        }
        if (t == 0 && Lq > 0.f) atomicMax(reinterpret_cast<unsigned*>(const_cast<float*>(g.it[i].amax)) + (along_row ? 1 : 2), __float_as_uint(Lq));
    }
    const int pp = tid & 15;                        // four consecutive output columns per thread: 8-byte stores of h and of m
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {          // 0: as stored [rows][cols]; 1: transposed [cols][rows]
        unsigned short* __restrict__ out = pass == 0 ? g.it[i].pk : g.it[i].pt;
        if (out == nullptr) continue;
        const bool tr = pass == 1;
        const int orows = tr ? cols : rows, ocols = tr ? rows : cols, or0 = tr ? c0 : r0, oc0 = tr ? r0 : c0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int orow = (tid >> 4) + 16 * k;
            float x[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = tr ? tile[4 * pp + e][orow] : tile[orow][4 * pp + e];
            u32x2 h, m;
            unsigned a_, b_;
            split2h_pair(x[0], x[1], sc, a_, b_); h[0] = a_; m[0] = b_;
            split2h_pair(x[2], x[3], sc, a_, b_); h[1] = a_; m[1] = b_;
            if (or0 + orow < orows && oc0 + 4 * pp + 3 < ocols) {          // rows, cols multiples of 16 (checked on the host): quads are in or out
                // the two pieces of a 16-wide k group sit side by side: [row][k / 16][piece][k % 16] -- one 64-byte segment per row and group
                const int oc = oc0 + 4 * pp;
                unsigned short* d = out + (long)(or0 + orow) * (2 * ocols) + (oc >> 4) * 32 + (oc & 15);
                *reinterpret_cast<u32x2*>(d) = h;
                *reinterpret_cast<u32x2*>(d + 16) = m;
            }
        }
    }
}

// ---- grouped absmax: the magnitude words of the fp16-piece products (gaot_gemm_desc.a_absmax ...), n matrices per launch.  A workgroup
// streams its share of one matrix (float4 loads when the rows allow, no per-element division), wave-reduces and publishes one atomic
// max per wave into the word's slot.
constexpr int ABSMAX_GROUP_MAX = 96;
struct AbsmaxItem { const float* x; float* out; long ld; int rows, cols, wg_end; };
struct AbsmaxGroupArgs { int n; AbsmaxItem it[ABSMAX_GROUP_MAX]; };
__global__ __launch_bounds__(256) void absmax_grouped_kernel(const AbsmaxGroupArgs g) {
    int lo_ = 0, hi_ = g.n - 1;                      // first item whose wg_end exceeds this workgroup's index (binary search: the table
    while (lo_ < hi_) {                              // sits in the kernel arguments, a linear scan is ~90 dependent scalar loads)
        const int mid = (lo_ + hi_) >> 1;
        if ((int)blockIdx.x >= g.it[mid].wg_end) lo_ = mid + 1; else hi_ = mid;
    }
    const int i = lo_;
    const int first = i > 0 ? g.it[i - 1].wg_end : 0, nwg = g.it[i].wg_end - first, b = (int)blockIdx.x - first;
    const float* __restrict__ x = g.it[i].x;
    const long ld = g.it[i].ld;
    const int rows = g.it[i].rows, cols = g.it[i].cols;
    float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f;
    const bool vec = (cols & 3) == 0 && (ld & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) & 15u) == 0);
    if (vec && ld == cols) {                     // contiguous: one flat stream of float4, four in flight per thread
        const long total = (long)rows * (cols >> 2), step = (long)nwg * 256;
        const f32x4* __restrict__ x4 = reinterpret_cast<const f32x4*>(x);
        long idx = (long)b * 256 + threadIdx.x;
        for (; idx + 3 * step < total; idx += 4 * step) {
            const f32x4 v0 = x4[idx], v1 = x4[idx + step], v2 = x4[idx + 2 * step], v3 = x4[idx + 3 * step];
            m0 = fmaxf(m0, fmaxf(fmaxf(fabsf(v0[0]), fabsf(v0[1])), fmaxf(fabsf(v0[2]), fabsf(v0[3]))));
            m1 = fmaxf(m1, fmaxf(fmaxf(fabsf(v1[0]), fabsf(v1[1])), fmaxf(fabsf(v1[2]), fabsf(v1[3]))));
            m2 = fmaxf(m2, fmaxf(fmaxf(fabsf(v2[0]), fabsf(v2[1])), fmaxf(fabsf(v2[2]), fabsf(v2[3]))));
            m3 = fmaxf(m3, fmaxf(fmaxf(fabsf(v3[0]), fabsf(v3[1])), fmaxf(fabsf(v3[2]), fabsf(v3[3]))));
        }
        for (; idx < total; idx += step) {
            const f32x4 v0 = x4[idx];
            m0 = fmaxf(m0, fmaxf(fmaxf(fabsf(v0[0]), fabsf(v0[1])), fmaxf(fabsf(v0[2]), fabsf(v0[3]))));
        }
    } else if (vec) {                            // strided rows: a wave per row
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        for (long r = (long)b * 4 + wave; r < rows; r += (long)nwg * 4)
            for (int c = lane * 4; c < cols; c += 256) {
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(x + r * ld + c);
                m0 = fmaxf(m0, fmaxf(fmaxf(fabsf(v0[0]), fabsf(v0[1])), fmaxf(fabsf(v0[2]), fabsf(v0[3]))));
            }
    } else {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        for (long r = (long)b * 4 + wave; r < rows; r += (long)nwg * 4)
            for (int c = lane; c < cols; c += 64) m0 = fmaxf(m0, fabsf(x[r * ld + c]));
    }
    amax_publish(g.it[i].out, fmaxf(fmaxf(m0, m1), fmaxf(m2, m3)), threadIdx.x & 63, (int)blockIdx.x * 4 + (threadIdx.x >> 6));
}

extern "C" int gaot_split_f16_planes_grouped(const gaot_f16_planes_item* items, int32_t n, gaot_stream_t stream) {
    GAOT_REQUIRE(items != nullptr && n > 0, "split_f16_planes_grouped: no items");
    for (int i = 0; i < n; ++i) {
        const gaot_f16_planes_item& it = items[i];
        GAOT_REQUIRE(it.src && it.absmax && (it.planes_k || it.planes_t) && it.rows > 0 && it.cols > 0 && it.rows % 16 == 0 && it.cols % 16 == 0 &&
                     it.ld >= it.cols && (reinterpret_cast<uintptr_t>(it.planes_k) & 15u) == 0 && (reinterpret_cast<uintptr_t>(it.planes_t) & 15u) == 0,
                     "split_f16_planes_grouped: item %d: rows, cols multiples of 16, 16-byte aligned planes", i);
    }
    for (int i0 = 0; i0 < n; i0 += F16PL_GROUP_MAX) {
        F16PlGroupArgs a;
        a.n = n - i0 < F16PL_GROUP_MAX ? n - i0 : F16PL_GROUP_MAX;
        int wg = 0;
        for (int i = 0; i < a.n; ++i) {
            const gaot_f16_planes_item& it = items[i0 + i];
            const int tc = cdiv(it.cols, 64);
            wg += cdiv(it.rows, 64) * tc;
            a.it[i] = F16PlItem{it.src, it.absmax, reinterpret_cast<unsigned short*>(it.planes_k), reinterpret_cast<unsigned short*>(it.planes_t),
                                (long)it.ld, it.rows, it.cols, tc, wg};
        }
        hipLaunchKernelGGL(split_f16_planes_kernel, dim3(wg), dim3(256), 0, ST(stream), a);
        GAOT_CHECK_LAUNCH("gaot_split_f16_planes_grouped");
    }
    return GAOT_OK;
}

extern "C" int gaot_absmax_grouped(const gaot_absmax_item* items, int32_t n, gaot_stream_t stream) {
    GAOT_REQUIRE(items != nullptr && n > 0, "absmax_grouped: no items");
    for (int i = 0; i < n; ++i)
        GAOT_REQUIRE(items[i].x && items[i].out && items[i].rows > 0 && items[i].cols > 0 && items[i].ld >= items[i].cols,
                     "absmax_grouped: item %d needs x, out, rows, cols > 0 and ld >= cols", i);
    for (int i0 = 0; i0 < n; i0 += ABSMAX_GROUP_MAX) {
        AbsmaxGroupArgs a;
        a.n = n - i0 < ABSMAX_GROUP_MAX ? n - i0 : ABSMAX_GROUP_MAX;
        int wg = 0;
        for (int i = 0; i < a.n; ++i) {
            const gaot_absmax_item& it = items[i0 + i];
            long w = ((long)it.rows * it.cols + 8191) / 8192;          // >= 32 KB of floats per workgroup
            wg += (int)(w > 1024 ? 1024 : (w < 1 ? 1 : w));
            a.it[i] = AbsmaxItem{it.x, it.out, (long)it.ld, it.rows, it.cols, wg};
        }
        hipLaunchKernelGGL(absmax_grouped_kernel, dim3(wg), dim3(256), 0, ST(stream), a);
        GAOT_CHECK_LAUNCH("gaot_absmax_grouped");
    }
    return GAOT_OK;
}

extern "C" int gaot_batchsum(const float* x, int32_t B, int64_t RN, float* out, gaot_stream_t stream) {
    GAOT_REQUIRE(x && out && B > 0 && RN > 0, "batchsum: bad arguments");
    hipLaunchKernelGGL(batchsum_kernel, dim3(cap_blocks(RN, 256, 4096)), dim3(256), 0, ST(stream), x, B, (long)RN, out);
    GAOT_CHECK_LAUNCH("gaot_batchsum");
    return GAOT_OK;
}

extern "C" int gaot_patchify(const float* in, int32_t B, int32_t H, int32_t W, int32_t Dz, int32_t P, int32_t C,
                             float* out, int32_t inverse, float* out_absmax, gaot_stream_t stream) {
    GAOT_REQUIRE(in && out && B > 0 && H > 0 && W > 0 && Dz >= 0 && P > 0 && C > 0, "patchify: bad arguments");
    GAOT_REQUIRE(H % P == 0 && W % P == 0 && (Dz == 0 || Dz % P == 0), "patchify: grid %dx%dx%d not divisible by patch %d", H, W, Dz, P);
    const long total = (long)B * H * W * (Dz > 0 ? Dz : 1) * C;
    if (C % 4 == 0 && aligned16(in) && aligned16(out))
    {
        if (total < (1L << 31)) hipLaunchKernelGGL(patchify_vec_kernel<unsigned>, dim3(cap_blocks(total / 4, 256, 8192)), dim3(256), 0, ST(stream), in, out, B, H, W, Dz, P, C, inverse, out_absmax);
        else hipLaunchKernelGGL(patchify_vec_kernel<long>, dim3(cap_blocks(total / 4, 256, 8192)), dim3(256), 0, ST(stream), in, out, B, H, W, Dz, P, C, inverse, out_absmax);
    }
    else {
        hipLaunchKernelGGL(patchify_kernel, dim3(cap_blocks(total, 256, 8192)), dim3(256), 0, ST(stream), in, out, B, H, W, Dz, P, C, inverse);
        if (out_absmax) {           // the scalar kernel does not publish: one grouped-absmax launch over the output
            const int32_t cols = inverse ? C : C * (Dz > 0 ? P * P * P : P * P);
            gaot_absmax_item it = {out, (int64_t)cols, (int32_t)(total / cols), cols, out_absmax};
            if (int rc = gaot_absmax_grouped(&it, 1, stream)) return rc;
        }
    }
    GAOT_CHECK_LAUNCH("gaot_patchify");
    return GAOT_OK;
}

// ---------------------------------------------------------------- MSE loss (base_trainer.py:71 nn.MSELoss(), mean reduction)
__global__ __launch_bounds__(256) void mse_partial_kernel(const float* __restrict__ p, const float* __restrict__ y, long n, float* __restrict__ part) {
    __shared__ float red[4];
    float s = 0.f;
    const long n4 = n / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 a = reinterpret_cast<const f32x4*>(p)[i], b = reinterpret_cast<const f32x4*>(y)[i];
        const f32x4 d = a - b;
        s += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n - n4 * 4)) { const float d = p[n4 * 4 + threadIdx.x] - y[n4 * 4 + threadIdx.x]; s += d * d; }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(64) void mse_final_kernel(const float* __restrict__ part, int nparts, float inv_n, float* __restrict__ loss) {
    float s = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 64) s += part[i];      // fixed assignment + fixed-order butterfly: deterministic
    s = wave_sum(s);
    if (threadIdx.x == 0) loss[0] = s * inv_n;
}
__global__ void mse_bwd_kernel(const float* __restrict__ p, const float* __restrict__ y, long n, const float* __restrict__ gout, float two_over_n,
                               float* __restrict__ dp) {
    const float c = two_over_n * gout[0];
    const long n4 = n / 4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 a = reinterpret_cast<const f32x4*>(p)[i], b = reinterpret_cast<const f32x4*>(y)[i];
        reinterpret_cast<f32x4*>(dp)[i] = (a - b) * c;
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n - n4 * 4)) dp[n4 * 4 + threadIdx.x] = (p[n4 * 4 + threadIdx.x] - y[n4 * 4 + threadIdx.x]) * c;
}
// forward AND the gradient for a unit seed in ONE launch (trainer.TrainStep: the loss of a training step is differentiated with
// d loss = 1): every workgroup writes its slice of dpred = 2 (p - y) / n and its partial sum; the LAST one to finish (ticket) adds the
// partials in the order mse_final_kernel would -- the same bits -- and, optionally, advances an optimizer's device-resident step
// counter (the 1-thread tick launch of gaot_adamw_step*, folded in here).  The ticket returns to zero.
__global__ __launch_bounds__(256) void mse_fused_kernel(const float* __restrict__ p, const float* __restrict__ y, long n, float* __restrict__ part,
                                                        int* __restrict__ ticket, float inv_n, float* __restrict__ loss, float* __restrict__ dp,
                                                        float* __restrict__ tick) {
    __shared__ float red[4];
    __shared__ int last_s;
    float s = 0.f;
    const float c = 2.0f * inv_n;
    const long n4 = n / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 a = reinterpret_cast<const f32x4*>(p)[i], b = reinterpret_cast<const f32x4*>(y)[i];
        const f32x4 d = a - b;
        reinterpret_cast<f32x4*>(dp)[i] = d * c;
        s += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n - n4 * 4)) {
        const float d = p[n4 * 4 + threadIdx.x] - y[n4 * 4 + threadIdx.x];
        dp[n4 * 4 + threadIdx.x] = d * c;
        s += d * d;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int t = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = t == (int)gridDim.x - 1;
        if (last) {
            __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        last_s = last;
    }
    __syncthreads();
    if (!last_s || threadIdx.x >= 64) return;
    float tot = 0.f;
    for (int i = threadIdx.x; i < (int)gridDim.x; i += 64) tot += __builtin_nontemporal_load(part + i);
    tot = wave_sum(tot);
    if (threadIdx.x == 0) {
        loss[0] = tot * inv_n;
        if (tick) tick[0] += 1.0f;
    }
}
static int mse_blocks(int64_t n) { long b = (n / 4 + 1023) / 1024; return (int)(b < 1 ? 1 : (b > 256 ? 256 : b)); }

extern "C" int gaot_mse_loss_fwd(const float* pred, const float* target, int64_t n, float* partial, float* loss, gaot_stream_t stream) {
    GAOT_REQUIRE(pred && target && partial && loss && n > 0 && aligned16(pred) && aligned16(target), "mse_loss_fwd: bad arguments (16-byte aligned inputs, partial[256])");
    const int nb = mse_blocks(n);
    hipLaunchKernelGGL(mse_partial_kernel, dim3(nb), dim3(256), 0, ST(stream), pred, target, (long)n, partial);
    hipLaunchKernelGGL(mse_final_kernel, dim3(1), dim3(64), 0, ST(stream), partial, nb, 1.0f / (float)n, loss);
    GAOT_CHECK_LAUNCH("gaot_mse_loss_fwd");
    return GAOT_OK;
}
extern "C" int gaot_mse_loss_bwd(const float* pred, const float* target, int64_t n, const float* grad_loss, float* dpred, gaot_stream_t stream) {
    GAOT_REQUIRE(pred && target && grad_loss && dpred && n > 0 && aligned16(pred) && aligned16(target) && aligned16(dpred), "mse_loss_bwd: bad arguments");
    hipLaunchKernelGGL(mse_bwd_kernel, dim3(cap_blocks(n / 4 + 1, 256, 1024)), dim3(256), 0, ST(stream), pred, target, (long)n, grad_loss, 2.0f / (float)n, dpred);
    GAOT_CHECK_LAUNCH("gaot_mse_loss_bwd");
    return GAOT_OK;
}

extern "C" int gaot_mse_loss_fwd_bwd(const float* pred, const float* target, int64_t n, float* partial, int32_t* ticket, float* loss, float* dpred,
                                     float* tick, gaot_stream_t stream) {
    GAOT_REQUIRE(pred && target && partial && ticket && loss && dpred && n > 0 && aligned16(pred) && aligned16(target) && aligned16(dpred),
                 "mse_loss_fwd_bwd: bad arguments (16-byte aligned inputs, partial[256], a zero ticket)");
    hipLaunchKernelGGL(mse_fused_kernel, dim3(mse_blocks(n)), dim3(256), 0, ST(stream), pred, target, (long)n, partial, ticket, 1.0f / (float)n, loss,
                       dpred, tick);
    GAOT_CHECK_LAUNCH("gaot_mse_loss_fwd_bwd");
    return GAOT_OK;
}

extern "C" int gaot_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                               float eps, float weight_decay, float* step, gaot_stream_t stream) {
    GAOT_REQUIRE(p && g && m && v && step && n > 0, "adamw_step: bad arguments");
    GAOT_REQUIRE(aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v), "adamw_step: flat buffers must be 16-byte aligned");
    hipLaunchKernelGGL(adamw_tick_kernel, dim3(1), dim3(1), 0, ST(stream), step);
    hipLaunchKernelGGL(adamw_kernel, dim3(cap_blocks(n / 4 + 1, 256, 2048)), dim3(256), 0, ST(stream), p, g, m, v, (long)n, lr, beta1,
                       beta2, eps, weight_decay, step, (const float*)nullptr);
    GAOT_CHECK_LAUNCH("gaot_adamw_step");
    return GAOT_OK;
}

extern "C" int gaot_adamw_step_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper, float* step,
                                   gaot_stream_t stream) {
    GAOT_REQUIRE(p && g && m && v && step && hyper && n > 0, "adamw_step_dev: bad arguments");
    GAOT_REQUIRE(aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v), "adamw_step_dev: flat buffers must be 16-byte aligned");
    hipLaunchKernelGGL(adamw_tick_kernel, dim3(1), dim3(1), 0, ST(stream), step);
    return gaot_adamw_apply_dev(p, g, m, v, n, hyper, step, stream);
}

extern "C" int gaot_adamw_apply_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper, const float* step,
                                    gaot_stream_t stream) {
    GAOT_REQUIRE(p && g && m && v && step && hyper && n > 0, "adamw_apply_dev: bad arguments");
    GAOT_REQUIRE(aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v), "adamw_apply_dev: flat buffers must be 16-byte aligned");
    hipLaunchKernelGGL(adamw_kernel, dim3(cap_blocks(n / 4 + 1, 256, 2048)), dim3(256), 0, ST(stream), p, g, m, v, (long)n, 0.f, 0.f,
                       0.f, 0.f, 0.f, step, hyper);
    GAOT_CHECK_LAUNCH("gaot_adamw_apply_dev");
    return GAOT_OK;
}
