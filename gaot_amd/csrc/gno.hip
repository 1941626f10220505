// Graph-neural-operator kernels: geometry plan (int32 CSR, transposed CSR, edge attention, geometry
// statistics) and the gather / per-edge-weight / segment-reduce integral transform with its two backward
// products.  All of this is HBM/L2-bound index work: lanes run along the channel dim (16 B per lane,
// 256 B per feature row), segments are reduced in registers in CSR order (no atomics, deterministic),
// and workgroups of one batch sample are pinned to one XCD so the sample's feature matrix (N*C*4 B,
// 4 MiB at the 16k-node config) stays resident in that XCD's private L2.
#include "common.h"
#include "segsort.h"

namespace gaot {

// ---------------------------------------------------------------------------------------------
// plan: int64 -> int32 CSR + edge -> query map
// ---------------------------------------------------------------------------------------------
__global__ void csr_prepare_kernel(const int64_t* __restrict__ idx64, const int64_t* __restrict__ sp64, int Q, int E,
                                   int n_src, int* __restrict__ idx32, int* __restrict__ sp32,
                                   int* __restrict__ eq, int* __restrict__ flag) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid <= Q) {
        const int64_t s = sp64[gid];
        // a broken list is FLAGGED (the host raises, at once or one step later when validation is lazy); what is stored is
        // clamped so that the kernels that may run before the flag is read stay inside their arrays
        sp32[gid] = (int)(s < 0 ? 0 : (s > E ? E : s));
        if (s < 0 || s > E || (gid > 0 && sp64[gid - 1] > s) || (gid == Q && s != E) || (gid == 0 && s != 0))
            atomicOr(flag, 1);
    }
    if (gid < E) {
        const int64_t j = idx64[gid];
        if (j < 0 || j >= n_src) atomicOr(flag, 2);
        idx32[gid] = (int)(j < 0 ? 0 : (j >= n_src ? n_src - 1 : j));
        // upper_bound(splits, gid) - 1
        int lo = 0, hi = Q;  // invariant: sp[lo] <= gid < sp[hi]  (hi == Q holds since sp[Q] == E > gid); any list: 0 <= lo < Q
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (sp64[mid] <= gid) lo = mid; else hi = mid;
        }
        eq[gid] = lo;
    }
}

// ---------------------------------------------------------------------------------------------
// plan: transposed CSR (edges grouped by SOURCE node, ascending edge id inside a group)
// ---------------------------------------------------------------------------------------------
__global__ void zero_i32_kernel(int* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0;
}
__global__ void count_kernel(const int* __restrict__ idx, int E, int* __restrict__ cnt) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < E) atomicAdd(&cnt[idx[e]], 1);
}
// single-workgroup exclusive scan of cnt[0..n) -> out[0..n]; fine for a once-per-geometry pass
__global__ __launch_bounds__(1024) void exscan_kernel(const int* __restrict__ cnt, int n, int* __restrict__ out) {
    __shared__ int part[1024];
    const int t = threadIdx.x;
    const int chunk = (n + 1023) / 1024;
    const int b = t * chunk, e = min(n, b + chunk);
    int s = 0;
    for (int i = b; i < e; ++i) s += cnt[i];
    part[t] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        int v = (t >= off) ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = (t == 0) ? 0 : part[t - 1];
    for (int i = b; i < e; ++i) { out[i] = run; run += cnt[i]; }
    if (t == 1023) out[n] = part[1023];
}
__global__ void fill_kernel(const int* __restrict__ idx, int E, const int* __restrict__ tsp, int* __restrict__ cnt,
                            int* __restrict__ tedge) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < E) {
        const int j = idx[e];
        const int slot = atomicSub(&cnt[j], 1) - 1;
        tedge[tsp[j] + slot] = e;
    }
}
// concatenation of per-sample int arrays with a per-part offset (block-diagonal union of per-sample CSR plans): out = [parts[0] +
// off[0], parts[1] + off[1], ...].  Up to 64 parts per launch (kernel-argument struct: no device-side pointer table to upload).
struct ConcatParts { const int* ptr[64]; int len[64]; int off[64]; int begin[64]; int n; };
__global__ void concat_offset_kernel(ConcatParts p, int total, int* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int lo = 0, hi = p.n;                       // part of element i: begin[lo] <= i < begin[lo + 1]
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (p.begin[mid] <= i) lo = mid; else hi = mid; }
    out[i] = p.ptr[lo][i - p.begin[lo]] + p.off[lo];
}

// Block-diagonal union of per-sample plans from a DEVICE-side table (vx training under a shuffling loader: the batch composition changes
// every step, the launch does not).  All arrays are static buffers sized for E_cap >= sum of the parts' edge counts; the edges past the
// real count ("pads") are never referenced by a row of either CSR (splits[Q] = t_splits[n_src] = E_real), and the flat per-edge kernels
// that do walk them see valid indices (source 0, query 0, t_edge = own id) and an edge scale of 0.
__global__ __launch_bounds__(256) void union_compose_kernel(const gaot_union_part* __restrict__ parts, int B, int Qe, int Se, int dsrc, int ddst,
                                                            int E_cap, int* __restrict__ index, int* __restrict__ eq, int* __restrict__ tedge,
                                                            int* __restrict__ splits, int* __restrict__ tsplits, float* __restrict__ src,
                                                            float* __restrict__ dst, int* __restrict__ e_real) {
    __shared__ int begin[1025];
    for (int b = threadIdx.x; b < B; b += 256) begin[b] = parts[b].e_begin;
    if (threadIdx.x == 0) begin[B] = parts[B - 1].e_begin + parts[B - 1].e_count;
    __syncthreads();
    const int E = min(begin[B], E_cap);
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i == 0) *e_real = E;
    if (i < E_cap) {
        if (i < E) {
            int lo = 0, hi = B;                       // begin[lo] <= i < begin[hi]
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (begin[mid] <= (int)i) lo = mid; else hi = mid; }
            const gaot_union_part& p = parts[lo];
            const int l = (int)i - begin[lo];
            index[i] = p.index[l] + lo * Se;
            eq[i] = p.edge_query[l] + lo * Qe;
            tedge[i] = p.t_edge[l] + begin[lo];
        } else {
            index[i] = 0; eq[i] = 0; tedge[i] = (int)i;
        }
    }
    const long Qt = (long)B * Qe, St = (long)B * Se;
    if (i <= Qt) splits[i] = i == Qt ? E : min(parts[i / Qe].splits[i % Qe] + begin[i / Qe], E);
    if (i <= St) tsplits[i] = i == St ? E : min(parts[i / Se].t_splits[i % Se] + begin[i / Se], E);
    if (src && i < St * dsrc) { const long r = i / dsrc; src[i] = parts[r / Se].src[(r % Se) * dsrc + i % dsrc]; }
    if (dst && i < Qt * ddst) { const long r = i / ddst; dst[i] = parts[r / Qe].dst[(r % Qe) * ddst + i % ddst]; }
}
// 1 / deg(query(e)) per edge, 0 on the pads ('mean' reduction of agno.py:264 as a per-edge scale)
__global__ void edge_inv_degree_kernel(const int* __restrict__ sp, const int* __restrict__ eq, int E, const int* __restrict__ e_real,
                                       float* __restrict__ out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    if (e_real && e >= *e_real) { out[e] = 0.f; return; }
    const int q = eq[e];
    out[e] = 1.0f / (float)max(sp[q + 1] - sp[q], 1);
}
// a[b, e, :] = 0 for e >= *e_real (the pads of a padded union: whatever per-edge scale the transform uses must vanish there, and so must
// per-edge gradient rows that a flat kernel filled before a reduction over all E rows reads them)
__global__ void edge_zero_pads_kernel(float* __restrict__ a, int B, int E, int width, const int* __restrict__ e_real) {
    const int er = *e_real;
    const long n = (long)(E - er) * width;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    a[((long)blockIdx.y * E + er) * width + i] = 0.f;
}

// ---------------------------------------------------------------------------------------------
// plan: cosine attention + segment softmax (agno.py:112-146,218-224).  One thread per query.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float cos_score(const float* __restrict__ x, const float* __restrict__ y, int dim, float xinv) {
    float yy = 0.f, xy = 0.f;
    for (int d = 0; d < dim; ++d) { const float v = y[d]; yy += v * v; }
    const float yinv = 1.0f / fmaxf(sqrtf(yy), 1e-12f);   // F.normalize: v / max(||v||, eps)
    for (int d = 0; d < dim; ++d) xy += (x[d] * xinv) * (y[d] * yinv);
    return xy;
}
// GEO_LANES lanes per query row (aligned lane groups of one wave): with a vx batch under a shuffling loader this runs every step
// (plan.StaticUnion), not once per geometry.  A lane strides its row's edges, the group reduces by xor shuffles inside the group.  Rows longer
// than GEO_LONG edges (the latent tokens next to an airfoil contour reach 350+, beside thousands of empty rows) are taken by the WHOLE WAVE, one
// after the other: 64 lanes stride the row, wave-wide reductions.  How a row is summed depends on its own length only -- not on the launch, not
// on its neighbours -- so a row of a block-diagonal union and the same row of the sample alone give the same bits.
#define GEO_LANES 8
#define GEO_LONG 64
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
    for (int off = GEO_LANES >> 1; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int off = GEO_LANES >> 1; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double group_sum_d(double v) {
#pragma unroll
    for (int off = GEO_LANES >> 1; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
// softmax of the cosine scores of row q = edges [b, e), walked by `n` lanes (this lane is number l of them); rmax / rsum reduce over those lanes.
// Up to 8 edges per lane stay in registers (one store per edge); longer rows go through `attn` itself.
template <typename RMAX, typename RSUM>
__device__ __forceinline__ void cosine_row(const float* __restrict__ src, const float* __restrict__ qry, int dim, const int* __restrict__ idx,
                                           float* __restrict__ attn, int q, int b, int e, int l, int n, RMAX rmax, RSUM rsum) {
    const float* x = qry + (long)q * dim;
    float xx = 0.f;
    for (int d = 0; d < dim; ++d) xx += x[d] * x[d];
    const float xinv = 1.0f / fmaxf(sqrtf(xx), 1e-12f);
    float mx = -INFINITY;
    if (e - b <= 8 * n) {
        float sv[8];
        int j[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { const int t = b + l + i * n; j[i] = t < e ? idx[t] : 0; }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int t = b + l + i * n;
            sv[i] = t < e ? cos_score(x, src + (long)j[i] * dim, dim, xinv) : -INFINITY;
            mx = fmaxf(mx, sv[i]);
        }
        mx = rmax(mx);
        float den = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { sv[i] = b + l + i * n < e ? expf(sv[i] - mx) : 0.f; den += sv[i]; }
        den = rsum(den);
#pragma unroll
        for (int i = 0; i < 8; ++i) { const int t = b + l + i * n; if (t < e) attn[t] = sv[i] / den; }
        return;
    }
    for (int t = b + l; t < e; t += n) {
        const float s = cos_score(x, src + (long)idx[t] * dim, dim, xinv);
        attn[t] = s;
        mx = fmaxf(mx, s);
    }
    mx = rmax(mx);
    float den = 0.f;
    for (int t = b + l; t < e; t += n) { const float v = expf(attn[t] - mx); attn[t] = v; den += v; }
    den = rsum(den);
    for (int t = b + l; t < e; t += n) attn[t] = attn[t] / den;
}
__global__ __launch_bounds__(256) void edge_attention_cosine_kernel(const float* __restrict__ src, const float* __restrict__ qry, int dim,
                                                                    const int* __restrict__ idx, const int* __restrict__ sp, int Q,
                                                                    float* __restrict__ attn, const int* __restrict__ guard) {
    if (guard && *guard == 0) return;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int q = gid / GEO_LANES, l = gid % GEO_LANES;
    const int b = q < Q ? sp[q] : 0, e = q < Q ? sp[q + 1] : 0;
    const bool is_long = e - b > GEO_LONG;
    if (e > b && !is_long)
        cosine_row(src, qry, dim, idx, attn, q, b, e, l, GEO_LANES, [](float v) { return group_max(v); }, [](float v) { return group_sum(v); });
    unsigned long long todo = __ballot(is_long && l == 0);          // one bit per long row of this wave (the first lane of its group)
    while (todo) {
        const int first = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int qq = __shfl(q, first, 64), bb = __shfl(b, first, 64), ee = __shfl(e, first, 64);
        cosine_row(src, qry, dim, idx, attn, qq, bb, ee, lane, 64, [](float v) { return wave_max(v); }, [](float v) { return wave_sum(v); });
    }
}
__global__ void segment_softmax_fwd_kernel(const float* __restrict__ score, const int* __restrict__ sp, int Q,
                                           float* __restrict__ attn) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= Q) return;
    const int b = sp[q], e = sp[q + 1];
    if (b == e) return;
    float mx = -INFINITY;
    for (int t = b; t < e; ++t) mx = fmaxf(mx, score[t]);
    float den = 0.f;
    for (int t = b; t < e; ++t) { const float v = expf(score[t] - mx); attn[t] = v; den += v; }
    for (int t = b; t < e; ++t) attn[t] = attn[t] / den;
}
__global__ void segment_softmax_bwd_kernel(const float* __restrict__ attn, const float* __restrict__ dattn,
                                           const int* __restrict__ sp, int Q, float* __restrict__ dscore) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= Q) return;
    const int b = sp[q], e = sp[q + 1];
    float dot = 0.f;
    for (int t = b; t < e; ++t) dot += attn[t] * dattn[t];
    for (int t = b; t < e; ++t) dscore[t] = attn[t] * (dattn[t] - dot);
}

__global__ void edge_features_kernel(const float* __restrict__ src, const float* __restrict__ qry, int dim,
                                     const int* __restrict__ idx, const int* __restrict__ eq, int E,
                                     float* __restrict__ feat, const int* __restrict__ guard) {
    if (guard && *guard == 0) return;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int w = 2 * dim;
    if (gid >= (long)E * w) return;
    const int e = (int)(gid / w), c = (int)(gid % w);
    feat[gid] = (c < dim) ? src[(long)idx[e] * dim + c] : qry[(long)eq[e] * dim + (c - dim)];
}

// ---------------------------------------------------------------------------------------------
// plan: geometry statistics (gemb.py:83-171).  fp64 inside (once per geometry), fp32 out.
// ---------------------------------------------------------------------------------------------
template <int DIM>
__device__ void sym_eig_desc(const double (&c)[DIM][DIM], double (&ev)[DIM]);

template <>
__device__ void sym_eig_desc<2>(const double (&c)[2][2], double (&ev)[2]) {
    const double tr = 0.5 * (c[0][0] + c[1][1]);
    const double df = 0.5 * (c[0][0] - c[1][1]);
    const double rad = sqrt(df * df + c[0][1] * c[0][1]);
    ev[0] = tr + rad;
    ev[1] = tr - rad;
}
template <>
__device__ void sym_eig_desc<3>(const double (&c)[3][3], double (&ev)[3]) {
    double a[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) a[i][j] = c[i][j];
    for (int sweep = 0; sweep < 12; ++sweep) {           // cyclic Jacobi
        const double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
        const double dg = a[0][0] * a[0][0] + a[1][1] * a[1][1] + a[2][2] * a[2][2];
        if (off <= 1e-40 * (dg + 1e-300) || off == 0.0) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (a[p][q] == 0.0) continue;
                const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
                for (int k = 0; k < 3; ++k) {            // A <- A J
                    const double akp = a[k][p], akq = a[k][q];
                    a[k][p] = cs * akp - sn * akq;
                    a[k][q] = sn * akp + cs * akq;
                }
                for (int k = 0; k < 3; ++k) {            // A <- J^T A
                    const double apk = a[p][k], aqk = a[q][k];
                    a[p][k] = cs * apk - sn * aqk;
                    a[q][k] = sn * apk + cs * aqk;
                }
            }
    }
    double e0 = a[0][0], e1 = a[1][1], e2 = a[2][2], t;
    if (e0 < e1) { t = e0; e0 = e1; e1 = t; }
    if (e1 < e2) { t = e1; e1 = e2; e2 = t; }
    if (e0 < e1) { t = e0; e0 = e1; e1 = t; }
    ev[0] = e0; ev[1] = e1; ev[2] = e2;
}

// raw statistics of row q = edges [b, e), walked by `n` lanes (this lane is number l of them); rsum reduces a double over those lanes; the lane with
// l == 0 writes the row
template <int DIM, typename RSUM>
__device__ __forceinline__ void geo_stats_row(const float* __restrict__ geom, const float* __restrict__ qry, const int* __restrict__ idx,
                                              float* __restrict__ raw, int q, int b, int e, int l, int n, RSUM rsum) {
    constexpr int F = 3 + 2 * DIM;
    float* o = raw + (long)q * F;
    const int cnt = e - b;
    double x[DIM], cen[DIM];
    for (int d = 0; d < DIM; ++d) { x[d] = qry[(long)q * DIM + d]; cen[d] = 0.0; }
    double sd = 0.0, sd2 = 0.0;
    for (int t = b + l; t < e; t += n) {
        const float* y = geom + (long)idx[t] * DIM;
        double d2 = 0.0;
        for (int d = 0; d < DIM; ++d) { const double dv = (double)y[d] - x[d]; d2 += dv * dv; cen[d] += y[d]; }
        sd += sqrt(d2);
        sd2 += d2;
    }
    sd = rsum(sd);
    sd2 = rsum(sd2);
    const double inv = 1.0 / cnt;
    for (int d = 0; d < DIM; ++d) cen[d] = rsum(cen[d]) * inv;
    double cov[DIM][DIM];
    for (int i = 0; i < DIM; ++i) for (int j = 0; j < DIM; ++j) cov[i][j] = 0.0;
    for (int t = b + l; t < e; t += n) {
        const float* y = geom + (long)idx[t] * DIM;
        double c[DIM];
        for (int d = 0; d < DIM; ++d) c[d] = (double)y[d] - cen[d];
        for (int i = 0; i < DIM; ++i) for (int j = i; j < DIM; ++j) cov[i][j] += c[i] * c[j];
    }
    for (int i = 0; i < DIM; ++i) for (int j = i; j < DIM; ++j) { cov[i][j] = rsum(cov[i][j]) * inv; cov[j][i] = cov[i][j]; }
    if (l != 0) return;
    double ev[DIM];
    sym_eig_desc<DIM>(cov, ev);
    const double mean = sd * inv;
    double var = sd2 * inv - mean * mean;
    if (var < 0.0) var = 0.0;
    o[0] = (float)cnt; o[1] = (float)mean; o[2] = (float)var;
    for (int d = 0; d < DIM; ++d) { o[3 + d] = (float)(cen[d] - x[d]); o[3 + DIM + d] = (float)ev[d]; }
}
template <int DIM>
__global__ __launch_bounds__(256) void geo_stats_raw_kernel(const float* __restrict__ geom, const float* __restrict__ qry,
                                                            const int* __restrict__ idx, const int* __restrict__ sp, int Q,
                                                            float* __restrict__ raw, const int* __restrict__ guard) {
    constexpr int F = 3 + 2 * DIM;
    if (guard && *guard == 0) return;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int q = gid / GEO_LANES, l = gid % GEO_LANES;
    const int b = q < Q ? sp[q] : 0, e = q < Q ? sp[q + 1] : 0;
    const bool is_long = e - b > GEO_LONG;
    if (q < Q && e == b && l == 0) for (int f = 0; f < F; ++f) raw[(long)q * F + f] = 0.f;
    if (e > b && !is_long) geo_stats_row<DIM>(geom, qry, idx, raw, q, b, e, l, GEO_LANES, [](double v) { return group_sum_d(v); });
    unsigned long long todo = __ballot(is_long && l == 0);
    while (todo) {
        const int first = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int qq = __shfl(q, first, 64), bb = __shfl(b, first, 64), ee = __shfl(e, first, 64);
        geo_stats_row<DIM>(geom, qry, idx, raw, qq, bb, ee, lane, 64, [](double v) { return wave_sum_d(v); });
    }
}
// column statistics, deterministic and parallel: pass 0 = sum x, pass 1 = sum (x - mean)^2, every workgroup writes ITS fp64 partial sums to
// part[group][pass][block][f] (no atomics); whoever needs a total adds the partials in block order.  Rows come in `groups` equal groups of Q
// rows (vx mode: one group per sample of the block-diagonal union), each standardised on its own (gemb.py:164-169 runs per sample there):
// blockIdx.y = group.  GEO_NB_MAX workgroups per group at most (the scratch is sized for that).
#define GEO_NB_MAX 16
#define GEO_F_MAX 9
__global__ __launch_bounds__(256) void geo_colstat_kernel(const float* __restrict__ raw, int Q, int F, int pass,
                                                          double* __restrict__ part_all, const int* __restrict__ guard) {
    __shared__ double red[4][GEO_F_MAX];
    if (guard && *guard == 0) return;
    const int nb = gridDim.x;
    raw += (long)blockIdx.y * Q * F;
    double* part = part_all + (long)blockIdx.y * 2 * GEO_NB_MAX * F;
    double mean[GEO_F_MAX], s[GEO_F_MAX];
    for (int f = 0; f < GEO_F_MAX; ++f) { mean[f] = 0.0; s[f] = 0.0; }
    if (pass) {
        for (int f = 0; f < F; ++f) {
            double t = 0.0;
            for (int b = 0; b < nb; ++b) t += part[(long)b * F + f];
            mean[f] = t / Q;
        }
    }
    for (int q = blockIdx.x * 256 + threadIdx.x; q < Q; q += nb * 256) {
#pragma unroll
        for (int f = 0; f < GEO_F_MAX; ++f) {
            if (f < F) {
                const double v = raw[(long)q * F + f];
                s[f] += pass ? (v - mean[f]) * (v - mean[f]) : v;
            }
        }
    }
#pragma unroll
    for (int f = 0; f < GEO_F_MAX; ++f) {
        if (f < F) {
            const double w = wave_sum_d(s[f]);
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][f] = w;
        }
    }
    __syncthreads();
    if (threadIdx.x < F) {
        const int f = threadIdx.x;
        part[((long)pass * GEO_NB_MAX + blockIdx.x) * F + f] = (red[0][f] + red[1][f]) + (red[2][f] + red[3][f]);
    }
}
// fin[group][0][f] = mean, fin[group][1][f] = the divisor (torch.std: unbiased; cast to fp32 before the 1e-6 test like the reference's
// fp32 tensor, gemb.py:165-166)
__global__ void geo_colstat_final_kernel(const double* __restrict__ part_all, double* __restrict__ fin_all, int Q, int F, int nb,
                                         const int* __restrict__ guard) {
    if (guard && *guard == 0) return;
    const int f = threadIdx.x;
    if (f >= F) return;
    const double* part = part_all + (long)blockIdx.x * 2 * GEO_NB_MAX * F;
    double m = 0.0, v = 0.0;
    for (int b = 0; b < nb; ++b) m += part[(long)b * F + f];
    for (int b = 0; b < nb; ++b) v += part[((long)GEO_NB_MAX + b) * F + f];
    float sd = (float)sqrt(v / (double)(Q - 1));
    if (sd < 1e-6f) sd = 1.0f;
    fin_all[(long)blockIdx.x * 2 * F + f] = m / Q;
    fin_all[(long)blockIdx.x * 2 * F + F + f] = (double)sd;
}
__global__ void geo_standardise_kernel(float* __restrict__ raw, int Q, int F, const double* __restrict__ fin_all,
                                       const int* __restrict__ guard, int groups) {
    if (guard && *guard == 0) return;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long)groups * Q * F) return;
    const int f = (int)(gid % F);
    const double* fin = fin_all + (gid / ((long)Q * F)) * 2 * F;
    raw[gid] = (float)(((double)raw[gid] - fin[f]) / fin[F + f]);
}

// ---------------------------------------------------------------------------------------------
// content guard of the geometry caches: a caller that uploads the coordinates anew every step (the reference trainer,
// static_trainer.py:167-170) hands over NEW tensors with the OLD bytes.  The plan keeps a copy of the bytes its cached arrays
// were computed from; guard_compare raises *flag when the new bytes differ, guard_update refreshes the copy, and the plan
// kernels above take the flag as `guard`: they return at once when it is 0.  Nothing is read back by the host.
// ---------------------------------------------------------------------------------------------
__global__ void guard_compare_kernel(const uint32_t* __restrict__ cur, const uint32_t* __restrict__ kept, long nwords,
                                     int* __restrict__ flag) {
    bool diff = false;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (long)gridDim.x * blockDim.x)
        diff |= cur[i] != kept[i];
    if (__any(diff) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}
// compare two (current, kept) pairs and take the new bytes into the kept copies in the same pass: flag |= any word differs
__global__ void guard_sync2_kernel(const uint32_t* __restrict__ c0, uint32_t* __restrict__ k0, long n0,
                                   const uint32_t* __restrict__ c1, uint32_t* __restrict__ k1, long n1, int* __restrict__ flag) {
    bool diff = false;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n0 + n1; i += (long)gridDim.x * blockDim.x) {
        const uint32_t v = i < n0 ? c0[i] : c1[i - n0];
        uint32_t* dst = i < n0 ? k0 + i : k1 + (i - n0);
        diff |= v != *dst;
        *dst = v;
    }
    if (__any(diff) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}
__global__ void guard_update_kernel(const uint32_t* __restrict__ cur, uint32_t* __restrict__ kept, long nwords,
                                    const int* __restrict__ flag) {
    if (*flag == 0) return;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (long)gridDim.x * blockDim.x) kept[i] = cur[i];
}

// ---------------------------------------------------------------------------------------------
// integral transform: out[b,r,:] = sum_t escale[edge(t)] * w[edge(t),:] * src[b, col(t), :]
// one thread per (r, b, channel-vector); workgroup id -> (row block, b) with b = id % B so that with
// B == 8 sample b lives on XCD b (dispatcher places workgroup id on XCD id % 8).
// ---------------------------------------------------------------------------------------------
template <int VW> struct Vec;
template <> struct Vec<4> { typedef f32x4 T; };
template <> struct Vec<2> { typedef float T __attribute__((ext_vector_type(2))); };
template <> struct Vec<1> { typedef float T; };

template <int VW> __device__ __forceinline__ typename Vec<VW>::T vzero();
template <> __device__ __forceinline__ f32x4 vzero<4>() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
template <> __device__ __forceinline__ Vec<2>::T vzero<2>() { return Vec<2>::T{0.f, 0.f}; }
template <> __device__ __forceinline__ float vzero<1>() { return 0.f; }

template <int VW, bool HAS_W, bool HAS_MAP>
__global__ __launch_bounds__(256) void gather_reduce_kernel(const float* __restrict__ w, const float* __restrict__ src,
                                                            int B, int n_src_rows, int C,
                                                            const int* __restrict__ sp, const int* __restrict__ cols,
                                                            const int* __restrict__ emap, int n_out,
                                                            const float* __restrict__ escale, float* __restrict__ out,
                                                            int lanes_per_row, int rows_per_block) {
    typedef typename Vec<VW>::T V;
    const int b = blockIdx.x % B;
    const int rblk = blockIdx.x / B;
    const int r = rblk * rows_per_block + threadIdx.x / lanes_per_row;
    const int c = (threadIdx.x % lanes_per_row) * VW;
    if (r >= n_out || c >= C) return;
    const int t0 = sp[r], t1 = sp[r + 1];
    const float* sb = src + (long)b * n_src_rows * C + c;
    V acc = vzero<VW>();
    // four edges in flight: (edge map ->) column index -> feature row is a dependent load chain and segments are short
    for (int t = t0; t < t1; t += 4) {
        int e[4], j[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int tt = min(t + u, t1 - 1); e[u] = HAS_MAP ? emap[tt] : tt; }
#pragma unroll
        for (int u = 0; u < 4; ++u) j[u] = cols[e[u]];
        V sv[4], wv[4];
        float a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            sv[u] = *reinterpret_cast<const V*>(sb + (long)j[u] * C);
            if (HAS_W) wv[u] = *reinterpret_cast<const V*>(w + (long)e[u] * C + c);
            a[u] = (t + u < t1) ? (escale ? escale[e[u]] : 1.0f) : 0.0f;            // padding edges weigh nothing
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            V sc = sv[u] * a[u];
            if (HAS_W) acc += wv[u] * sc; else acc += sc;
        }
    }
    *reinterpret_cast<V*>(out + ((long)b * n_out + r) * C + c) = acc;
}

// dW[e,:] = escale[e] * sum_b dOut[b, eq[e], :] * f[b, idx[e], :]
template <int VW>
__global__ __launch_bounds__(256) void edge_grad_kernel(const float* __restrict__ dout, const float* __restrict__ f,
                                                        int B, int Q, int n_src, int C, const int* __restrict__ idx,
                                                        const int* __restrict__ eq, int E,
                                                        const float* __restrict__ escale, float* __restrict__ dw,
                                                        int lanes_per_row, int rows_per_block) {
    typedef typename Vec<VW>::T V;
    const int e = blockIdx.x * rows_per_block + threadIdx.x / lanes_per_row;
    const int c = (threadIdx.x % lanes_per_row) * VW;
    if (e >= E || c >= C) return;
    const int q = eq[e], j = idx[e];
    V acc = vzero<VW>();
    for (int b = 0; b < B; ++b) {
        const V g = *reinterpret_cast<const V*>(dout + ((long)b * Q + q) * C + c);
        const V v = *reinterpret_cast<const V*>(f + ((long)b * n_src + j) * C + c);
        acc += g * v;
    }
    if (escale) acc *= escale[e];
    *reinterpret_cast<V*>(dw + (long)e * C + c) = acc;
}

// out[b,q,:] = rowscale[q] * sum_{e in seg(q)} x[b,e,:]
template <int VW>
__global__ __launch_bounds__(256) void segment_sum_kernel(const float* __restrict__ x, int B, int E, int C,
                                                          const int* __restrict__ sp, int Q,
                                                          const float* __restrict__ rowscale, float* __restrict__ out,
                                                          int lanes_per_row, int rows_per_block) {
    typedef typename Vec<VW>::T V;
    const int b = blockIdx.x % B;
    const int q = (blockIdx.x / B) * rows_per_block + threadIdx.x / lanes_per_row;
    const int c = (threadIdx.x % lanes_per_row) * VW;
    if (q >= Q || c >= C) return;
    V acc = vzero<VW>();
    const float* xb = x + (long)b * E * C + c;
    for (int t = sp[q]; t < sp[q + 1]; ++t) acc += *reinterpret_cast<const V*>(xb + (long)t * C);
    if (rowscale) acc *= rowscale[q];
    *reinterpret_cast<V*>(out + ((long)b * Q + q) * C + c) = acc;
}

static inline int pow2_ceil(int v) { int p = 1; while (p < v) p <<= 1; return p; }
static inline int pick_vw(int C, const void* a, const void* b, const void* c) {
    if (C % 4 == 0 && aligned16(a) && aligned16(b) && aligned16(c)) return 4;
    if (C % 2 == 0 && ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c)) & 7u) == 0) return 2;
    return 1;
}

}  // namespace gaot

using namespace gaot;
#define ST(s) reinterpret_cast<hipStream_t>(s)

extern "C" int gaot_csr_prepare(const int64_t* index_i64, const int64_t* splits_i64, int32_t Q, int32_t E, int32_t n_src,
                                int32_t* index32, int32_t* splits32, int32_t* edge_query, int32_t* status_flag,
                                gaot_stream_t stream) {
    GAOT_REQUIRE(splits_i64 && splits32 && status_flag, "csr_prepare: null pointer");
    GAOT_REQUIRE(Q >= 0 && E >= 0 && n_src >= 0, "csr_prepare: negative size");
    GAOT_REQUIRE(E == 0 || (index_i64 && index32 && edge_query), "csr_prepare: null index pointer with E > 0");
    const int n = (E > Q + 1) ? E : Q + 1;
    hipLaunchKernelGGL(zero_i32_kernel, dim3(1), dim3(64), 0, ST(stream), status_flag, 1);
    hipLaunchKernelGGL(csr_prepare_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ST(stream), index_i64, splits_i64, Q, E,
                       n_src, index32, splits32, edge_query, status_flag);
    GAOT_CHECK_LAUNCH("gaot_csr_prepare");
    return GAOT_OK;
}

extern "C" int gaot_csr_transpose(const int32_t* index32, int32_t E, int32_t n_src, int32_t* t_splits, int32_t* t_edge,
                                  int32_t* scratch, gaot_stream_t stream) {
    GAOT_REQUIRE(t_splits && scratch && n_src > 0 && E >= 0, "csr_transpose: bad arguments (scratch: n_src + 1 + E int32)");
    hipLaunchKernelGGL(zero_i32_kernel, dim3(cdiv(n_src + 1, 256)), dim3(256), 0, ST(stream), scratch, n_src + 1);
    if (E > 0) hipLaunchKernelGGL(count_kernel, dim3(cdiv(E, 256)), dim3(256), 0, ST(stream), index32, E, scratch);
    hipLaunchKernelGGL(exscan_kernel, dim3(1), dim3(1024), 0, ST(stream), scratch, n_src, t_splits);
    if (E > 0) {
        hipLaunchKernelGGL(fill_kernel, dim3(cdiv(E, 256)), dim3(256), 0, ST(stream), index32, E, t_splits, scratch, t_edge);
        sort_segments<int, int>(t_splits, n_src, t_edge, scratch + n_src + 1, ST(stream));     // ascending edge ids: deterministic backward
    }
    GAOT_CHECK_LAUNCH("gaot_csr_transpose");
    return GAOT_OK;
}

extern "C" int gaot_edge_attention_cosine(const float* src, const float* qry, int32_t dim, const int32_t* index32,
                                          const int32_t* splits32, int32_t Q, float* attn, const int32_t* guard,
                                          gaot_stream_t stream) {
    GAOT_REQUIRE(src && qry && splits32 && dim > 0 && Q >= 0, "edge_attention_cosine: bad arguments");
    if (Q == 0) return GAOT_OK;
    hipLaunchKernelGGL(edge_attention_cosine_kernel, dim3(cdiv((long)Q * GEO_LANES, 256)), dim3(256), 0, ST(stream), src, qry, dim,
                       index32, splits32, Q, attn, guard);
    GAOT_CHECK_LAUNCH("gaot_edge_attention_cosine");
    return GAOT_OK;
}

extern "C" int gaot_segment_softmax_fwd(const float* score, const int32_t* splits32, int32_t Q, float* attn,
                                        gaot_stream_t stream) {
    GAOT_REQUIRE(splits32 && Q >= 0, "segment_softmax_fwd: bad arguments");
    if (Q == 0) return GAOT_OK;
    hipLaunchKernelGGL(segment_softmax_fwd_kernel, dim3(cdiv(Q, 128)), dim3(128), 0, ST(stream), score, splits32, Q, attn);
    GAOT_CHECK_LAUNCH("gaot_segment_softmax_fwd");
    return GAOT_OK;
}

extern "C" int gaot_segment_softmax_bwd(const float* attn, const float* dattn, const int32_t* splits32, int32_t Q,
                                        float* dscore, gaot_stream_t stream) {
    GAOT_REQUIRE(splits32 && Q >= 0, "segment_softmax_bwd: bad arguments");
    if (Q == 0) return GAOT_OK;
    hipLaunchKernelGGL(segment_softmax_bwd_kernel, dim3(cdiv(Q, 128)), dim3(128), 0, ST(stream), attn, dattn, splits32, Q,
                       dscore);
    GAOT_CHECK_LAUNCH("gaot_segment_softmax_bwd");
    return GAOT_OK;
}

extern "C" int gaot_edge_features(const float* src, const float* qry, int32_t dim, const int32_t* index32,
                                  const int32_t* edge_query, int32_t E, float* feat, const int32_t* guard, gaot_stream_t stream) {
    GAOT_REQUIRE(dim > 0 && E >= 0, "edge_features: bad arguments");
    if (E == 0) return GAOT_OK;
    GAOT_REQUIRE(src && qry && index32 && edge_query && feat, "edge_features: null pointer");
    hipLaunchKernelGGL(edge_features_kernel, dim3(cdiv((long)E * 2 * dim, 256)), dim3(256), 0, ST(stream), src, qry, dim,
                       index32, edge_query, E, feat, guard);
    GAOT_CHECK_LAUNCH("gaot_edge_features");
    return GAOT_OK;
}

extern "C" int gaot_geo_stats(const float* geom, const float* qry, int32_t dim, const int32_t* index32,
                              const int32_t* splits32, int32_t Q, float* stats, double* scratch, const int32_t* guard,
                              int32_t groups, gaot_stream_t stream) {
    GAOT_REQUIRE(dim == 2 || dim == 3, "geo_stats: coord dim must be 2 or 3 (got %d)", dim);
    GAOT_REQUIRE(geom && qry && splits32 && stats && scratch && Q > 0 && groups >= 1 && Q % groups == 0,
                 "geo_stats: bad arguments (Q must be a multiple of groups)");
    const int F = 3 + 2 * dim;
    const int Qg = Q / groups;
    if (dim == 2)
        hipLaunchKernelGGL(geo_stats_raw_kernel<2>, dim3(cdiv((long)Q * GEO_LANES, 256)), dim3(256), 0, ST(stream), geom, qry, index32, splits32, Q, stats, guard);
    else
        hipLaunchKernelGGL(geo_stats_raw_kernel<3>, dim3(cdiv((long)Q * GEO_LANES, 256)), dim3(256), 0, ST(stream), geom, qry, index32, splits32, Q, stats, guard);
    // Column sums in a FIXED order, whatever the number of workgroups: every workgroup writes its fp64 partial sums, the consumers add them in
    // block order.  (Partial sums that met in an atomicAdd in arrival order made a standardised statistic come out one fp32 ulp different from
    // run to run once in a while -- enough for two trainings from the same seed to end ~1e-2 apart in loss after 100 steps, tools/det_batch.py;
    // ONE workgroup per group, the first cure, is serial over 1e5..1e6 query rows and runs every step for the unions of a vx batch.)
    double* fin = scratch + (long)groups * 2 * GEO_NB_MAX * F;
    int nb = cdiv(Qg, 2048);
    nb = nb < 1 ? 1 : (nb > GEO_NB_MAX ? GEO_NB_MAX : nb);
    hipLaunchKernelGGL(geo_colstat_kernel, dim3(nb, groups), dim3(256), 0, ST(stream), stats, Qg, F, 0, scratch, guard);
    hipLaunchKernelGGL(geo_colstat_kernel, dim3(nb, groups), dim3(256), 0, ST(stream), stats, Qg, F, 1, scratch, guard);
    hipLaunchKernelGGL(geo_colstat_final_kernel, dim3(groups), dim3(64), 0, ST(stream), scratch, fin, Qg, F, nb, guard);
    hipLaunchKernelGGL(geo_standardise_kernel, dim3(cdiv((long)Q * F, 256)), dim3(256), 0, ST(stream), stats, Qg, F, fin, guard, groups);
    GAOT_CHECK_LAUNCH("gaot_geo_stats");
    return GAOT_OK;
}

extern "C" int gaot_concat_offset(const int32_t* const* parts, const int32_t* lens, const int32_t* offsets, int32_t n_parts, int32_t* out,
                                  gaot_stream_t stream) {
    GAOT_REQUIRE(parts && lens && offsets && out && n_parts >= 1, "concat_offset: bad arguments");
    long done = 0;
    for (int base = 0; base < n_parts; base += 64) {
        ConcatParts p;
        p.n = n_parts - base < 64 ? n_parts - base : 64;
        int total = 0;
        for (int i = 0; i < 64; ++i) {
            const bool live = i < p.n;
            p.ptr[i] = live ? parts[base + i] : nullptr;
            p.len[i] = live ? lens[base + i] : 0;
            p.off[i] = live ? offsets[base + i] : 0;
            p.begin[i] = total;
            GAOT_REQUIRE(!live || p.len[i] == 0 || p.ptr[i], "concat_offset: null part %d", base + i);
            total += p.len[i];
        }
        if (total > 0) hipLaunchKernelGGL(concat_offset_kernel, dim3(cdiv(total, 256)), dim3(256), 0, ST(stream), p, total, out + done);
        done += total;
    }
    GAOT_CHECK_LAUNCH("gaot_concat_offset");
    return GAOT_OK;
}

extern "C" int gaot_union_compose(const gaot_union_part* parts_dev, int32_t n_parts, int32_t q_each, int32_t n_src_each, int32_t dim_src,
                                  int32_t dim_dst, int32_t e_cap, int32_t* index32, int32_t* edge_query, int32_t* t_edge, int32_t* splits32,
                                  int32_t* t_splits, float* src, float* dst, int32_t* e_real, gaot_stream_t stream) {
    GAOT_REQUIRE(parts_dev && n_parts >= 1 && n_parts <= 1024 && q_each > 0 && n_src_each > 0 && e_cap >= 1 && dim_src >= 0 && dim_dst >= 0,
                 "union_compose: bad sizes (1 <= n_parts <= 1024, q_each, n_src_each, e_cap > 0)");
    GAOT_REQUIRE(index32 && edge_query && t_edge && splits32 && t_splits && e_real, "union_compose: null output pointer");
    long n = e_cap;
    const long Qt = (long)n_parts * q_each, St = (long)n_parts * n_src_each;
    if (Qt + 1 > n) n = Qt + 1;
    if (St + 1 > n) n = St + 1;
    if (src && St * dim_src > n) n = St * dim_src;
    if (dst && Qt * dim_dst > n) n = Qt * dim_dst;
    hipLaunchKernelGGL(union_compose_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ST(stream), parts_dev, n_parts, q_each, n_src_each, dim_src,
                       dim_dst, e_cap, index32, edge_query, t_edge, splits32, t_splits, src, dst, e_real);
    GAOT_CHECK_LAUNCH("gaot_union_compose");
    return GAOT_OK;
}

extern "C" int gaot_edge_inv_degree(const int32_t* splits32, const int32_t* edge_query, int32_t E, const int32_t* e_real, float* out,
                                    gaot_stream_t stream) {
    GAOT_REQUIRE(E >= 0, "edge_inv_degree: negative size");
    if (E == 0) return GAOT_OK;
    GAOT_REQUIRE(splits32 && edge_query && out, "edge_inv_degree: null pointer");
    hipLaunchKernelGGL(edge_inv_degree_kernel, dim3(cdiv(E, 256)), dim3(256), 0, ST(stream), splits32, edge_query, E, e_real, out);
    GAOT_CHECK_LAUNCH("gaot_edge_inv_degree");
    return GAOT_OK;
}

extern "C" int gaot_edge_zero_pads(float* a, int32_t B, int32_t E, int32_t width, const int32_t* e_real, int32_t max_pads, gaot_stream_t stream) {
    GAOT_REQUIRE(B >= 1 && E >= 0 && width >= 1 && max_pads >= 0, "edge_zero_pads: bad sizes");
    if (E == 0 || max_pads == 0) return GAOT_OK;
    GAOT_REQUIRE(a && e_real, "edge_zero_pads: null pointer");
    if (max_pads > E) max_pads = E;
    hipLaunchKernelGGL(edge_zero_pads_kernel, dim3(cdiv((long)max_pads * width, 256), B), dim3(256), 0, ST(stream), a, B, E, width, e_real);
    GAOT_CHECK_LAUNCH("gaot_edge_zero_pads");
    return GAOT_OK;
}

extern "C" int gaot_guard_begin(int32_t* flag, gaot_stream_t stream) {
    GAOT_REQUIRE(flag != nullptr, "guard_begin: null flag");
    hipLaunchKernelGGL(zero_i32_kernel, dim3(1), dim3(64), 0, ST(stream), flag, 1);
    GAOT_CHECK_LAUNCH("gaot_guard_begin");
    return GAOT_OK;
}

extern "C" int gaot_guard_compare(const void* current, const void* kept, int64_t nbytes, int32_t* flag, gaot_stream_t stream) {
    GAOT_REQUIRE(current && kept && flag && nbytes >= 0 && nbytes % 4 == 0, "guard_compare: bad arguments (nbytes %% 4 == 0)");
    if (nbytes == 0) return GAOT_OK;
    const long nw = nbytes / 4;
    hipLaunchKernelGGL(guard_compare_kernel, dim3(cap_blocks(nw, 256, 512)), dim3(256), 0, ST(stream), (const uint32_t*)current,
                       (const uint32_t*)kept, nw, flag);
    GAOT_CHECK_LAUNCH("gaot_guard_compare");
    return GAOT_OK;
}

extern "C" int gaot_guard_sync2(const void* cur0, void* kept0, int64_t nbytes0, const void* cur1, void* kept1, int64_t nbytes1,
                                int32_t* flag, gaot_stream_t stream) {
    GAOT_REQUIRE(cur0 && kept0 && cur1 && kept1 && flag && nbytes0 >= 0 && nbytes1 >= 0 && nbytes0 % 4 == 0 && nbytes1 % 4 == 0,
                 "guard_sync2: bad arguments (sizes %% 4 == 0)");
    const long nw = (nbytes0 + nbytes1) / 4;
    if (nw == 0) return GAOT_OK;
    hipLaunchKernelGGL(guard_sync2_kernel, dim3(cap_blocks(nw, 256, 512)), dim3(256), 0, ST(stream), (const uint32_t*)cur0, (uint32_t*)kept0,
                       (long)(nbytes0 / 4), (const uint32_t*)cur1, (uint32_t*)kept1, (long)(nbytes1 / 4), flag);
    GAOT_CHECK_LAUNCH("gaot_guard_sync2");
    return GAOT_OK;
}

extern "C" int gaot_guard_update(const void* current, void* kept, int64_t nbytes, const int32_t* flag, gaot_stream_t stream) {
    GAOT_REQUIRE(current && kept && flag && nbytes >= 0 && nbytes % 4 == 0, "guard_update: bad arguments (nbytes %% 4 == 0)");
    if (nbytes == 0) return GAOT_OK;
    const long nw = nbytes / 4;
    hipLaunchKernelGGL(guard_update_kernel, dim3(cap_blocks(nw, 256, 512)), dim3(256), 0, ST(stream), (const uint32_t*)current,
                       (uint32_t*)kept, nw, flag);
    GAOT_CHECK_LAUNCH("gaot_guard_update");
    return GAOT_OK;
}

extern "C" int gaot_gno_gather_reduce(const float* w, const float* src, int32_t B, int32_t n_src_rows, int32_t C,
                                      const int32_t* splits, const int32_t* cols, const int32_t* edge_map,
                                      int32_t n_out_rows, const float* escale, float* out, gaot_stream_t stream) {
    GAOT_REQUIRE(src && splits && out, "gno_gather_reduce: null pointer");
    GAOT_REQUIRE(B > 0 && C > 0 && n_out_rows >= 0 && n_src_rows >= 0, "gno_gather_reduce: bad sizes");
    if (n_out_rows == 0) return GAOT_OK;
    const int vw = pick_vw(C, w ? (const void*)w : (const void*)src, src, out);
    const int lpr = pow2_ceil(cdiv(C, vw));
    GAOT_REQUIRE(lpr <= 256, "gno_gather_reduce: channel count %d too large", C);
    const int rpb = 256 / lpr;
    dim3 grid((unsigned)(cdiv(n_out_rows, rpb) * (long)B)), block(256);
#define GR(VW, HW, HM) hipLaunchKernelGGL((gather_reduce_kernel<VW, HW, HM>), grid, block, 0, ST(stream), w, src, B, \
                                          n_src_rows, C, splits, cols, edge_map, n_out_rows, escale, out, lpr, rpb)
#define GR_V(VW) do { if (w && edge_map) GR(VW, true, true); else if (w) GR(VW, true, false); \
                      else if (edge_map) GR(VW, false, true); else GR(VW, false, false); } while (0)
    if (vw == 4) GR_V(4); else if (vw == 2) GR_V(2); else GR_V(1);
#undef GR_V
#undef GR
    GAOT_CHECK_LAUNCH("gaot_gno_gather_reduce");
    return GAOT_OK;
}

extern "C" int gaot_gno_edge_grad(const float* dout, const float* f, int32_t B, int32_t Q, int32_t n_src, int32_t C,
                                  const int32_t* index32, const int32_t* edge_query, int32_t E, const float* escale,
                                  float* dw, gaot_stream_t stream) {
    GAOT_REQUIRE(B > 0 && C > 0 && E >= 0, "gno_edge_grad: bad sizes");
    if (E == 0) return GAOT_OK;
    GAOT_REQUIRE(dout && f && index32 && edge_query && dw, "gno_edge_grad: null pointer");
    const int vw = pick_vw(C, dout, f, dw);
    const int lpr = pow2_ceil(cdiv(C, vw));
    GAOT_REQUIRE(lpr <= 256, "gno_edge_grad: channel count %d too large", C);
    const int rpb = 256 / lpr;
    dim3 grid(cdiv(E, rpb)), block(256);
#define EG(VW) hipLaunchKernelGGL((edge_grad_kernel<VW>), grid, block, 0, ST(stream), dout, f, B, Q, n_src, C, index32, \
                                  edge_query, E, escale, dw, lpr, rpb)
    if (vw == 4) EG(4); else if (vw == 2) EG(2); else EG(1);
#undef EG
    GAOT_CHECK_LAUNCH("gaot_gno_edge_grad");
    return GAOT_OK;
}

namespace gaot {
// ---------------------------------------------------------------- lifting fused into the encoder's integral transform
// The encoder's features are a point-wise LINEAR lifting of the raw node data (magno.py:334 -> ChannelMLP, one Conv1d(k=1)):
//     f[b,j,:] = Wl pn[b,j,:] + bl            (pn has CI <= 4 input channels)
// so   out[b,q,:] = sum_e a_e k_e (*) f[b,j(e),:]  =  sum_ci Wl[:,ci] (*) S_ci[b,q,:]  +  bl (*) S_0[q,:]
// with S_ci[b,q,:] = sum_e a_e pn[b,j(e),ci] k_e  and  S_0[q,:] = sum_e a_e k_e.
// The [B,n,C] lifted tensor (33.5 MB at 16 k nodes x 8) is never written or gathered: per edge and sample the kernel
// reads CI scalars instead of a 256-byte feature row.  Threads: (query row, 4-channel quad); grid.y = batch chunks of BCH.
template <int CI, int BCH>
__global__ __launch_bounds__(256) void lift_gather_reduce_kernel(const float* __restrict__ k, const float* __restrict__ pn,
                                                                 const float* __restrict__ wl, const float* __restrict__ bl,
                                                                 int B, int n_src, int C, const int* __restrict__ sp,
                                                                 const int* __restrict__ cols, int Q, const float* __restrict__ escale,
                                                                 float* __restrict__ out, int lanes_per_row, int rows_per_block,
                                                                 float* __restrict__ out_amax) {
    __shared__ float amred[4];
    const int r = blockIdx.x * rows_per_block + threadIdx.x / lanes_per_row;
    const int c = (threadIdx.x % lanes_per_row) * 4;
    const int b0 = blockIdx.y * BCH;
    const bool live = r < Q && c < C;            // (idle threads run an empty segment: the workgroup publishes the output's magnitude together)
    const int t0 = live ? sp[r] : 0, t1 = live ? sp[r + 1] : 0;
    f32x4 acc[BCH][CI], s0 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < BCH; ++b)
#pragma unroll
        for (int ci = 0; ci < CI; ++ci) acc[b][ci] = f32x4{0.f, 0.f, 0.f, 0.f};
    // four edges in flight: the index -> node-data loads are a dependent chain, the segment is short (3-30 edges)
    for (int t = t0; t < t1; t += 4) {
        int j[4]; f32x4 kq[4]; float pv[4][BCH][CI];
#pragma unroll
        for (int u = 0; u < 4; ++u) j[u] = cols[min(t + u, t1 - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int tt = min(t + u, t1 - 1);
            kq[u] = *reinterpret_cast<const f32x4*>(k + (long)tt * C + c);
            const float a = (t + u < t1) ? (escale ? escale[tt] : 1.0f) : 0.0f;       // padding edges weigh nothing
            kq[u] *= a;
#pragma unroll
            for (int b = 0; b < BCH; ++b) {
                const float* pr = pn + ((long)min(b0 + b, B - 1) * n_src + j[u]) * CI;
#pragma unroll
                for (int ci = 0; ci < CI; ++ci) pv[u][b][ci] = pr[ci];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            s0 += kq[u];
#pragma unroll
            for (int b = 0; b < BCH; ++b)
#pragma unroll
                for (int ci = 0; ci < CI; ++ci) acc[b][ci] += kq[u] * pv[u][b][ci];
        }
    }
    float am = 0.f;
    if (live) {
        f32x4 wq[CI];
#pragma unroll
        for (int ci = 0; ci < CI; ++ci) wq[ci] = f32x4{wl[(c + 0) * CI + ci], wl[(c + 1) * CI + ci], wl[(c + 2) * CI + ci], wl[(c + 3) * CI + ci]};
        const f32x4 bq = bl ? *reinterpret_cast<const f32x4*>(bl + c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < BCH; ++b) {
            if (b0 + b >= B) break;
            f32x4 o = bq * s0;
#pragma unroll
            for (int ci = 0; ci < CI; ++ci) o += wq[ci] * acc[b][ci];
            *reinterpret_cast<f32x4*>(out + ((long)(b0 + b) * Q + r) * C + c) = o;
            am = fmaxf(am, fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
        }
    }
    if (out_amax != nullptr) amax_publish_block<4>(out_amax, am, amred);      // the input word of the Linear that follows (its weight gradient)
}

// backward of the above for one edge row and channel quad: t_ci = sum_b dOut[b,q,:] pn[b,j,ci], u = sum_b dOut[b,q,:]
//   dk[e,:]   = a_e (sum_ci Wl[:,ci] t_ci + bl u)
//   dWl[:,ci] += a_e k_e t_ci ,  dbl += a_e k_e u         (per-workgroup partial rows [C*(CI+1)], summed by gaot_colsum)
template <int CI>
__global__ __launch_bounds__(256) void lift_edge_grad_kernel(const float* __restrict__ dout, const float* __restrict__ k,
                                                             const float* __restrict__ pn, const float* __restrict__ wl,
                                                             const float* __restrict__ bl, int B, int Q, int n_src, int C,
                                                             const int* __restrict__ idx, const int* __restrict__ eq, int E,
                                                             const float* __restrict__ escale, float* __restrict__ dk,
                                                             float* __restrict__ part, int lanes_per_row, int rows_per_block) {
    extern __shared__ __attribute__((aligned(16))) float red[];      // [rows_per_block][(CI + 1) * C]
    const int row = threadIdx.x / lanes_per_row;
    const int c = (threadIdx.x % lanes_per_row) * 4;
    const bool cok = c < C;
    f32x4 wq[CI], pw[CI], pb = {0.f, 0.f, 0.f, 0.f};
    f32x4 bq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ci = 0; ci < CI; ++ci) {
        pw[ci] = f32x4{0.f, 0.f, 0.f, 0.f};
        wq[ci] = cok ? f32x4{wl[(c + 0) * CI + ci], wl[(c + 1) * CI + ci], wl[(c + 2) * CI + ci], wl[(c + 3) * CI + ci]} : pb;
    }
    if (cok && bl) bq = *reinterpret_cast<const f32x4*>(bl + c);
    // a contiguous stretch of the edge list per workgroup, a contiguous range of stretches per XCD (workgroup b runs on XCD b % 8): the
    // edges are sorted by query row, so an XCD's L2 sees one eighth of the gradient rows dOut[b,q,:] (as proj_edge_grad_kernel)
    int lb;
    {
        const int nb = gridDim.x, q8 = nb >> 3, r8 = nb & 7, x = blockIdx.x & 7, slot = blockIdx.x >> 3;
        lb = (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + slot;
    }
    const int per = ((E + (int)gridDim.x * rows_per_block - 1) / ((int)gridDim.x * rows_per_block)) * rows_per_block;
    const int e_end = min(E, (lb + 1) * per);
    for (int e = lb * per + row; e < e_end && cok; e += rows_per_block) {
        const int q = eq[e], j = idx[e];
        f32x4 t[CI], u = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ci = 0; ci < CI; ++ci) t[ci] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int b = 0; b < B; ++b) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(dout + ((long)b * Q + q) * C + c);
            const float* pr = pn + ((long)b * n_src + j) * CI;
            u += g;
#pragma unroll
            for (int ci = 0; ci < CI; ++ci) t[ci] += g * pr[ci];
        }
        const float a = escale ? escale[e] : 1.0f;
        f32x4 d = bq * u;
#pragma unroll
        for (int ci = 0; ci < CI; ++ci) d += wq[ci] * t[ci];
        *reinterpret_cast<f32x4*>(dk + (long)e * C + c) = d * a;
        const f32x4 ka = *reinterpret_cast<const f32x4*>(k + (long)e * C + c) * a;
        pb += ka * u;
#pragma unroll
        for (int ci = 0; ci < CI; ++ci) pw[ci] += ka * t[ci];
    }
    // fixed-order sum over the workgroup's rows, then one partial row per workgroup: [ci][C] weights, then [C] bias
    const int W = (CI + 1) * C;
    if (cok) {
#pragma unroll
        for (int ci = 0; ci < CI; ++ci) *reinterpret_cast<f32x4*>(red + row * W + ci * C + c) = pw[ci];
        *reinterpret_cast<f32x4*>(red + row * W + CI * C + c) = pb;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < W; i += 256) {
        float sacc = 0.f;
        for (int rr = 0; rr < rows_per_block; ++rr) sacc += red[rr * W + i];
        part[(long)blockIdx.x * W + i] = sacc;
    }
}

// ---------------------------------------------------------------- output projection fused into the decoder's transform
// The decoder's integral transform is followed by point-wise LINEAR maps only (recovery Conv1d, projection Conv1d:
// magno.py:640-668), folded on the host into one [OC, C] matrix weff and a per-query row bias.  With OC <= 4 output channels
//     y[b,q,o] = sum_ch weff[o,ch] * (sum_e a_e k[e,ch] f[b,j(e),ch]) + rowb[q,o] + bias[o]
// is produced directly: the [B,Nq,C] transform output (33.5 MB at 16 k query nodes x 8) is never written, and in the backward
// its gradient G[b,q,ch] = sum_o dY[b,q,o] weff[o,ch] is formed on the fly from the OC scalars of a row.
template <int OC>
__global__ __launch_bounds__(256) void proj_gather_reduce_kernel(const float* __restrict__ k, const float* __restrict__ f,
                                                                 const float* __restrict__ weff, const float* __restrict__ rowb,
                                                                 const float* __restrict__ bias, int B, int n_src, int C,
                                                                 const int* __restrict__ sp, const int* __restrict__ cols, int Q,
                                                                 const float* __restrict__ escale, float* __restrict__ y,
                                                                 int lanes_per_row, int rows_per_block) {
    const int b = blockIdx.x % B;
    const int r = (blockIdx.x / B) * rows_per_block + threadIdx.x / lanes_per_row;
    const int lr = threadIdx.x % lanes_per_row;
    const int c = lr * 4;
    const bool ok = r < Q && c < C;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (ok) {
        const int t0 = sp[r], t1 = sp[r + 1];
        const float* fb = f + (long)b * n_src * C + c;
        for (int t = t0; t < t1; t += 4) {
            int j[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) j[u] = cols[min(t + u, t1 - 1)];
            f32x4 fv[4], kv[4]; float a[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int tt = min(t + u, t1 - 1);
                fv[u] = *reinterpret_cast<const f32x4*>(fb + (long)j[u] * C);
                kv[u] = *reinterpret_cast<const f32x4*>(k + (long)tt * C + c);
                a[u] = (t + u < t1) ? (escale ? escale[tt] : 1.0f) : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += kv[u] * (fv[u] * a[u]);
        }
    }
#pragma unroll
    for (int o = 0; o < OC; ++o) {
        float d = 0.f;
        if (ok) { const f32x4 w = *reinterpret_cast<const f32x4*>(weff + (long)o * C + c); d = (acc[0] * w[0] + acc[1] * w[1]) + (acc[2] * w[2] + acc[3] * w[3]); }
        for (int off = lanes_per_row >> 1; off > 0; off >>= 1) d += __shfl_xor(d, off, 64);
        if (lr == 0 && r < Q) y[((long)b * Q + r) * OC + o] = d + (rowb ? rowb[(long)r * OC + o] : 0.f) + (bias ? bias[o] : 0.f);
    }
}

// per edge row: T_o[e,:] = a_e sum_b dY[b,q,o] f[b,j,:];  dk[e,:] = sum_o weff[o,:] T_o;  dweff[o,:] += k[e,:] T_o (partials)
template <int OC>
__global__ __launch_bounds__(256) void proj_edge_grad_kernel(const float* __restrict__ dy, const float* __restrict__ k,
                                                             const float* __restrict__ f, const float* __restrict__ weff, int B,
                                                             int Q, int n_src, int C, const int* __restrict__ idx,
                                                             const int* __restrict__ eq, int E, const float* __restrict__ escale,
                                                             float* __restrict__ dk, float* __restrict__ part, int lanes_per_row,
                                                             int rows_per_block, const int* __restrict__ order) {
    extern __shared__ __attribute__((aligned(16))) float red[];      // [rows_per_block][OC * C]
    const int row = threadIdx.x / lanes_per_row;
    const int c = (threadIdx.x % lanes_per_row) * 4;
    const bool cok = c < C;
    f32x4 wq[OC], pw[OC];
#pragma unroll
    for (int o = 0; o < OC; ++o) {
        pw[o] = f32x4{0.f, 0.f, 0.f, 0.f};
        wq[o] = cok ? *reinterpret_cast<const f32x4*>(weff + (long)o * C + c) : pw[o];
    }
    // The edges are walked in the order of the TRANSPOSED CSR (`order` = its edge list: sorted by source row j), a contiguous stretch per
    // workgroup and a contiguous range of stretches per XCD (workgroup b runs on XCD b % 8): the B gathered feature rows f[b,j,:] of
    // consecutive edges are the same rows, and an XCD's L2 sees one eighth of f.  In CSR order (rows = mesh points in the caller's order,
    // i.e. spatially random) every edge pulled its B rows through the fabric again: 99 MB per launch for 8 MB of features (PMC).
    int lb;
    {
        const int nb = gridDim.x, q8 = nb >> 3, r8 = nb & 7, x = blockIdx.x & 7, slot = blockIdx.x >> 3;
        lb = (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + slot;
    }
    const int per = ((E + (int)gridDim.x * rows_per_block - 1) / ((int)gridDim.x * rows_per_block)) * rows_per_block;     // positions per workgroup
    const int i_end = min(E, (lb + 1) * per);
    for (int i = lb * per + row; i < i_end && cok; i += rows_per_block) {
        const int e = order ? order[i] : i;
        const int q = eq[e], j = idx[e];
        f32x4 t[OC];
#pragma unroll
        for (int o = 0; o < OC; ++o) t[o] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int b = 0; b < B; ++b) {
            const f32x4 fv = *reinterpret_cast<const f32x4*>(f + ((long)b * n_src + j) * C + c);
            const float* gy = dy + ((long)b * Q + q) * OC;
#pragma unroll
            for (int o = 0; o < OC; ++o) t[o] += fv * gy[o];
        }
        const float a = escale ? escale[e] : 1.0f;
        const f32x4 kv = *reinterpret_cast<const f32x4*>(k + (long)e * C + c);
        f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int o = 0; o < OC; ++o) { t[o] *= a; d += wq[o] * t[o]; pw[o] += kv * t[o]; }
        *reinterpret_cast<f32x4*>(dk + (long)e * C + c) = d;
    }
    const int W = OC * C;
    if (cok) {
#pragma unroll
        for (int o = 0; o < OC; ++o) *reinterpret_cast<f32x4*>(red + row * W + o * C + c) = pw[o];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < W; i += 256) {
        float sacc = 0.f;
        for (int rr = 0; rr < rows_per_block; ++rr) sacc += red[rr * W + i];
        part[(long)blockIdx.x * W + i] = sacc;
    }
}

// dF[b,j,:] = sum_{t in tseg(j)} a_e k[e,:] (*) (sum_o dY[b,q(e),o] weff[o,:]),  e = t_edge[t]
template <int OC>
__global__ __launch_bounds__(256) void proj_gather_t_kernel(const float* __restrict__ k, const float* __restrict__ dy,
                                                            const float* __restrict__ weff, int B, int Q, int C,
                                                            const int* __restrict__ tsp, const int* __restrict__ tedge,
                                                            const int* __restrict__ eq, int n_src, const float* __restrict__ escale,
                                                            float* __restrict__ df, int lanes_per_row, int rows_per_block) {
    const int b = blockIdx.x % B;
    const int r = (blockIdx.x / B) * rows_per_block + threadIdx.x / lanes_per_row;
    const int c = (threadIdx.x % lanes_per_row) * 4;
    if (r >= n_src || c >= C) return;
    f32x4 wq[OC];
#pragma unroll
    for (int o = 0; o < OC; ++o) wq[o] = *reinterpret_cast<const f32x4*>(weff + (long)o * C + c);
    const float* dyb = dy + (long)b * Q * OC;
    const int t0 = tsp[r], t1 = tsp[r + 1];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int t = t0; t < t1; t += 4) {
        int e[4], q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) e[u] = tedge[min(t + u, t1 - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u) q[u] = eq[e[u]];
        f32x4 kv[4], g[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            kv[u] = *reinterpret_cast<const f32x4*>(k + (long)e[u] * C + c);
            const float a = (t + u < t1) ? (escale ? escale[e[u]] : 1.0f) : 0.0f;
            kv[u] *= a;
            g[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int o = 0; o < OC; ++o) g[u] += wq[o] * dyb[(long)q[u] * OC + o];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += kv[u] * g[u];
    }
    *reinterpret_cast<f32x4*>(df + ((long)b * n_src + r) * C + c) = acc;
}

}  // namespace gaot

extern "C" int gaot_gno_lift_gather_reduce(const float* k, const float* pn, const float* wl, const float* bl, int32_t B,
                                           int32_t n_src, int32_t c_in, int32_t C, const int32_t* splits, const int32_t* cols,
                                           int32_t Q, const float* escale, float* out, float* out_absmax, gaot_stream_t stream) {
    GAOT_REQUIRE(B > 0 && n_src > 0 && Q >= 0 && c_in >= 1 && c_in <= 4 && C > 0 && C % 4 == 0 && C <= 1024,
                 "gno_lift_gather_reduce: need 1 <= c_in <= 4 and C %% 4 == 0 (got c_in %d, C %d)", c_in, C);
    if (Q == 0) return GAOT_OK;
    GAOT_REQUIRE(k && pn && wl && splits && cols && out && aligned16(k) && aligned16(out) && (!bl || aligned16(bl)),
                 "gno_lift_gather_reduce: null or misaligned pointer");
    const int lpr = pow2_ceil(C / 4), rpb = 256 / lpr;
    constexpr int BCH = 2;
    dim3 grid(cdiv(Q, rpb), cdiv(B, BCH)), block(256);
#define LG(CI) hipLaunchKernelGGL((lift_gather_reduce_kernel<CI, BCH>), grid, block, 0, ST(stream), k, pn, wl, bl, B, n_src, C, splits, \
                                  cols, Q, escale, out, lpr, rpb, out_absmax)
    if (c_in == 1) LG(1); else if (c_in == 2) LG(2); else if (c_in == 3) LG(3); else LG(4);
#undef LG
    GAOT_CHECK_LAUNCH("gaot_gno_lift_gather_reduce");
    return GAOT_OK;
}

extern "C" int32_t gaot_gno_lift_edge_grad_parts(int32_t E, int32_t C) {
    const int lpr = pow2_ceil(C / 4), rpb = 256 / lpr;
    int nb = cdiv(E > 0 ? E : 1, rpb);
    return nb > 1024 ? 1024 : nb;
}

extern "C" int gaot_gno_lift_edge_grad(const float* dout, const float* k, const float* pn, const float* wl, const float* bl,
                                       int32_t B, int32_t Q, int32_t n_src, int32_t c_in, int32_t C, const int32_t* index32,
                                       const int32_t* edge_query, int32_t E, const float* escale, float* dk, float* partial,
                                       gaot_stream_t stream) {
    GAOT_REQUIRE(B > 0 && E > 0 && c_in >= 1 && c_in <= 4 && C > 0 && C % 4 == 0 && C <= 1024,
                 "gno_lift_edge_grad: need E > 0, 1 <= c_in <= 4 and C %% 4 == 0 (got c_in %d, C %d)", c_in, C);
    GAOT_REQUIRE(dout && k && pn && wl && index32 && edge_query && dk && partial && aligned16(dout) && aligned16(k) && aligned16(dk) &&
                 (!bl || aligned16(bl)), "gno_lift_edge_grad: null or misaligned pointer");
    const int lpr = pow2_ceil(C / 4), rpb = 256 / lpr;
    const int nb = gaot_gno_lift_edge_grad_parts(E, C);
    const size_t lds = sizeof(float) * (size_t)rpb * (c_in + 1) * C;
    GAOT_REQUIRE(lds <= 64 * 1024, "gno_lift_edge_grad: C = %d too wide for the workgroup reduction", C);
#define LE(CI) hipLaunchKernelGGL((lift_edge_grad_kernel<CI>), dim3(nb), dim3(256), lds, ST(stream), dout, k, pn, wl, bl, B, Q, n_src, C, \
                                  index32, edge_query, E, escale, dk, partial, lpr, rpb)
    if (c_in == 1) LE(1); else if (c_in == 2) LE(2); else if (c_in == 3) LE(3); else LE(4);
#undef LE
    GAOT_CHECK_LAUNCH("gaot_gno_lift_edge_grad");
    return GAOT_OK;
}

static int proj_check(int B, int C, int OC) {
    GAOT_REQUIRE(B > 0 && OC >= 1 && OC <= 4 && C > 0 && C % 4 == 0 && C <= 1024,
                 "gno_proj_*: need 1 <= out_channels <= 4 and C %% 4 == 0 (got %d, C %d)", OC, C);
    return GAOT_OK;
}

extern "C" int gaot_gno_proj_gather_reduce(const float* k, const float* f, const float* weff, const float* rowbias, const float* bias,
                                           int32_t B, int32_t n_src, int32_t C, int32_t out_channels, const int32_t* splits,
                                           const int32_t* cols, int32_t Q, const float* escale, float* y, gaot_stream_t stream) {
    if (int rc = proj_check(B, C, out_channels)) return rc;
    if (Q == 0) return GAOT_OK;
    GAOT_REQUIRE(k && f && weff && splits && cols && y && aligned16(k) && aligned16(f) && aligned16(weff), "gno_proj_gather_reduce: null or misaligned pointer");
    const int lpr = pow2_ceil(C / 4), rpb = 256 / lpr;
    GAOT_REQUIRE(lpr <= 64, "gno_proj_gather_reduce: C = %d too wide (the row reduction is one wave)", C);
    dim3 grid((unsigned)(cdiv(Q, rpb) * (long)B)), block(256);
#define PG(OC) hipLaunchKernelGGL((proj_gather_reduce_kernel<OC>), grid, block, 0, ST(stream), k, f, weff, rowbias, bias, B, n_src, C, splits, \
                                  cols, Q, escale, y, lpr, rpb)
    if (out_channels == 1) PG(1); else if (out_channels == 2) PG(2); else if (out_channels == 3) PG(3); else PG(4);
#undef PG
    GAOT_CHECK_LAUNCH("gaot_gno_proj_gather_reduce");
    return GAOT_OK;
}

extern "C" int gaot_gno_proj_backward(const float* dy, const float* k, const float* f, const float* weff, int32_t B, int32_t Q,
                                      int32_t n_src, int32_t C, int32_t out_channels, const int32_t* index32,
                                      const int32_t* edge_query, int32_t E, const int32_t* t_splits, const int32_t* t_edge,
                                      const float* escale, float* dk, float* dweff_partial, float* df, gaot_stream_t stream) {
    if (int rc = proj_check(B, C, out_channels)) return rc;
    GAOT_REQUIRE(E > 0 && dy && k && f && weff && index32 && edge_query && dk && dweff_partial && aligned16(k) && aligned16(f) && aligned16(weff) &&
                 aligned16(dk), "gno_proj_backward: E > 0 and non-null 16-byte aligned operands required");
    const int lpr = pow2_ceil(C / 4), rpb = 256 / lpr;
    const int nb = gaot_gno_lift_edge_grad_parts(E, C);
    const size_t lds = sizeof(float) * (size_t)rpb * out_channels * C;
    GAOT_REQUIRE(lds <= 64 * 1024, "gno_proj_backward: C = %d too wide for the workgroup reduction", C);
#define PE(OC) hipLaunchKernelGGL((proj_edge_grad_kernel<OC>), dim3(nb), dim3(256), lds, ST(stream), dy, k, f, weff, B, Q, n_src, C, index32, \
                                  edge_query, E, escale, dk, dweff_partial, lpr, rpb, t_edge)
    if (out_channels == 1) PE(1); else if (out_channels == 2) PE(2); else if (out_channels == 3) PE(3); else PE(4);
#undef PE
    if (df) {
        GAOT_REQUIRE(t_splits && t_edge && aligned16(df), "gno_proj_backward: dF needs the transposed CSR");
        dim3 grid((unsigned)(cdiv(n_src, rpb) * (long)B)), block(256);
#define PT(OC) hipLaunchKernelGGL((proj_gather_t_kernel<OC>), grid, block, 0, ST(stream), k, dy, weff, B, Q, C, t_splits, t_edge, edge_query, \
                                  n_src, escale, df, lpr, rpb)
        if (out_channels == 1) PT(1); else if (out_channels == 2) PT(2); else if (out_channels == 3) PT(3); else PT(4);
#undef PT
    }
    GAOT_CHECK_LAUNCH("gaot_gno_proj_backward");
    return GAOT_OK;
}

extern "C" int gaot_gno_segment_sum(const float* x, int32_t B, int32_t E, int32_t C, const int32_t* splits, int32_t Q,
                                    const float* rowscale, float* out, gaot_stream_t stream) {
    GAOT_REQUIRE(B > 0 && C > 0 && Q >= 0 && E >= 0, "gno_segment_sum: bad sizes");
    if (Q == 0) return GAOT_OK;
    GAOT_REQUIRE(splits && out, "gno_segment_sum: null pointer");
    const int vw = pick_vw(C, x ? (const void*)x : (const void*)out, out, out);
    const int lpr = pow2_ceil(cdiv(C, vw));
    GAOT_REQUIRE(lpr <= 256, "gno_segment_sum: channel count %d too large", C);
    const int rpb = 256 / lpr;
    dim3 grid((unsigned)(cdiv(Q, rpb) * (long)B)), block(256);
#define SS(VW) hipLaunchKernelGGL((segment_sum_kernel<VW>), grid, block, 0, ST(stream), x, B, E, C, splits, Q, rowscale, out, lpr, rpb)
    if (vw == 4) SS(4); else if (vw == 2) SS(2); else SS(1);
#undef SS
    GAOT_CHECK_LAUNCH("gaot_gno_segment_sum");
    return GAOT_OK;
}
