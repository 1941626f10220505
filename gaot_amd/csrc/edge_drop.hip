// Training-time neighbour sub-sampling ON THE DEVICE (reference edge_drop.py:8-106: 'ratio' keeps every edge with probability sample_ratio,
// 'max_neighbors' keeps a uniformly random subset of max_neighbors edges of every row that has more), as a compaction of a plan's arrays
// into STATIC buffers of the plan's own capacity: nothing about the draw is a launch argument or a host value -- the seed is a device word
// advanced per draw (gaot_attention_seed_next), the kept edge count stays on the device (the padded-plan contract of gaot_union_compose:
// edges past *e_real_out belong to no row, carry source 0 / query 0 / t_edge = own id, and need an edge scale of 0) -- so a captured
// (hipGraph) training step draws a fresh subset on every replay.
//
//   mark      keep[e]            per edge ('ratio': hash(seed, e) < ratio; 'max_neighbors': rank of hash(seed, e) inside its row < max)
//   flags     fa[e] = keep[e] for e < *e_real else 0;   ft[p] = fa[t_edge[p]]          (the transposed list, in ITS order)
//   scan      pa = exclusive scan(fa), pt = exclusive scan(ft)                           (three launches, both arrays per launch)
//   compact   index', edge_query' at pa[e];  t_edge'[pt[p]] = pa[t_edge[p]];  splits'[r] = pa[splits[r]];  t_splits'[j] = pt[t_splits[j]]
// Both CSRs stay sorted (a compaction preserves order), so no sort is needed and the backward stays deterministic.
#include "common.h"
#include "segsort.h"

namespace gaot {

__device__ __forceinline__ unsigned drop_key(unsigned long long seed, int e) {
    unsigned long long x = seed + 0x9e3779b97f4a7c15ull * (unsigned long long)(e + 1);
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27; x *= 0x94d049bb133111ebull;
    x ^= x >> 31;
    return (unsigned)(x >> 32);
}

__global__ void drop_mark_ratio_kernel(const unsigned long long* __restrict__ seed, int E, unsigned thresh, int* __restrict__ keep) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < E) keep[e] = drop_key(*seed, e) < thresh ? 1 : 0;
}

// 8 lanes per row; rows within the limit keep everything (the reference returns such rows untouched), longer rows keep the max_n edges
// with the smallest (key, position): a uniformly random subset of that size
#define DROP_LANES 8
__global__ __launch_bounds__(256) void drop_mark_maxn_kernel(const unsigned long long* __restrict__ seed, const int* __restrict__ sp, int Q,
                                                             int max_n, int* __restrict__ keep) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int q = gid / DROP_LANES, l = gid % DROP_LANES;
    if (q >= Q) return;
    const int b = sp[q], e = sp[q + 1];
    if (e - b <= max_n) { for (int t = b + l; t < e; t += DROP_LANES) keep[t] = 1; return; }
    const unsigned long long s = *seed;
    for (int t = b + l; t < e; t += DROP_LANES) {
        const unsigned kt = drop_key(s, t);
        int rank = 0;
        for (int u = b; u < e; ++u) { const unsigned ku = drop_key(s, u); rank += (ku < kt || (ku == kt && u < t)) ? 1 : 0; }
        keep[t] = rank < max_n ? 1 : 0;
    }
}

__global__ void drop_flags_kernel(const int* __restrict__ keep, const int* __restrict__ tedge, int E, const int* __restrict__ e_real_in,
                                  int* __restrict__ fa, int* __restrict__ ft) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= E) return;
    const int er = e_real_in ? min(*e_real_in, E) : E;
    fa[i] = i < er ? keep[i] : 0;
    ft[i] = i < er ? keep[tedge[i]] : 0;
}

// exclusive scan of two int arrays of n entries each (blockIdx.y picks the array), SCAN_PER elements per workgroup
#define SCAN_PER 2048
__global__ __launch_bounds__(256) void scan_sums_kernel(const int* __restrict__ a0, const int* __restrict__ a1, int n, int* __restrict__ bsum, int nb) {
    __shared__ int red[4];
    const int* a = blockIdx.y ? a1 : a0;
    const int base = blockIdx.x * SCAN_PER;
    int s = 0;
    for (int i = threadIdx.x; i < SCAN_PER; i += 256) s += (base + i < n) ? a[base + i] : 0;
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) bsum[blockIdx.y * (nb + 1) + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(1024) void scan_mid_kernel(int* __restrict__ bsum, int nb) {      // exclusive, in place; one workgroup per array
    __shared__ int part[1024];
    int* v = bsum + blockIdx.x * (nb + 1);
    const int t = threadIdx.x;
    const int chunk = (nb + 1023) / 1024;
    const int b = t * chunk, e = min(nb, b + chunk);
    int s = 0;
    for (int i = b; i < e; ++i) s += v[i];
    part[t] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int w = (t >= off) ? part[t - off] : 0;
        __syncthreads();
        part[t] += w;
        __syncthreads();
    }
    int run = (t == 0) ? 0 : part[t - 1];
    for (int i = b; i < e; ++i) { const int c = v[i]; v[i] = run; run += c; }
    if (t == 1023) v[nb] = part[1023];          // entry nb = the total (read by the workgroup that writes the scan's closing entry)
}
__global__ __launch_bounds__(256) void scan_final_kernel(const int* __restrict__ a0, const int* __restrict__ a1, int n, const int* __restrict__ bsum,
                                                         int nb, int* __restrict__ o0, int* __restrict__ o1) {
    __shared__ int part[256];
    const int* a = blockIdx.y ? a1 : a0;
    int* o = blockIdx.y ? o1 : o0;
    const int base = blockIdx.x * SCAN_PER + threadIdx.x * (SCAN_PER / 256);
    int v[SCAN_PER / 256], s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_PER / 256; ++j) { v[j] = (base + j < n) ? a[base + j] : 0; s += v[j]; }
    part[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const int w = (threadIdx.x >= off) ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += w;
        __syncthreads();
    }
    int run = bsum[blockIdx.y * (nb + 1) + blockIdx.x] + (threadIdx.x ? part[threadIdx.x - 1] : 0);
#pragma unroll
    for (int j = 0; j < SCAN_PER / 256; ++j) { if (base + j <= n) o[base + j] = run; run += v[j]; }      // (entry n = the total)
}

__global__ void drop_compact_kernel(const int* __restrict__ index, const int* __restrict__ eq, const int* __restrict__ tedge,
                                    const int* __restrict__ sp, const int* __restrict__ tsp, int Q, int n_src, int E,
                                    const int* __restrict__ fa, const int* __restrict__ ft, const int* __restrict__ pa, const int* __restrict__ pt,
                                    int* __restrict__ o_index, int* __restrict__ o_eq, int* __restrict__ o_tedge, int* __restrict__ o_sp,
                                    int* __restrict__ o_tsp, int* __restrict__ e_real_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = pa[E];
    if (i == 0) *e_real_out = total;
    if (i < E) {
        if (fa[i]) { const int n = pa[i]; o_index[n] = index[i]; o_eq[n] = eq[i]; }
        if (ft[i]) o_tedge[pt[i]] = pa[tedge[i]];
        if (i >= total) { o_index[i] = 0; o_eq[i] = 0; o_tedge[i] = i; }
    }
    if (i <= Q) o_sp[i] = pa[min(sp[i], E)];
    if (i <= n_src) o_tsp[i] = pt[min(tsp[i], E)];
}

// ---- block-diagonal union straight from the callers' int64 CSR lists (gaot_union_compose_raw) + its transposed CSR with a device-side edge count
__global__ __launch_bounds__(256) void union_compose_raw_kernel(const gaot_union_part_raw* __restrict__ parts, int B, int Qe, int Se, int dsrc, int ddst,
                                                                int E_cap, int* __restrict__ index, int* __restrict__ eq, int* __restrict__ splits,
                                                                float* __restrict__ src, float* __restrict__ dst, int* __restrict__ e_real,
                                                                int* __restrict__ flag) {
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
            const gaot_union_part_raw& p = parts[lo];
            const long l = i - begin[lo];
            const int64_t j = p.index[l];
            if (j < 0 || j >= Se) atomicOr(flag, 2);          // (clamped: the kernels that run before the host reads the flag stay in bounds)
            index[i] = (int)(j < 0 ? 0 : (j >= Se ? Se - 1 : j)) + lo * Se;
            int a = 0, c = Qe;                        // upper_bound(splits, l) - 1: sp[a] <= l < sp[c] on a valid list
            while (c - a > 1) { const int mid = (a + c) >> 1; if (p.splits[mid] <= l) a = mid; else c = mid; }
            eq[i] = a + lo * Qe;
        } else {
            index[i] = 0; eq[i] = 0;
        }
    }
    const long Qt = (long)B * Qe, St = (long)B * Se;
    if (i <= Qt) {
        if (i == Qt) splits[i] = E;
        else {
            const gaot_union_part_raw& p = parts[i / Qe];
            const int r = (int)(i % Qe);
            const int64_t v = p.splits[r], nx = p.splits[r + 1];
            if (v < 0 || v > p.e_count || nx < v || (r == 0 && v != 0) || (r == Qe - 1 && nx != p.e_count)) atomicOr(flag, 1);
            splits[i] = min((int)(v < 0 ? 0 : (v > p.e_count ? p.e_count : v)) + begin[i / Qe], E);
        }
    }
    if (src && i < St * dsrc) { const long r = i / dsrc; src[i] = parts[r / Se].src[(r % Se) * dsrc + i % dsrc]; }
    if (dst && i < Qt * ddst) { const long r = i / ddst; dst[i] = parts[r / Qe].dst[(r % Qe) * ddst + i % ddst]; }
}
__global__ void tdev_zero_kernel(int* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0;
}
__global__ void tdev_count_kernel(const int* __restrict__ idx, int E, const int* __restrict__ e_real, int* __restrict__ cnt) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < min(E, *e_real)) atomicAdd(&cnt[idx[e]], 1);
}
__global__ void tdev_fill_kernel(const int* __restrict__ idx, int E, const int* __restrict__ e_real, const int* __restrict__ tsp, int* __restrict__ cnt,
                                 int* __restrict__ tedge) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    if (e < *e_real) {
        const int j = idx[e];
        const int slot = atomicSub(&cnt[j], 1) - 1;
        tedge[tsp[j] + slot] = e;
    } else {
        tedge[e] = e;
    }
}

}  // namespace gaot

using namespace gaot;
#define ST(s) reinterpret_cast<hipStream_t>(s)

extern "C" int64_t gaot_edge_drop_scratch(int32_t E) {
    const int nb = cdiv(E > 0 ? E : 1, SCAN_PER);
    return 5L * (E + 1) + 2L * (nb + 1);
}

extern "C" int gaot_edge_drop(const int32_t* index32, const int32_t* edge_query, const int32_t* t_edge, const int32_t* splits32,
                              const int32_t* t_splits, int32_t Q, int32_t n_src, int32_t E, const int32_t* e_real_in, int32_t mode,
                              float sample_ratio, int32_t max_neighbors, const uint64_t* seed, int32_t* out_index, int32_t* out_edge_query,
                              int32_t* out_t_edge, int32_t* out_splits, int32_t* out_t_splits, int32_t* e_real_out, int32_t* scratch,
                              gaot_stream_t stream) {
    GAOT_REQUIRE(Q >= 0 && n_src >= 0 && E >= 1 && (mode == 1 || mode == 2), "edge_drop: need E >= 1 and mode 1 ('ratio') or 2 ('max_neighbors')");
    GAOT_REQUIRE(index32 && edge_query && t_edge && splits32 && t_splits && seed && out_index && out_edge_query && out_t_edge && out_splits &&
                 out_t_splits && e_real_out && scratch, "edge_drop: null pointer");
    GAOT_REQUIRE(mode != 1 || (sample_ratio > 0.f && sample_ratio <= 1.f), "edge_drop: sample_ratio must be in (0, 1]");
    GAOT_REQUIRE(mode != 2 || max_neighbors > 0, "edge_drop: max_neighbors must be > 0");
    const int nb = cdiv(E, SCAN_PER);
    int* keep = scratch;
    int* fa = keep + (E + 1);
    int* ft = fa + (E + 1);
    int* pa = ft + (E + 1);
    int* pt = pa + (E + 1);
    int* bsum = pt + (E + 1);
    const unsigned long long* sd = reinterpret_cast<const unsigned long long*>(seed);
    if (mode == 1) {
        const double t = (double)sample_ratio * 4294967296.0;
        const unsigned thresh = t >= 4294967295.0 ? 0xffffffffu : (unsigned)t;
        hipLaunchKernelGGL(drop_mark_ratio_kernel, dim3(cdiv(E, 256)), dim3(256), 0, ST(stream), sd, E, thresh, keep);
    } else if (Q > 0) {
        hipLaunchKernelGGL(drop_mark_maxn_kernel, dim3(cdiv((long)Q * DROP_LANES, 256)), dim3(256), 0, ST(stream), sd, splits32, Q, max_neighbors, keep);
    }
    hipLaunchKernelGGL(drop_flags_kernel, dim3(cdiv(E, 256)), dim3(256), 0, ST(stream), keep, t_edge, E, e_real_in, fa, ft);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(nb, 2), dim3(256), 0, ST(stream), fa, ft, E, bsum, nb);
    hipLaunchKernelGGL(scan_mid_kernel, dim3(2), dim3(1024), 0, ST(stream), bsum, nb);
    hipLaunchKernelGGL(scan_final_kernel, dim3(cdiv(E + 1, SCAN_PER), 2), dim3(256), 0, ST(stream), fa, ft, E, bsum, nb, pa, pt);
    int n = E;
    if (Q + 1 > n) n = Q + 1;
    if (n_src + 1 > n) n = n_src + 1;
    hipLaunchKernelGGL(drop_compact_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ST(stream), index32, edge_query, t_edge, splits32, t_splits, Q, n_src,
                       E, fa, ft, pa, pt, out_index, out_edge_query, out_t_edge, out_splits, out_t_splits, e_real_out);
    GAOT_CHECK_LAUNCH("gaot_edge_drop");
    return GAOT_OK;
}

extern "C" int gaot_union_compose_raw(const gaot_union_part_raw* parts_dev, int32_t n_parts, int32_t q_each, int32_t n_src_each, int32_t dim_src,
                                      int32_t dim_dst, int32_t e_cap, int32_t* index32, int32_t* edge_query, int32_t* splits32, float* src,
                                      float* dst, int32_t* e_real, int32_t* status_flag, gaot_stream_t stream) {
    GAOT_REQUIRE(parts_dev && n_parts >= 1 && n_parts <= 1024 && q_each > 0 && n_src_each > 0 && e_cap >= 1 && dim_src >= 0 && dim_dst >= 0,
                 "union_compose_raw: bad sizes (1 <= n_parts <= 1024, q_each, n_src_each, e_cap > 0)");
    GAOT_REQUIRE(index32 && edge_query && splits32 && e_real && status_flag, "union_compose_raw: null output pointer");
    long n = e_cap;
    const long Qt = (long)n_parts * q_each, St = (long)n_parts * n_src_each;
    if (Qt + 1 > n) n = Qt + 1;
    if (src && St * dim_src > n) n = St * dim_src;
    if (dst && Qt * dim_dst > n) n = Qt * dim_dst;
    hipLaunchKernelGGL(union_compose_raw_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ST(stream), parts_dev, n_parts, q_each, n_src_each, dim_src,
                       dim_dst, e_cap, index32, edge_query, splits32, src, dst, e_real, status_flag);
    GAOT_CHECK_LAUNCH("gaot_union_compose_raw");
    return GAOT_OK;
}

extern "C" int64_t gaot_csr_transpose_dev_scratch(int32_t E, int32_t n_src) {
    return (int64_t)(n_src + 1) + 2L * (cdiv(n_src > 0 ? n_src : 1, SCAN_PER) + 1) + (E > 0 ? E : 1);
}

extern "C" int gaot_csr_transpose_dev(const int32_t* index32, int32_t E, const int32_t* e_real, int32_t n_src, int32_t* t_splits, int32_t* t_edge,
                                      int32_t* scratch, gaot_stream_t stream) {
    GAOT_REQUIRE(index32 && e_real && t_splits && t_edge && scratch && n_src > 0 && E >= 1, "csr_transpose_dev: bad arguments");
    const int nb = cdiv(n_src, SCAN_PER);
    int* cnt = scratch;
    int* bsum = cnt + (n_src + 1);
    int* sortbuf = bsum + 2 * (nb + 1);
    hipLaunchKernelGGL(tdev_zero_kernel, dim3(cdiv(n_src + 1, 256)), dim3(256), 0, ST(stream), cnt, n_src + 1);
    hipLaunchKernelGGL(tdev_count_kernel, dim3(cdiv(E, 256)), dim3(256), 0, ST(stream), index32, E, e_real, cnt);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(nb, 1), dim3(256), 0, ST(stream), cnt, cnt, n_src, bsum, nb);
    hipLaunchKernelGGL(scan_mid_kernel, dim3(1), dim3(1024), 0, ST(stream), bsum, nb);
    hipLaunchKernelGGL(scan_final_kernel, dim3(cdiv(n_src + 1, SCAN_PER), 1), dim3(256), 0, ST(stream), cnt, cnt, n_src, bsum, nb, t_splits, t_splits);
    hipLaunchKernelGGL(tdev_fill_kernel, dim3(cdiv(E, 256)), dim3(256), 0, ST(stream), index32, E, e_real, t_splits, cnt, t_edge);
    sort_segments<int, int>(t_splits, n_src, t_edge, sortbuf, ST(stream));      // ascending edge ids per source: deterministic backward
    GAOT_CHECK_LAUNCH("gaot_csr_transpose_dev");
    return GAOT_OK;
}
