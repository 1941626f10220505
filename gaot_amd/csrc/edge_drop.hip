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
