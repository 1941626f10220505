// Radius-graph builder (cell list) -> the reference's CSR neighbour dict (neighbor_search.py:65-146 semantics:
// inclusive `dist <= r`, unbounded degree, neighbours of a query in ascending data index like the `native` backend).
// Replaces the O(Q*N) distance matrix of torch.cdist with O(Q * points-in-27-cells).  Once per geometry.
#include "common.h"
#include "segsort.h"

namespace gaot {

struct CellGrid {
    float ox, oy, oz;      // origin
    float inv_cell;        // 1 / cell size
    int nx, ny, nz;        // cells per axis (nz = 1 in 2-D)
    int dim;
};

__device__ __forceinline__ int cell_coord(float v, float o, float inv, int n) {
    int c = (int)floorf((v - o) * inv);
    return c < 0 ? 0 : (c >= n ? n - 1 : c);
}
__device__ __forceinline__ int cell_of(const CellGrid& g, const float* __restrict__ p) {
    const int cx = cell_coord(p[0], g.ox, g.inv_cell, g.nx);
    const int cy = cell_coord(p[1], g.oy, g.inv_cell, g.ny);
    const int cz = g.dim == 3 ? cell_coord(p[2], g.oz, g.inv_cell, g.nz) : 0;
    return (cz * g.ny + cy) * g.nx + cx;
}
// the SAME arithmetic as the host-side exact search: separate roundings of the squares and of their sum, no FMA
__device__ __forceinline__ bool within(const float* __restrict__ q, const float* __restrict__ d, int dim, float r) {
    float s = 0.f;
    for (int k = 0; k < dim; ++k) {
        const float df = __fsub_rn(q[k], d[k]);
        s = __fadd_rn(s, __fmul_rn(df, df));
    }
    return __fsqrt_rn(s) <= r;
}

// torch_cluster.radius semantics (its CUDA kernel, what the reference's `torch_cluster` backend calls with the default
// max_num_neighbors = 32, neighbor_search.py:163-165): STRICT test on the squared distance accumulated with fused multiply-adds
__device__ __forceinline__ bool within_sq_strict(const float* __restrict__ q, const float* __restrict__ d, int dim, float r) {
    float s = 0.f;
    for (int k = 0; k < dim; ++k) {
        const float df = __fsub_rn(d[k], q[k]);
        s = __fmaf_rn(df, df, s);
    }
    return s < __fmul_rn(r, r);
}

__global__ void cell_count_kernel(const float* __restrict__ data, int n, CellGrid g, int* __restrict__ cell_id, int* __restrict__ cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = cell_of(g, data + (long)i * g.dim);
    cell_id[i] = c;
    atomicAdd(&cnt[c], 1);
}
__global__ void cell_fill_kernel(const int* __restrict__ cell_id, int n, const int* __restrict__ start, int* __restrict__ cnt,
                                 int* __restrict__ pts) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = cell_id[i];
    pts[start[c] + atomicSub(&cnt[c], 1) - 1] = i;
}
__global__ void zero_i32(int* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0;
}
__global__ __launch_bounds__(1024) void exscan_i32_kernel(const int* __restrict__ cnt, int n, int* __restrict__ out32,
                                                           int64_t* __restrict__ out64) {
    __shared__ long part[1024];
    const int t = threadIdx.x;
    const int chunk = (n + 1023) / 1024;
    const int b = t * chunk, e = min(n, b + chunk);
    long s = 0;
    for (int i = b; i < e; ++i) s += cnt[i];
    part[t] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const long v = (t >= off) ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    long run = (t == 0) ? 0 : part[t - 1];
    for (int i = b; i < e; ++i) {
        if (out32) out32[i] = (int)run;
        if (out64) out64[i] = run;
        run += cnt[i];
    }
    if (t == 1023) { if (out32) out32[n] = (int)part[1023]; if (out64) out64[n] = part[1023]; }
}

// one thread per query; FILL = false: count, FILL = true: write ascending neighbour indices at splits[q].
// cap > 0: keep only the `cap` SMALLEST data indices of a row (torch_cluster scans the data points in index order and stops at
// max_num_neighbors); strict: torch_cluster's `d^2 < r^2` instead of the in-repo backends' `d <= r`.
template <bool FILL>
__global__ void radius_query_kernel(const float* __restrict__ qry, int m, const float* __restrict__ data, CellGrid g, float r,
                                    const int* __restrict__ start, const int* __restrict__ pts, int* __restrict__ deg,
                                    const int64_t* __restrict__ splits, int64_t* __restrict__ index, int cap, int strict) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= m) return;
    const float* x = qry + (long)q * g.dim;
    // a query may lie outside the data's bounding box: clamp the CELL RANGE, not the point
    const int reach = 1;
    int lo[3], hi[3];
    const float o[3] = {g.ox, g.oy, g.oz};
    const int nn[3] = {g.nx, g.ny, g.nz};
    bool empty = false;
    for (int k = 0; k < 3; ++k) {
        if (k >= g.dim) { lo[k] = hi[k] = 0; continue; }
        const int c = (int)floorf((x[k] - o[k]) * g.inv_cell);
        lo[k] = max(c - reach, 0);
        hi[k] = min(c + reach, nn[k] - 1);
        if (lo[k] > hi[k]) empty = true;
    }
    int count = 0;
    int64_t base = FILL ? splits[q] : 0;
    if (!empty)
        for (int cz = lo[2]; cz <= hi[2]; ++cz)
            for (int cy = lo[1]; cy <= hi[1]; ++cy)
                for (int cx = lo[0]; cx <= hi[0]; ++cx) {
                    const int c = (cz * g.ny + cy) * g.nx + cx;
                    for (int t = start[c]; t < start[c + 1]; ++t) {
                        const int j = pts[t];
                        const float* dj = data + (long)j * g.dim;
                        if (strict ? within_sq_strict(x, dj, g.dim, r) : within(x, dj, g.dim, r)) {
                            if (FILL) {
                                if (cap > 0) {      // sorted insertion into the row's <= cap slots; a full row drops its largest index
                                    int k = count < cap ? count : cap - 1;
                                    if (count >= cap && index[base + k] <= j) continue;
                                    while (k > 0 && index[base + k - 1] > j) { index[base + k] = index[base + k - 1]; --k; }
                                    index[base + k] = j;
                                    if (count < cap) ++count;
                                    continue;
                                }
                                index[base + count] = j;
                            }
                            ++count;
                        }
                    }
                }
    if (!FILL) { deg[q] = (cap > 0 && count > cap) ? cap : count; return; }
    if (cap > 0) return;          // already in ascending order
    // ascending data index inside the segment (cells were visited in grid order)
    for (int i = 1; i < count; ++i) {
        const int64_t v = index[base + i];
        int k = i - 1;
        while (k >= 0 && index[base + k] > v) { index[base + k + 1] = index[base + k]; --k; }
        index[base + k + 1] = v;
    }
}

// One WAVE per query (few queries against many, unevenly spread points: the encoder side of an airfoil-like cloud has cells of
// ~1000 points around the body; a single thread walking 9 such cells serialises thousands of distance tests).  Lanes stride
// over the points of a cell; hits are counted with a ballot and, in the fill pass, placed by the lane's rank among the hits
// (deterministic); the row is sorted afterwards (sort_segments).  Uncapped, inclusive `dist <= r` semantics only.
template <bool FILL>
__global__ __launch_bounds__(256) void radius_query_wave_kernel(const float* __restrict__ qry, int m, const float* __restrict__ data,
                                                                CellGrid g, float r, const int* __restrict__ start,
                                                                const int* __restrict__ pts, int* __restrict__ deg,
                                                                const int64_t* __restrict__ splits, int64_t* __restrict__ index) {
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (q >= m) return;
    const float* x = qry + (long)q * g.dim;
    int lo[3], hi[3];
    const float o[3] = {g.ox, g.oy, g.oz};
    const int nn[3] = {g.nx, g.ny, g.nz};
    bool empty = false;
    for (int k = 0; k < 3; ++k) {
        if (k >= g.dim) { lo[k] = hi[k] = 0; continue; }
        const int c = (int)floorf((x[k] - o[k]) * g.inv_cell);
        lo[k] = max(c - 1, 0);
        hi[k] = min(c + 1, nn[k] - 1);
        if (lo[k] > hi[k]) empty = true;
    }
    int count = 0;
    const int64_t base = FILL ? splits[q] : 0;
    if (!empty)
        for (int cz = lo[2]; cz <= hi[2]; ++cz)
            for (int cy = lo[1]; cy <= hi[1]; ++cy)
                for (int cx = lo[0]; cx <= hi[0]; ++cx) {
                    const int c = (cz * g.ny + cy) * g.nx + cx;
                    const int t0 = start[c], t1 = start[c + 1];
                    for (int t = t0; t < t1; t += 64) {
                        const int tt = t + lane;
                        int j = -1;
                        bool hit = false;
                        if (tt < t1) { j = pts[tt]; hit = within(x, data + (long)j * g.dim, g.dim, r); }
                        const unsigned long long mask = __ballot(hit);
                        if (FILL && hit) index[base + count + __popcll(mask & ((1ull << lane) - 1ull))] = j;
                        count += __popcll(mask);
                    }
                }
    if (!FILL && lane == 0) deg[q] = count;
}

}  // namespace gaot

using namespace gaot;
#define ST(s) reinterpret_cast<hipStream_t>(s)

static int make_grid(CellGrid& g, int dim, const float* origin, float cell, const int32_t* dims) {
    g.dim = dim;
    g.ox = origin[0]; g.oy = origin[1]; g.oz = dim == 3 ? origin[2] : 0.f;
    g.inv_cell = 1.0f / cell;
    g.nx = dims[0]; g.ny = dims[1]; g.nz = dim == 3 ? dims[2] : 1;
    return 0;
}

// cell list of the data points.  origin[dim], dims[dim] are HOST arrays (the bounding box is computed by the caller);
// cell >= radius.  Device outputs: cell_start[ncell+1], cell_points[n]; scratch: 2 * n + ncell + 1 int32.
extern "C" int gaot_cells_build(const float* data, int32_t n, int32_t dim, const float* origin, float cell, const int32_t* dims,
                                int32_t* cell_start, int32_t* cell_points, int32_t* scratch, gaot_stream_t stream) {
    GAOT_REQUIRE(data && origin && dims && cell_start && cell_points && scratch, "cells_build: null pointer");
    GAOT_REQUIRE((dim == 2 || dim == 3) && n > 0 && cell > 0.f, "cells_build: bad arguments");
    CellGrid g;
    make_grid(g, dim, origin, cell, dims);
    const long ncell = (long)g.nx * g.ny * g.nz;
    GAOT_REQUIRE(ncell > 0 && ncell < (1L << 30), "cells_build: %ld cells", ncell);
    int* cell_id = scratch;
    int* cnt = scratch + n;
    hipLaunchKernelGGL(zero_i32, dim3(cdiv(ncell + 1, 256)), dim3(256), 0, ST(stream), cnt, (int)ncell + 1);
    hipLaunchKernelGGL(cell_count_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ST(stream), data, n, g, cell_id, cnt);
    hipLaunchKernelGGL(exscan_i32_kernel, dim3(1), dim3(1024), 0, ST(stream), cnt, (int)ncell, cell_start, (int64_t*)nullptr);
    hipLaunchKernelGGL(cell_fill_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ST(stream), cell_id, n, cell_start, cnt, cell_points);
    sort_segments<int, int>(cell_start, (int)ncell, cell_points, scratch + n + ncell + 1, ST(stream));   // ascending point ids per cell
    GAOT_CHECK_LAUNCH("gaot_cells_build");
    return GAOT_OK;
}

// a wave per query pays when queries are few and their neighbourhoods can be large: more data points than queries
static inline bool wave_per_query(int m, int n_data, int cap) { return cap == 0 && n_data >= m && (long)m * 64 <= (1L << 27); }

// degree per query + row splits (int64, [m+1]).  The caller reads splits[m] to size the index array.
extern "C" int gaot_radius_count(const float* queries, int32_t m, const float* data, int32_t n_data, int32_t dim, float radius,
                                 const float* origin, float cell, const int32_t* dims, const int32_t* cell_start,
                                 const int32_t* cell_points, int32_t* deg, int64_t* splits, int32_t max_neighbors, int32_t strict,
                                 gaot_stream_t stream) {
    GAOT_REQUIRE(queries && data && origin && dims && cell_start && cell_points && deg && splits, "radius_count: null pointer");
    GAOT_REQUIRE(m > 0 && radius >= 0.f && cell >= radius && max_neighbors >= 0, "radius_count: need cell >= radius, max_neighbors >= 0");
    CellGrid g;
    make_grid(g, dim, origin, cell, dims);
    if (wave_per_query(m, n_data, max_neighbors) && !strict)
        hipLaunchKernelGGL(radius_query_wave_kernel<false>, dim3(cdiv(m, 4)), dim3(256), 0, ST(stream), queries, m, data, g, radius,
                           cell_start, cell_points, deg, (const int64_t*)nullptr, (int64_t*)nullptr);
    else
        hipLaunchKernelGGL(radius_query_kernel<false>, dim3(cdiv(m, 128)), dim3(128), 0, ST(stream), queries, m, data, g, radius,
                           cell_start, cell_points, deg, (const int64_t*)nullptr, (int64_t*)nullptr, (int)max_neighbors, (int)strict);
    hipLaunchKernelGGL(exscan_i32_kernel, dim3(1), dim3(1024), 0, ST(stream), deg, m, (int*)nullptr, splits);
    GAOT_CHECK_LAUNCH("gaot_radius_count");
    return GAOT_OK;
}

// scratch: E int64 (only rows longer than 4096 neighbours use it; may be NULL when the caller knows the maximum degree is smaller)
extern "C" int gaot_radius_fill(const float* queries, int32_t m, const float* data, int32_t n_data, int32_t dim, float radius,
                                const float* origin, float cell, const int32_t* dims, const int32_t* cell_start,
                                const int32_t* cell_points, const int64_t* splits, int64_t* index, int64_t* scratch,
                                int32_t max_neighbors, int32_t strict, gaot_stream_t stream) {
    GAOT_REQUIRE(queries && data && origin && dims && cell_start && cell_points && splits && max_neighbors >= 0, "radius_fill: bad arguments");
    CellGrid g;
    make_grid(g, dim, origin, cell, dims);
    if (wave_per_query(m, n_data, max_neighbors) && !strict) {
        hipLaunchKernelGGL(radius_query_wave_kernel<true>, dim3(cdiv(m, 4)), dim3(256), 0, ST(stream), queries, m, data, g, radius,
                           cell_start, cell_points, (int*)nullptr, splits, index);
        sort_segments<int64_t, int64_t>(splits, m, index, scratch, ST(stream));       // ascending data index inside every row
    } else {
        hipLaunchKernelGGL(radius_query_kernel<true>, dim3(cdiv(m, 128)), dim3(128), 0, ST(stream), queries, m, data, g, radius,
                           cell_start, cell_points, (int*)nullptr, splits, index, (int)max_neighbors, (int)strict);
    }
    GAOT_CHECK_LAUNCH("gaot_radius_fill");
    return GAOT_OK;
}
