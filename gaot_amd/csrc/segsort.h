// Ascending sort of every segment [start[s], start[s+1]) of an array of DISTINCT integers (edge ids of a transposed CSR
// row, point ids of a grid cell, neighbour ids of a query).  Once-per-geometry work, but on skewed meshes a few segments are
// hundreds to thousands long (an airfoil-like cloud against a 64x64 latent grid: transposed rows of ~350 edges, cells of
// ~1000 points), where one thread per segment with an insertion sort costs milliseconds.
//   short segments (<= SEGSORT_SHORT): one thread each, insertion sort in place;
//   long segments: one workgroup each, RANK sort -- the segment is staged in LDS, every thread counts for its elements how
//   many are smaller (LDS broadcast reads) and writes them to their final position.  O(n^2 / 256) per thread, deterministic,
//   no atomics.  Segments beyond the LDS stage (32 KB) rank against global memory and go through `scratch`.
#pragma once
#include "common.h"

namespace gaot {

constexpr int SEGSORT_SHORT = 32;
constexpr int SEGSORT_LDS_BYTES = 32768;      // LDS stage of a long segment: 8192 int32 / 4096 int64

template <typename T, typename S>
__global__ void segsort_short_kernel(const S* __restrict__ start, int nseg, T* __restrict__ v) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nseg) return;
    const long b = (long)start[s], e = (long)start[s + 1];
    if (e - b > SEGSORT_SHORT) return;
    for (long i = b + 1; i < e; ++i) {
        const T x = v[i];
        long k = i - 1;
        while (k >= b && v[k] > x) { v[k + 1] = v[k]; --k; }
        v[k + 1] = x;
    }
}

template <typename T, typename S>
__global__ __launch_bounds__(256) void segsort_long_kernel(const S* __restrict__ start, int nseg, T* __restrict__ v, T* __restrict__ scratch) {
    constexpr int SEGSORT_LDS = SEGSORT_LDS_BYTES / (int)sizeof(T);
    __shared__ T stage[SEGSORT_LDS];
    for (int s = blockIdx.x; s < nseg; s += gridDim.x) {
        const long b = (long)start[s], e = (long)start[s + 1];
        const long n = e - b;
        if (n <= SEGSORT_SHORT) continue;
        if (n <= SEGSORT_LDS) {
            for (long i = threadIdx.x; i < n; i += 256) stage[i] = v[b + i];
            __syncthreads();
            for (long i = threadIdx.x; i < n; i += 256) {
                const T x = stage[i];
                int rank = 0;
                for (long k = 0; k < n; ++k) rank += stage[k] < x ? 1 : 0;
                v[b + rank] = x;
            }
            __syncthreads();
        } else {          // beyond the LDS stage: rank against global memory, out of place
            for (long i = threadIdx.x; i < n; i += 256) {
                const T x = v[b + i];
                long rank = 0;
                for (long k = 0; k < n; ++k) rank += v[b + k] < x ? 1 : 0;
                scratch[b + rank] = x;
            }
            __syncthreads();
            for (long i = threadIdx.x; i < n; i += 256) v[b + i] = scratch[b + i];
            __syncthreads();
        }
    }
}

// scratch: as long as v (only touched by segments longer than the LDS stage; may be nullptr if the caller knows there are none)
template <typename T, typename S>
static inline void sort_segments(const S* start, int nseg, T* v, T* scratch, hipStream_t st) {
    if (nseg <= 0) return;
    hipLaunchKernelGGL((segsort_short_kernel<T, S>), dim3(cdiv(nseg, 256)), dim3(256), 0, st, start, nseg, v);
    const int nb = nseg < 4096 ? nseg : 4096;
    hipLaunchKernelGGL((segsort_long_kernel<T, S>), dim3(nb), dim3(256), 0, st, start, nseg, v, scratch);
}

}  // namespace gaot
