// Probe: the second fp16 piece of common.h split2h_pair through v_fma_mixlo_f16 / v_fma_mixhi_f16 (m = rn16(y - (float)h) in ONE instruction per
// element instead of v_cvt_f32_f16 + v_sub_f32 (+ a shared v_cvt_pk_f16_f32)): bit-identical to the five-instruction form?
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I gaot_amd/csrc tools/probe/mix_split.hip -o tools/bin/mix_split
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include "common.h"
using namespace gaot;
__device__ __forceinline__ void split2h_pair_mix(float x0, float x1, float sc, unsigned& ph, unsigned& pm) {
    const f32x2 x = {mul_scalar(x0, sc), mul_scalar(x1, sc)};
    const f16x2_t h = __builtin_convertvector(x, f16x2_t);
    ph = __builtin_bit_cast(unsigned, h);
    unsigned m = 0u;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(m) : "v"(ph), "v"(x[0]));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(m) : "v"(ph), "v"(x[1]));
    pm = m;
}
__global__ void probe(const float* in, int n, float sc, unsigned* out, int* bad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    unsigned a, b, c, d;
    split2h_pair(in[2 * i], in[2 * i + 1], sc, a, b);
    split2h_pair_mix(in[2 * i], in[2 * i + 1], sc, c, d);
    out[4 * i] = a; out[4 * i + 1] = b; out[4 * i + 2] = c; out[4 * i + 3] = d;
    if (a != c || b != d) atomicAdd(bad, 1);
}
int main() {
    const int n = 1 << 22;
    float* h = (float*)malloc(n * 4);
    srand(1);
    for (int i = 0; i < n; ++i) {
        const int kind = i & 7;
        float v = (float)rand() / RAND_MAX * 2.f - 1.f;
        if (kind == 1) v *= 1e-4f; if (kind == 2) v *= 1e-7f; if (kind == 3) v *= 3.9f; if (kind == 4) v = ldexpf(v, -(rand() % 40)); if (kind == 5) v = (rand() & 1) ? 0.f : -0.f;
        if (kind == 6) { unsigned u = ((unsigned)rand() << 16) ^ (unsigned)rand(); u &= 0xbfffffffu; memcpy(&v, &u, 4); if (!std::isfinite(v) || fabsf(v) > 3.9f) v = 0.5f; }
        h[i] = v;
    }
    float* d; unsigned* o; int* bad;
    hipMalloc(&d, n * 4); hipMalloc(&o, n * 8); hipMalloc(&bad, 4);
    hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
    for (float sc : {8192.f, 1.f, 16384.f, 1e-3f, 4096.f * 4096.f}) {
        hipMemset(bad, 0, 4);
        hipLaunchKernelGGL(probe, dim3(n / 2 / 256), dim3(256), 0, 0, d, n, sc, o, bad);
        int nb; hipMemcpy(&nb, bad, 4, hipMemcpyDeviceToHost);
        printf("scale %g: %d of %d pairs differ\n", sc, nb, n / 2);
    }
    return 0;
}
