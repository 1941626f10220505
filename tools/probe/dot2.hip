// v_dot2c_f32_bf16 probe: what does it compute?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__global__ void k(const unsigned* a, const unsigned* b, const float* c, float* out) {
    const int i = threadIdx.x;
    out[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a[i]), __builtin_bit_cast(bf16x2_t, b[i]), c[i], false);
}
static unsigned pk(float lo, float hi) { unsigned l, h; memcpy(&l, &lo, 4); memcpy(&h, &hi, 4); return (l >> 16) | (h & 0xffff0000u); }
int main() {
    const int n = 6;
    float lo[n] = {1.f, 2.f, -3.f, 0.5f, 1.5f, 100.f}, hi[n] = {10.f, 20.f, 30.f, -0.25f, 2.5f, -100.f}, c[n] = {0.f, 1.f, 0.f, 0.f, 1000.f, 0.125f};
    unsigned ha[n], hb[n]; for (int i = 0; i < n; ++i) { ha[i] = pk(lo[i], hi[i]); hb[i] = pk(1.f, 1.f); }
    unsigned *da, *db; float *dc, *dout, ho[n];
    hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dc, n * 4); hipMalloc(&dout, n * 4);
    hipMemcpy(da, ha, n * 4, hipMemcpyHostToDevice); hipMemcpy(db, hb, n * 4, hipMemcpyHostToDevice); hipMemcpy(dc, c, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(n), 0, 0, da, db, dc, dout);
    hipMemcpy(ho, dout, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) printf("lo %g hi %g c %g -> %g (expect %g)\n", lo[i], hi[i], c[i], ho[i], lo[i] + hi[i] + c[i]);
    return 0;
}
