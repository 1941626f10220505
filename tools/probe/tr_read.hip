// ds_read_b64_tr_b16 semantics probe: LDS tile[r][c] = r * 64 + c (as 16-bit ints), 64 B rows (32 elements).
// Each lane supplies &tile[rowsel][colsel]; prints what every lane gets.  Build: hipcc --offload-arch=gfx950 -o tr_read tr_read.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(int mode, int* out) {
    __shared__ unsigned short tile[64][32];
    for (int i = threadIdx.x; i < 64 * 32; i += 64) tile[i / 32][i % 32] = (unsigned short)((i / 32) * 64 + (i % 32));
    __syncthreads();
    const int l = threadIdx.x, g = l >> 4, t = l & 15;
    int r, c;
    if (mode == 0) { r = (t >> 2); c = (t & 3) * 4 + g * 0; }          // lane t of a group: row t/4, col chunk t%4 (block of 4 rows x 16 cols), all groups same block
    else if (mode == 1) { r = t & 3; c = (t >> 2) * 4; }               // alternative: row t%4, chunk t/4
    else { r = 4 * g + (t >> 2); c = (t & 3) * 4; }                    // per-group row block
    // low 32 bits of a __shared__ object's flat address are its LDS offset; deriving the operand from the pointer also keeps
    // the fill above alive (with a bare integer offset the compiler deleted the tile: 0 ds_writes, group segment 0)
    const unsigned addr = (unsigned)(size_t)(&tile[r][c]);
    s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}
int main() {
    int* d; hipMalloc(&d, 64 * 4 * sizeof(int));
    int h[256];
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, mode, d);
        hipError_t e1 = hipGetLastError();
        hipError_t e2 = hipDeviceSynchronize();
        hipError_t e3 = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("errors: %s | %s | %s\n", hipGetErrorString(e1), hipGetErrorString(e2), hipGetErrorString(e3));
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) {
            printf(" %2d:", l);
            for (int j = 0; j < 4; ++j) printf("%d.%d ", h[l * 4 + j] / 64, h[l * 4 + j] % 64);
            if (l % 8 == 7) printf("\n");
        }
    }
    return 0;
}
