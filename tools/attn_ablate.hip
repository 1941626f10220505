// Stand-alone ablation bench of the pipelined split-bf16 attention forward (tuning tool, not part of the library).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Igaot_amd/csrc tools/attn_ablate.hip -o tools/bin/attn_ablate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../gaot_amd/csrc/attention.hip"
namespace gaot { void set_error(const char*, ...) {} }

template <int ABL>
static float run(const AttnArgs& a, int iters) {
    dim3 grid(cdiv(a.S, 256) * a.B * a.H), block(512);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(attn_fwd_split_pipe_kernel<ABL>, grid, block, 0, 0, a);
    hipEventRecord(s, 0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(attn_fwd_split_pipe_kernel<ABL>, grid, block, 0, 0, a);
    hipEventRecord(e, 0); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    return ms * 1e3f / iters;
}

int main() {
    const int B = 8, S = 1024, H = 8, D = 32;
    float *qkv, *o, *lse;
    hipMalloc(&qkv, (size_t)B * S * 3 * H * D * 4); hipMalloc(&o, (size_t)B * S * H * D * 4); hipMalloc(&lse, (size_t)B * H * S * 4);
    std::vector<float> h((size_t)B * S * 3 * H * D);
    for (auto& v : h) v = ((float)rand() / RAND_MAX - 0.5f) * 2.f;
    hipMemcpy(qkv, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    AttnArgs a{};
    a.q = qkv; a.k = qkv + H * D; a.v = qkv + 2 * H * D; a.ldq = a.ldk = a.ldv = 3 * H * D;
    a.B = B; a.S = S; a.H = H; a.Hkv = H; a.D = D; a.scale = 1.0f / sqrtf((float)D); a.vec = 1;
    a.o = o; a.ldo = H * D; a.lse = lse;
    printf("full %.1f us | -S mfma %.1f | -PV mfma %.1f | -all mfma %.1f | -P split %.1f | -staging %.1f | -exp %.1f | -barrier %.1f"
           " | -split-exp %.1f | -mfma-split-exp %.1f | -mfma-split-exp-staging %.1f | -everything %.1f | -split-exp-staging %.1f\n",
           run<0>(a, 20), run<1>(a, 20), run<2>(a, 20), run<3>(a, 20), run<4>(a, 20), run<8>(a, 20), run<16>(a, 20), run<32>(a, 20),
           run<20>(a, 20), run<23>(a, 20), run<31>(a, 20), run<63>(a, 20), run<28>(a, 20));
    return 0;
}
