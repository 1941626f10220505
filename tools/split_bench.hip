// Stand-alone ablation bench of the split-bf16 GEMM kernel (tuning tool, not part of the library).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Igaot_amd/csrc tools/split_bench.hip -o tools/bin/split_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../gaot_amd/csrc/gemm_split.hip"
namespace gaot { static thread_local char g_err[512]; void set_error(const char*, ...) {} }
using namespace gaot;

template <bool AK, bool BKM, int ABL>
static float run(GemmArgs a, int iters) {
    a.tiles_m = cdiv(a.M, S_BM); a.tiles_n = cdiv(a.N, S_BN);
    dim3 grid(a.tiles_m * a.tiles_n, 1, a.split_k > 1 ? a.split_k : 1), block(256);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((gemm_split_kernel<AK, BKM, 128, ABL>), grid, block, 0, 0, a);
    hipEventRecord(s, 0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((gemm_split_kernel<AK, BKM, 128, ABL>), grid, block, 0, 0, a);
    hipEventRecord(e, 0); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    return ms * 1e3f / iters;
}

template <bool AK, bool BKM>
static void sweep(const char* kind, int M, int N, int K, int split) {
    float *A, *B, *C, *ws;
    hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&B, (size_t)N * K * 4); hipMalloc(&C, (size_t)M * N * 4);
    hipMalloc(&ws, (size_t)(split > 1 ? split : 1) * ((size_t)M * N + M) * 4);
    std::vector<float> h((size_t)(M > N ? M : N) * K);
    for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(A, h.data(), (size_t)M * K * 4, hipMemcpyHostToDevice); hipMemcpy(B, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice);
    GemmArgs a{}; a.M = M; a.N = N; a.K = K; a.A = A; a.lda = AK ? K : M; a.B = B; a.ldb = BKM ? K : N; a.C = C; a.ldc = N;
    a.split_k = split; a.ktiles_per_split = cdiv(cdiv(K, 32), split); a.ws = ws; a.vec_epi = 1; a.rb_period = 1;
    const double gf = 2.0 * M * N * K * 1e-6;
    float t0 = run<AK, BKM, 0>(a, 20), t1 = run<AK, BKM, 1>(a, 20), t2 = run<AK, BKM, 2>(a, 20), t4 = run<AK, BKM, 4>(a, 20),
          t8 = run<AK, BKM, 8>(a, 20), t16 = run<AK, BKM, 16>(a, 20), t3 = run<AK, BKM, 3>(a, 20), t27 = run<AK, BKM, 27>(a, 20), t31 = run<AK, BKM, 31>(a, 20),
          t63 = run<AK, BKM, 63>(a, 20), t127 = run<AK, BKM, 127>(a, 20), t15 = run<AK, BKM, 15>(a, 20), t47 = run<AK, BKM, 47>(a, 20), t6 = run<AK, BKM, 6>(a, 20);
    printf("%s M=%d N=%d K=%d sk=%d | full %.1fus %.0fTF | -split %.1f | -mfma %.1f | -stores %.1f | -ldsread %.1f | -gload %.1f | -split-mfma %.1f | only stores %.1f | nothing %.1f"
           " | nothing-ldswrite %.1f | nothing-ldswrite-barrier %.1f | loads only (no split/mfma/epilogue/ldsread) %.1f | loads only, no lds write %.1f | -mfma-epilogue %.1f\n",
           kind, M, N, K, split, t0, gf / t0, t1, t2, t4, t8, t16, t3, t27, t31, t63, t127, t15, t47, t6);
    hipFree(A); hipFree(B); hipFree(C); hipFree(ws);
}

template <bool AK, bool BKM>
static void dephase(const char* kind, int M, int N, int K) {
    float *A, *B, *C;
    hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&B, (size_t)N * K * 4); hipMalloc(&C, (size_t)M * N * 4);
    hipMemset(A, 0, (size_t)M * K * 4); hipMemset(B, 0, (size_t)N * K * 4);
    GemmArgs a{}; a.M = M; a.N = N; a.K = K; a.A = A; a.lda = AK ? K : M; a.B = B; a.ldb = BKM ? K : N; a.C = C; a.ldc = N;
    a.split_k = 1; a.ktiles_per_split = cdiv(K, 32); a.vec_epi = 1; a.rb_period = 1;
    printf("%s M=%d N=%d K=%d de-phasing half of the workgroups at start:", kind, M, N, K);
    printf(" none %.1fus", run<AK, BKM, 0>(a, 20));
    for (int d = 1; d <= 8; ++d) { a.ablate = d; printf(" | %d x 3.4us -> %.1f", d, run<AK, BKM, 128>(a, 20)); }
    printf("\n");
    hipFree(A); hipFree(B); hipFree(C);
}

int main() {
    dephase<true, true>("nt", 8192, 2048, 256);
    dephase<true, true>("nt", 8192, 768, 256);
    dephase<true, false>("nn", 8192, 1024, 256);
    return 0;
    sweep<true, true>("nt", 4096, 4096, 4096, 1);
    sweep<true, true>("nt", 8192, 2048, 256, 1);
    sweep<true, true>("nt", 8192, 256, 1024, 1);
    sweep<true, true>("nt", 8192, 256, 256, 1);
    sweep<true, false>("nn", 8192, 1024, 256, 1);
    sweep<true, false>("nn", 8192, 256, 2048, 1);
    sweep<false, false>("tn", 2048, 256, 8192, 16);
    sweep<false, false>("tn", 256, 1024, 8192, 32);
    return 0;
}
