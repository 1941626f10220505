// Stand-alone bench + check of the A-direct fp16-piece tiles (gemm_ad.hip) against the LDS-staged ones (gemm_split.hip): same planes,
// same words; the two must agree BIT FOR BIT (same piece products in the same order), and both are compared with a float64 product.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Igaot_amd/csrc tools/ad_bench.hip -o tools/bin/ad_bench
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../gaot_amd/csrc/gemm_split.hip"
#include "../gaot_amd/csrc/gemm_ad.hip"
namespace gaot { void set_error(const char*, ...) {} }
using namespace gaot;

__global__ void ref_kernel(const float* A, const float* W, double* C, int M, int N, int K) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= N) return;
    double s = 0.0;
    for (int k = 0; k < K; ++k) s += (double)A[(long)m * K + k] * (double)W[(long)n * K + k];
    C[(long)m * N + n] = s;
}

static float amax_of(const std::vector<float>& v) { float m = 0.f; for (float x : v) m = fmaxf(m, fabsf(x)); return m; }
static float scale_of(float amax) { unsigned b; memcpy(&b, &amax, 4); const int e = (b >> 23) & 0xff; int se = 140 - e; se = se > 126 ? 126 : (se < -126 ? -126 : se); return ldexpf(1.f, se); }

struct Case { int M, N, K, bm, act; const char* name; int sk = 1; int bm_old = 0; };

static void run_case(const Case& c, int iters) {
    const int M = c.M, N = c.N, K = c.K;
    std::vector<float> hA((size_t)M * K), hW((size_t)N * K);
    for (auto& v : hA) v = ((float)rand() / RAND_MAX - 0.5f) * 4.f;
    for (auto& v : hW) v = ((float)rand() / RAND_MAX - 0.5f) * 0.1f;
    const float amA = amax_of(hA), amW = amax_of(hW), scW = scale_of(amW);
    std::vector<unsigned short> hP((size_t)N * K * 2);
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) {
            const float v = scW * hW[(size_t)n * K + k];
            const _Float16 h = (_Float16)v; const float r = v - (float)h; const _Float16 m = (_Float16)r;
            unsigned short hb, mb; memcpy(&hb, &h, 2); memcpy(&mb, &m, 2);
            hP[(size_t)n * 2 * K + (k / 16) * 32 + k % 16] = hb;
            hP[(size_t)n * 2 * K + (k / 16) * 32 + 16 + k % 16] = mb;
        }
    std::vector<float> wordA(1024, 0.f), wordW(1024, 0.f);
    wordA[0] = amA; wordW[0] = amW;
    float *A, *W, *C0, *C1, *wa, *ww, *aux; unsigned short* P; double* R;
    const int NO = c.act == GAOT_ACT_SWIGLU ? N / 2 : (c.act == GAOT_ACT_SWIGLU_BWD ? 2 * N : N);
    hipMalloc(&A, hA.size() * 4); hipMalloc(&W, hW.size() * 4); hipMalloc(&P, hP.size() * 2);
    hipMalloc(&C0, (size_t)M * N * 8); hipMalloc(&C1, (size_t)M * N * 8); hipMalloc(&R, (size_t)M * N * 8); hipMalloc(&aux, (size_t)M * N * 8);
    hipMalloc(&wa, 4096); hipMalloc(&ww, 4096);
    hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice); hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(P, hP.data(), hP.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(wa, wordA.data(), 4096, hipMemcpyHostToDevice); hipMemcpy(ww, wordW.data(), 4096, hipMemcpyHostToDevice);
    hipMemset(C0, 0, (size_t)M * N * 4); hipMemset(C1, 0xff, (size_t)M * N * 4);
    GemmArgs a{}; a.M = M; a.N = N; a.K = K; a.A = A; a.lda = K; a.B = W; a.ldb = K; a.ldc = NO;
    a.split_k = c.sk; a.ktiles_per_split = cdiv(K / 32, c.sk); a.vec_epi = 1; a.rb_period = 1; a.act = c.act;
    float* ws = nullptr; if (c.sk > 1) { hipMalloc(&ws, (size_t)c.sk * ((size_t)M * N + M) * 4); a.ws = ws; }
    a.a_amax = wa; a.b_amax = ww; a.Bpl = P; a.ld_bpl = 2 * K; a.bpl_stride = 16;
    if (c.act == GAOT_ACT_SWIGLU) { a.aux_out = aux; a.ld_aux = N; }
    if (c.act == GAOT_ACT_SWIGLU_BWD) { std::vector<float> hu((size_t)M * 2 * N); for (auto& v : hu) v = ((float)rand() / RAND_MAX - 0.5f) * 4.f; hipMemcpy(aux, hu.data(), hu.size() * 4, hipMemcpyHostToDevice); a.aux_in = aux; a.ld_aux = 2 * N; }
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    float t_old, t_new;
    {
        GemmArgs b = a; b.C = C0;
        const int bmo = c.bm_old ? c.bm_old : c.bm;
        for (int i = 0; i < 5; ++i) launch_split(b, true, true, 0, bmo, 4);
        hipEventRecord(s, 0);
        for (int i = 0; i < iters; ++i) launch_split(b, true, true, 0, bmo, 4);
        hipEventRecord(e, 0); hipEventSynchronize(e); hipEventElapsedTime(&t_old, s, e);
        if (c.sk > 1) hipMemcpy(C0, ws + (size_t)M * N, (size_t)M * N * 4, hipMemcpyDeviceToDevice);
    }
    {
        GemmArgs b = a; b.C = C1;
        for (int i = 0; i < 5; ++i) launch_ad(b, true, 0, c.bm);
        hipEventRecord(s, 0);
        for (int i = 0; i < iters; ++i) launch_ad(b, true, 0, c.bm);
        hipEventRecord(e, 0); hipEventSynchronize(e); hipEventElapsedTime(&t_new, s, e);
        if (c.sk > 1) hipMemcpy(C1, ws + (size_t)M * N, (size_t)M * N * 4, hipMemcpyDeviceToDevice);
    }
    float t_nd;
    {
        GemmArgs b = a; b.C = C1; b.tiles_m = cdiv(b.M, c.bm); b.tiles_n = cdiv(b.N, 128); b.bpl_flag = 1;
        auto go = [&]() { launch_ad(b, true, 0, 64, 64); };
        for (int i = 0; i < 5; ++i) go();
        hipEventRecord(s, 0);
        for (int i = 0; i < iters; ++i) go();
        hipEventRecord(e, 0); hipEventSynchronize(e); hipEventElapsedTime(&t_nd, s, e);
    }
    hipError_t err = hipDeviceSynchronize();
    std::vector<float> h0((size_t)M * NO), h1((size_t)M * NO);
    hipMemcpy(h0.data(), C0, h0.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(h1.data(), C1, h1.size() * 4, hipMemcpyDeviceToHost);
    size_t diff = 0; for (size_t i = 0; i < h0.size(); ++i) diff += memcmp(&h0[i], &h1[i], 4) != 0;
    double e_old = -1, e_new = -1;
    if (c.act == 0 && c.sk == 1) {
        ref_kernel<<<dim3((N + 255) / 256, M), 256>>>(A, W, R, M, N, K);
        std::vector<double> hr((size_t)M * N);
        hipMemcpy(hr.data(), R, hr.size() * 8, hipMemcpyDeviceToHost);
        double mx = 0, d0 = 0, d1 = 0;
        for (size_t i = 0; i < hr.size(); ++i) { mx = fmax(mx, fabs(hr[i])); d0 = fmax(d0, fabs(h0[i] - hr[i])); d1 = fmax(d1, fabs(h1[i] - hr[i])); }
        e_old = d0 / mx; e_new = d1 / mx;
    }
    const double gf = 2.0 * M * N * K * 1e-6;
    printf("%-10s M=%5d N=%5d K=%5d bm=%3d act=%d | staged %.1f us (%.0f TF) | direct %.1f us (%.0f TF) | x%.2f | all-DMA 64 x 64 %.1f us | differing outputs %zu of %zu | err vs f64: staged %.2e direct %.2e | redo %u/%u | %s\n",
           c.name, M, N, K, c.bm, c.act, t_old * 1e3 / iters, gf / (t_old * 1e3 / iters), t_new * 1e3 / iters, gf / (t_new * 1e3 / iters), t_old / t_new, t_nd * 1e3 / iters,
           diff, h0.size(), e_old, e_new, split_redo_count(true), ad_redo_count(true), hipGetErrorString(err));
    hipFree(A); hipFree(W); hipFree(P); hipFree(C0); hipFree(C1); hipFree(R); hipFree(aux); hipFree(wa); hipFree(ww);
}

template <int ABL>
static float run_abl(GemmArgs a, int bm, int iters) {
    a.tiles_m = cdiv(a.M, bm); a.tiles_n = cdiv(a.N, 128); a.bpl_flag = 1;
    dim3 grid(a.tiles_m * a.tiles_n), block(256);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    auto go = [&]() { if (bm == 64) hipLaunchKernelGGL((gemm_ad_kernel<64, 2, 3, true, ABL>), grid, block, 0, 0, a); else hipLaunchKernelGGL((gemm_ad_kernel<128, 4, 2, true, ABL>), grid, block, 0, 0, a); };
    for (int i = 0; i < 3; ++i) go();
    hipEventRecord(s, 0);
    for (int i = 0; i < iters; ++i) go();
    hipEventRecord(e, 0); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    return ms * 1e3f / iters;
}
static void ablate(int M, int N, int K, int bm, int iters) {
    float *A, *W, *C, *wa, *ww; unsigned short* P;
    hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&W, (size_t)N * K * 4); hipMalloc(&P, (size_t)N * K * 4); hipMalloc(&C, (size_t)M * N * 4);
    hipMalloc(&wa, 4096); hipMalloc(&ww, 4096);
    std::vector<float> h((size_t)M * K); for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    std::vector<unsigned short> hp((size_t)N * K * 2); for (auto& v : hp) v = 0x3800 + (rand() & 0x3ff);
    hipMemcpy(P, hp.data(), hp.size() * 2, hipMemcpyHostToDevice);
    std::vector<float> word(1024, 0.f); word[0] = 0.5f;
    hipMemcpy(wa, word.data(), 4096, hipMemcpyHostToDevice); hipMemcpy(ww, word.data(), 4096, hipMemcpyHostToDevice);
    GemmArgs a{}; a.M = M; a.N = N; a.K = K; a.A = A; a.lda = K; a.B = W; a.ldb = K; a.C = C; a.ldc = N;
    a.split_k = 1; a.ktiles_per_split = K / 32; a.vec_epi = 1; a.rb_period = 1;
    a.a_amax = wa; a.b_amax = ww; a.Bpl = P; a.ld_bpl = 2 * K; a.bpl_stride = 16;
    printf("ablate M=%d N=%d K=%d bm=%d | full(no detect) %.1f | -split %.1f | -mfma %.1f | -epilogue %.1f | -dma %.1f | -dma-epi %.1f | -dma-epi-split-mfma %.1f | -split-mfma %.1f\n", M, N, K, bm,
           run_abl<32>(a, bm, iters), run_abl<33>(a, bm, iters), run_abl<34>(a, bm, iters), run_abl<36>(a, bm, iters), run_abl<40>(a, bm, iters),
           run_abl<44>(a, bm, iters), run_abl<47>(a, bm, iters), run_abl<35>(a, bm, iters));
    hipFree(A); hipFree(W); hipFree(P); hipFree(C); hipFree(wa); hipFree(ww);
}

int main(int argc, char** argv) {
    if (argc > 2) {
        ablate(8192, 2048, 256, 128, 30);
        ablate(8192, 768, 256, 64, 30);
        ablate(8192, 256, 256, 64, 30);
        ablate(8192, 256, 2048, 64, 30);
        return 0;
    }
    const int iters = argc > 1 ? atoi(argv[1]) : 50;
    const Case cases[] = {
        {8192, 2048, 256, 128, GAOT_ACT_SWIGLU, "w1w3"},
        {8192, 2048, 256, 128, 0, "w1w3-lin"},
        {8192, 1024, 256, 128, GAOT_ACT_SWIGLU_BWD, "w2b-swiglu"},
        {8192, 2048, 256, 64, GAOT_ACT_SWIGLU, "w1w3-64"},
        {8192, 2048, 256, 64, 0, "w1w3-lin64"},
        {8192, 768, 256, 64, 0, "qkv"},
        {8192, 768, 256, 128, 0, "qkv"},
        {8192, 256, 256, 64, 0, "o_proj"},
        {8192, 256, 1024, 64, 0, "w2"},
        {8192, 1024, 256, 128, 0, "w2-bwd"},
        {8192, 256, 2048, 64, 0, "w13-bwd"},
        {8192, 256, 768, 64, 0, "qkv-bwd"},
        {4096, 2048, 256, 128, 0, "c4-w1w3"},
        {4096, 256, 256, 64, 0, "c4-o"},
        {8000, 200, 96, 64, 0, "ragged"},
        {300, 136, 32, 128, 0, "tiny"},
        {8192, 256, 2048, 64, 0, "w13b-sk2", 2, 128},
        {8192, 256, 2048, 64, 0, "w13b-sk2", 2, 64},
        {8192, 256, 2048, 64, 0, "w13b-sk4", 4, 128},
        {8192, 256, 1024, 64, 0, "w2-sk2", 2, 64},
        {8192, 768, 256, 64, 0, "qkv", 1, 64},
    };
    for (const Case& c : cases) run_case(c, iters);
    return 0;
}
