// Edge-partitioned integral-transform kernels with segmented reductions (the skew-proof forms of lift_gather_reduce and
// proj_gather_t in gno.hip).
//
// The row-parallel kernels give one lane group to one CSR row.  On a degree-skewed mesh (BASELINE config 3: airfoil-like
// clouds, rows of 350+ edges next to thousands of empty rows) that serialises the long rows and idles the lanes of the empty
// ones: profiles/r2e_pmc_c3.json, ~1 TB/s on a kernel whose operands are streamed exactly once.
// Here the EDGE list is cut into equal chunks of EPC edges; one lane group (C/4 lanes, 16 B per lane along the channels) owns one
// chunk and walks it in order, keeping a running sum per batch sample; when the row id changes the sum is flushed:
//   * a row that lies completely inside the chunk is finished and stored (with the fused epilogue);
//   * the first / last row of a chunk may continue in a neighbouring chunk: its partial sum goes to a carry slot
//     ws[chunk][0 = row began in an earlier chunk | 1 = row continues in the next chunk][b][C];
// and a second, tiny row-parallel kernel adds the carries of the rows that span chunks, in chunk order (deterministic, no
// atomics), and zero-fills the empty rows.  Every lane group does the same amount of work whatever the degree distribution.
#include "common.h"

namespace gaot {

static inline int ep_pow2_ceil(int v) { int p = 1; while (p < v) p <<= 1; return p; }

// ---------------------------------------------------------------- encoder: lifting + gather + segmented reduce
// out[b,q,:] = bl (*) S_0[q,:] + sum_ci Wl[:,ci] (*) S_ci[b,q,:],  S_ci = sum_{e in row q} a_e pn[b,j(e),ci] k_e,  S_0 = sum_e a_e k_e
// The lifting is applied at flush time (it is linear, so partial sums may be mixed before they are added).
template <int CI, int BCH, int UNR>
__global__ __launch_bounds__(256) void lift_ep_kernel(const float* __restrict__ k, const float* __restrict__ pn, const float* __restrict__ wl,
                                                      const float* __restrict__ bl, int B, int n_src, int C, const int* __restrict__ sp,
                                                      const int* __restrict__ cols, const int* __restrict__ eq, int Q, int E,
                                                      const float* __restrict__ escale, float* __restrict__ out, float* __restrict__ ws,
                                                      int lanes, int epc, int abl, const int* __restrict__ e_real) {
    if (e_real) E = min(E, *e_real);          // a padded union (plan.StaticUnion): the launch is sized for the capacity, the list ends earlier
    const int groups_per_block = 256 / lanes;
    const int g = blockIdx.x * groups_per_block + threadIdx.x / lanes;
    const int c = (threadIdx.x % lanes) * 4;
    const int b0 = blockIdx.y * BCH;
    const int begin = g * epc, end = min(begin + epc, E);
    if (begin >= E || c >= C) return;
    f32x4 wq[CI];
#pragma unroll
    for (int ci = 0; ci < CI; ++ci) wq[ci] = f32x4{wl[(c + 0) * CI + ci], wl[(c + 1) * CI + ci], wl[(c + 2) * CI + ci], wl[(c + 3) * CI + ci]};
    const f32x4 bq = bl ? *reinterpret_cast<const f32x4*>(bl + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 acc[BCH][CI], s0 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < BCH; ++b)
#pragma unroll
        for (int ci = 0; ci < CI; ++ci) acc[b][ci] = f32x4{0.f, 0.f, 0.f, 0.f};
    int cur = eq[begin];
    // the rows just outside the chunk tell whether its first / last row continues across the boundary: no row_splits reads
    // (dependent loads) on the flush path
    const int row_before = begin > 0 ? eq[begin - 1] : -1, row_after = end < E ? eq[end] : -1;
    bool first = true;
    auto flush = [&](int row, bool last) {
        const bool starts_here = !(first && row == row_before), ends_here = !(last && row == row_after);
        const bool complete = starts_here && ends_here;
        const int slot = starts_here ? 1 : 0;
        first = false;
#pragma unroll
        for (int b = 0; b < BCH; ++b) {
            if (b0 + b >= B) break;
            f32x4 o = bq * s0;
#pragma unroll
            for (int ci = 0; ci < CI; ++ci) { o += wq[ci] * acc[b][ci]; acc[b][ci] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            if ((abl & 2) && o[0] != 123.456f) continue;
            if (complete) *reinterpret_cast<f32x4*>(out + ((long)(b0 + b) * Q + row) * C + c) = o;
            else *reinterpret_cast<f32x4*>(ws + (((long)g * 2 + slot) * B + (b0 + b)) * C + c) = o;
        }
        s0 = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    for (int t = begin; t < end; t += UNR) {
        int j[UNR], row[UNR]; f32x4 kq[UNR]; float pv[UNR][BCH][CI];
#pragma unroll
        for (int u = 0; u < UNR; ++u) { const int tt = min(t + u, end - 1); j[u] = cols[tt]; row[u] = eq[tt]; }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int tt = min(t + u, end - 1);
            kq[u] = (abl & 4) ? f32x4{1.f * tt, 2.f, 3.f, 4.f} : *reinterpret_cast<const f32x4*>(k + (long)tt * C + c) * (escale ? escale[tt] : 1.0f);
#pragma unroll
            for (int b = 0; b < BCH; ++b) {
                const float* pr = pn + ((long)min(b0 + b, B - 1) * n_src + ((abl & 1) ? 0 : j[u])) * CI;
#pragma unroll
                for (int ci = 0; ci < CI; ++ci) pv[u][b][ci] = pr[ci];
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (t + u >= end) break;
            if (row[u] != cur) { flush(cur, false); cur = row[u]; }
            s0 += kq[u];
#pragma unroll
            for (int b = 0; b < BCH; ++b)
#pragma unroll
                for (int ci = 0; ci < CI; ++ci) acc[b][ci] += kq[u] * pv[u][b][ci];
        }
    }
    flush(cur, true);
}

// rows that span chunks: out[b,row,:] = carry(last slot of the chunk the row starts in) + sum of the first slots of the chunks it
// continues into; empty rows: 0.  One lane group per row, grid.y = batch.
__global__ __launch_bounds__(256) void ep_fixup_kernel(const int* __restrict__ sp, int R, int B, int C, const float* __restrict__ ws,
                                                       float* __restrict__ out, int lanes, int epc, float* __restrict__ out_amax) {
    const int rows_per_block = 256 / lanes;
    const int r = blockIdx.x * rows_per_block + threadIdx.x / lanes;
    const int c = (threadIdx.x % lanes) * 4;
    const int b = blockIdx.y;
    float am = 0.f;          // max |x| of what this thread stores (out_amax: every lane of a wave stays to the end)
    if (r < R && c < C) {
        const int t0 = sp[r], t1 = sp[r + 1];
        float* o = out + ((long)b * R + r) * C + c;
        if (t1 == t0) *reinterpret_cast<f32x4*>(o) = f32x4{0.f, 0.f, 0.f, 0.f};
        else {
            const int g0 = t0 / epc, g1 = (t1 - 1) / epc;
            if (g0 != g1) {                                  // (else: finished by the chunk that contains it)
                f32x4 acc = *reinterpret_cast<const f32x4*>(ws + (((long)g0 * 2 + 1) * B + b) * C + c);
                for (int g = g0 + 1; g <= g1; ++g) acc += *reinterpret_cast<const f32x4*>(ws + (((long)g * 2 + 0) * B + b) * C + c);
                *reinterpret_cast<f32x4*>(o) = acc;
                am = fmaxf(fmaxf(fabsf(acc[0]), fabsf(acc[1])), fmaxf(fabsf(acc[2]), fabsf(acc[3])));
            }
        }
    }
    if (out_amax != nullptr) amax_publish(out_amax, am, (int)threadIdx.x & 63, (int)(blockIdx.x + blockIdx.y * gridDim.x) * 4 + ((int)threadIdx.x >> 6));
}

// ---------------------------------------------------------------- decoder backward: dF over the TRANSPOSED CSR, edge-partitioned
// dF[b,j,:] = sum_{t in trow(j)} a_e k[e,:] (*) (sum_o dY[b,q(e),o] weff[o,:]),  e = t_edge[t];  the row of position t is index[e].
template <int OC, int BCH>
__global__ __launch_bounds__(256) void proj_t_ep_kernel(const float* __restrict__ k, const float* __restrict__ dy, const float* __restrict__ weff,
                                                        int B, int Q, int C, const int* __restrict__ tsp, const int* __restrict__ tedge,
                                                        const int* __restrict__ idx, const int* __restrict__ eq, int n_src, int E,
                                                        const float* __restrict__ escale, float* __restrict__ df, float* __restrict__ ws,
                                                        int lanes, int epc, const int* __restrict__ e_real, float* __restrict__ df_amax) {
    if (e_real) E = min(E, *e_real);
    const int groups_per_block = 256 / lanes;
    const int g = blockIdx.x * groups_per_block + threadIdx.x / lanes;
    const int c = (threadIdx.x % lanes) * 4;
    const int b0 = blockIdx.y * BCH;
    const int begin = g * epc, end = min(begin + epc, E);
    float am = 0.f;          // max |x| of the COMPLETE rows this thread stores (df_amax; the rows ep_fixup_kernel finishes are published there)
    if (begin < E && c < C) {
    f32x4 wq[OC];
#pragma unroll
    for (int o = 0; o < OC; ++o) wq[o] = *reinterpret_cast<const f32x4*>(weff + (long)o * C + c);
    f32x4 acc[BCH];
#pragma unroll
    for (int b = 0; b < BCH; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
    int cur = idx[tedge[begin]];
    const int row_before = begin > 0 ? idx[tedge[begin - 1]] : -1, row_after = end < E ? idx[tedge[end]] : -1;
    bool first = true;
    auto flush = [&](int row, bool last) {
        const bool starts_here = !(first && row == row_before), ends_here = !(last && row == row_after);
        const bool complete = starts_here && ends_here;
        const int slot = starts_here ? 1 : 0;
        first = false;
#pragma unroll
        for (int b = 0; b < BCH; ++b) {
            if (b0 + b >= B) break;
            if (complete) {
                *reinterpret_cast<f32x4*>(df + ((long)(b0 + b) * n_src + row) * C + c) = acc[b];
                am = fmaxf(am, fmaxf(fmaxf(fabsf(acc[b][0]), fabsf(acc[b][1])), fmaxf(fabsf(acc[b][2]), fabsf(acc[b][3]))));
            }
            else *reinterpret_cast<f32x4*>(ws + (((long)g * 2 + slot) * B + (b0 + b)) * C + c) = acc[b];
            acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    for (int t = begin; t < end; t += 4) {
        int e[4], q[4], row[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) e[u] = tedge[min(t + u, end - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u) { q[u] = eq[e[u]]; row[u] = idx[e[u]]; }
        f32x4 kv[4]; float gy[4][BCH][OC];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            kv[u] = *reinterpret_cast<const f32x4*>(k + (long)e[u] * C + c) * (escale ? escale[e[u]] : 1.0f);
#pragma unroll
            for (int b = 0; b < BCH; ++b) {
                const float* gr = dy + ((long)min(b0 + b, B - 1) * Q + q[u]) * OC;
#pragma unroll
                for (int o = 0; o < OC; ++o) gy[u][b][o] = gr[o];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (t + u >= end) break;
            if (row[u] != cur) { flush(cur, false); cur = row[u]; }
#pragma unroll
            for (int b = 0; b < BCH; ++b) {
                f32x4 gv = wq[0] * gy[u][b][0];
#pragma unroll
                for (int o = 1; o < OC; ++o) gv += wq[o] * gy[u][b][o];
                acc[b] += kv[u] * gv;
            }
        }
    }
    flush(cur, true);
    }
    if (df_amax != nullptr) amax_publish(df_amax, am, (int)threadIdx.x & 63, (int)(blockIdx.x + blockIdx.y * gridDim.x) * 4 + ((int)threadIdx.x >> 6));
}

// ---------------------------------------------------------------- decoder forward with the batch inside the lane group
// y[b,q,o] = sum_ch weff[o,ch] (sum_e a_e k[e,ch] f[b,j(e),ch]) + rowb[q,o] + bias[o]: the rows are short (<= a handful of latent
// tokens per mesh point) but the kernel-value rows k[e,:] are the big operand: with the batch as the grid's fast index every
// sample streamed all of k again (8 x 14 MB at the bench configuration, profiles/r2e_pmc_c2.json: 131 MB fetched for 24 MB of
// operands).  Here a lane group keeps BCH samples' sums and reads each k row once.
template <int OC, int BCH>
__global__ __launch_bounds__(256) void proj_fwd_bin_kernel(const float* __restrict__ k, const float* __restrict__ f, const float* __restrict__ weff,
                                                           const float* __restrict__ rowb, const float* __restrict__ bias, int B, int n_src,
                                                           int C, const int* __restrict__ sp, const int* __restrict__ cols, int Q,
                                                           const float* __restrict__ escale, float* __restrict__ y, int lanes,
                                                           const int* __restrict__ order) {
    const int rows_per_block = 256 / lanes;
    // `order` (optional): the rows sorted by their first source row, walked in contiguous ranges per XCD (workgroup b runs on XCD b % 8):
    // mesh points arrive in the caller's order, i.e. spatially random, and every row then pulled its gathered feature rows f[b,j,:]
    // through the fabric again (91 MB per launch for 8 MB of features, PMC); rows that share latent neighbours now share workgroups
    int r;
    {
        const int nb = gridDim.x, q8 = nb >> 3, r8 = nb & 7, x = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int lb = order ? (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + slot : (int)blockIdx.x;
        const int pos = lb * rows_per_block + threadIdx.x / lanes;
        r = (order && pos < Q) ? order[pos] : pos;
    }
    const int lr = threadIdx.x % lanes;
    const int c = lr * 4;
    const int b0 = blockIdx.y * BCH;
    const bool ok = r < Q && c < C;
    f32x4 acc[BCH];
#pragma unroll
    for (int b = 0; b < BCH; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (ok) {
        const int t0 = sp[r], t1 = sp[r + 1];
        for (int t = t0; t < t1; t += 2) {
            int j[2]; f32x4 kv[2]; f32x4 fv[2][BCH];
#pragma unroll
            for (int u = 0; u < 2; ++u) j[u] = cols[min(t + u, t1 - 1)];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int tt = min(t + u, t1 - 1);
                const float a = (t + u < t1) ? (escale ? escale[tt] : 1.0f) : 0.0f;
                kv[u] = *reinterpret_cast<const f32x4*>(k + (long)tt * C + c) * a;
#pragma unroll
                for (int b = 0; b < BCH; ++b) fv[u][b] = *reinterpret_cast<const f32x4*>(f + ((long)min(b0 + b, B - 1) * n_src + j[u]) * C + c);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int b = 0; b < BCH; ++b) acc[b] += kv[u] * fv[u][b];
        }
    }
#pragma unroll
    for (int b = 0; b < BCH; ++b)
#pragma unroll
        for (int o = 0; o < OC; ++o) {
            float d = 0.f;
            if (ok) { const f32x4 w = *reinterpret_cast<const f32x4*>(weff + (long)o * C + c); d = (acc[b][0] * w[0] + acc[b][1] * w[1]) + (acc[b][2] * w[2] + acc[b][3] * w[3]); }
            for (int off = lanes >> 1; off > 0; off >>= 1) d += __shfl_xor(d, off, 64);
            if (lr == 0 && r < Q && b0 + b < B)
                y[((long)(b0 + b) * Q + r) * OC + o] = d + (rowb ? rowb[(long)r * OC + o] : 0.f) + (bias ? bias[o] : 0.f);
        }
}

}  // namespace gaot

using namespace gaot;
#define ST(s) reinterpret_cast<hipStream_t>(s)

static int g_ep_chunk = 0;       // tuning hook: 0 = heuristic
extern "C" int gaot_debug_set_ep_chunk(int n) { const int old = g_ep_chunk; g_ep_chunk = n; return old; }
// measured (tools/gno_ep_bench.py, union of 16 skewed samples, 445 k edges): 16 / 32 / 64 / 128 edges per chunk -> 48 / 47 / 65 / 95 us
// chunk length: the kernels are latency-bound (a lane group lives for chunk / 4 dependent load rounds), so small edge lists want
// more, shorter chunks (C2, 55 k edges: dF 18.9 us at 16 vs 22.0 at 32; C3, 445 k edges: 35 us at 32 vs 43 at 16)
static inline int ep_chunk(int E) { return (g_ep_chunk & 0xffff) > 0 ? (g_ep_chunk & 0xffff) : (E < 131072 ? 16 : 32); }
static inline int ep_abl() { return g_ep_chunk >> 16; }      // tuning only: 1 = no pn gathers, 2 = no stores, 4 = no k_e loads

extern "C" int64_t gaot_gno_ep_workspace(int32_t E, int32_t C, int32_t B) {
    const int epc = ep_chunk(E);
    return (int64_t)cdiv(E > 0 ? E : 1, epc) * 2 * B * C;
}

extern "C" int gaot_gno_lift_gather_reduce_ep(const float* k, const float* pn, const float* wl, const float* bl, int32_t B, int32_t n_src,
                                              int32_t c_in, int32_t C, const int32_t* splits, const int32_t* cols, const int32_t* edge_query,
                                              int32_t Q, int32_t E, const float* escale, float* out, float* ws, const int32_t* e_real,
                                              gaot_stream_t stream) {
    GAOT_REQUIRE(B > 0 && n_src > 0 && Q >= 0 && E >= 0 && c_in >= 1 && c_in <= 4 && C > 0 && C % 4 == 0 && C <= 256,
                 "gno_lift_gather_reduce_ep: need 1 <= c_in <= 4, C %% 4 == 0, C <= 256 (got c_in %d, C %d)", c_in, C);
    if (Q == 0) return GAOT_OK;
    GAOT_REQUIRE(k && pn && wl && splits && out && ws && aligned16(k) && aligned16(out) && aligned16(ws) && (!bl || aligned16(bl)) &&
                 (E == 0 || (cols && edge_query)), "gno_lift_gather_reduce_ep: null or misaligned pointer");
    const int lanes = ep_pow2_ceil(C / 4), gpb = 256 / lanes, epc = ep_chunk(E);
    if (E > 0) {
        // one sample (the vx union of a batch): no batch blocking, eight edge rows in flight per lane group (the kernel is a stream
        // of k_e rows; with a chunk of 32 edges a lane group lives for only 32 / UNR dependent load rounds)
        const int BCH = B == 1 ? 1 : 2;
        dim3 grid(cdiv(cdiv(E, epc), gpb), cdiv(B, BCH)), block(256);
#define LG(CI) do { if (B == 1) hipLaunchKernelGGL((lift_ep_kernel<CI, 1, 8>), grid, block, 0, ST(stream), k, pn, wl, bl, B, n_src, C, splits, cols, edge_query, Q, E, \
                                  escale, out, ws, lanes, epc, ep_abl(), e_real); \
                    else hipLaunchKernelGGL((lift_ep_kernel<CI, 2, 4>), grid, block, 0, ST(stream), k, pn, wl, bl, B, n_src, C, splits, cols, edge_query, Q, E, \
                                  escale, out, ws, lanes, epc, ep_abl(), e_real); } while (0)
        if (c_in == 1) LG(1); else if (c_in == 2) LG(2); else if (c_in == 3) LG(3); else LG(4);
#undef LG
    }
    hipLaunchKernelGGL(ep_fixup_kernel, dim3(cdiv(Q, gpb), B), dim3(256), 0, ST(stream), splits, Q, B, C, ws, out, lanes, epc, (float*)nullptr);
    GAOT_CHECK_LAUNCH("gaot_gno_lift_gather_reduce_ep");
    return GAOT_OK;
}

extern "C" int gaot_gno_proj_gather_t_ep(const float* k, const float* dy, const float* weff, int32_t B, int32_t Q, int32_t n_src, int32_t C,
                                         int32_t out_channels, const int32_t* index32, const int32_t* edge_query, int32_t E,
                                         const int32_t* t_splits, const int32_t* t_edge, const float* escale, float* df, float* ws,
                                         const int32_t* e_real, gaot_stream_t stream) {
    return gaot_gno_proj_gather_t_ep_w(k, dy, weff, B, Q, n_src, C, out_channels, index32, edge_query, E, t_splits, t_edge, escale, df, ws, e_real, nullptr, stream);
}

extern "C" int gaot_gno_proj_gather_t_ep_w(const float* k, const float* dy, const float* weff, int32_t B, int32_t Q, int32_t n_src, int32_t C,
                                           int32_t out_channels, const int32_t* index32, const int32_t* edge_query, int32_t E,
                                           const int32_t* t_splits, const int32_t* t_edge, const float* escale, float* df, float* ws,
                                           const int32_t* e_real, float* df_absmax, gaot_stream_t stream) {
    GAOT_REQUIRE(B > 0 && out_channels >= 1 && out_channels <= 4 && C > 0 && C % 4 == 0 && C <= 256 && n_src > 0 && E >= 0,
                 "gno_proj_gather_t_ep: need 1 <= out_channels <= 4, C %% 4 == 0, C <= 256");
    GAOT_REQUIRE(k && dy && weff && t_splits && df && ws && aligned16(k) && aligned16(weff) && aligned16(df) && aligned16(ws) &&
                 (E == 0 || (index32 && edge_query && t_edge)), "gno_proj_gather_t_ep: null or misaligned pointer");
    const int lanes = ep_pow2_ceil(C / 4), gpb = 256 / lanes, epc = ep_chunk(E);
    constexpr int BCH = 4;
    if (E > 0) {
        dim3 grid(cdiv(cdiv(E, epc), gpb), cdiv(B, BCH)), block(256);
#define PT(OC) hipLaunchKernelGGL((proj_t_ep_kernel<OC, BCH>), grid, block, 0, ST(stream), k, dy, weff, B, Q, C, t_splits, t_edge, index32, edge_query, \
                                  n_src, E, escale, df, ws, lanes, epc, e_real, df_absmax)
        if (out_channels == 1) PT(1); else if (out_channels == 2) PT(2); else if (out_channels == 3) PT(3); else PT(4);
#undef PT
    }
    hipLaunchKernelGGL(ep_fixup_kernel, dim3(cdiv(n_src, gpb), B), dim3(256), 0, ST(stream), t_splits, n_src, B, C, ws, df, lanes, epc, df_absmax);
    GAOT_CHECK_LAUNCH("gaot_gno_proj_gather_t_ep");
    return GAOT_OK;
}

extern "C" int gaot_gno_proj_gather_reduce_bin(const float* k, const float* f, const float* weff, const float* rowbias, const float* bias,
                                               int32_t B, int32_t n_src, int32_t C, int32_t out_channels, const int32_t* splits,
                                               const int32_t* cols, int32_t Q, const float* escale, float* y, const int32_t* row_order,
                                               gaot_stream_t stream) {
    GAOT_REQUIRE(B > 0 && out_channels >= 1 && out_channels <= 4 && C > 0 && C % 4 == 0 && C <= 256,
                 "gno_proj_gather_reduce_bin: need 1 <= out_channels <= 4, C %% 4 == 0, C <= 256");
    if (Q == 0) return GAOT_OK;
    GAOT_REQUIRE(k && f && weff && splits && cols && y && aligned16(k) && aligned16(f) && aligned16(weff), "gno_proj_gather_reduce_bin: null or misaligned pointer");
    const int lanes = ep_pow2_ceil(C / 4), rpb = 256 / lanes;
    constexpr int BCH = 4;
    dim3 grid(cdiv(Q, rpb), cdiv(B, BCH)), block(256);
#define PG(OC) hipLaunchKernelGGL((proj_fwd_bin_kernel<OC, BCH>), grid, block, 0, ST(stream), k, f, weff, rowbias, bias, B, n_src, C, splits, cols, Q, \
                                  escale, y, lanes, row_order)
    if (out_channels == 1) PG(1); else if (out_channels == 2) PG(2); else if (out_channels == 3) PG(3); else PG(4);
#undef PG
    GAOT_CHECK_LAUNCH("gaot_gno_proj_gather_reduce_bin");
    return GAOT_OK;
}
