/*
 * gaot_hip.h -- C ABI of libgaot_hip.so: the MI355X (gfx950 / CDNA4) kernels behind the GAOT
 * forward/backward hot path (camlab-ethz/GAOT src/model; citations are file:line in that repo).
 *
 * Conventions
 *   - every entry point returns 0 on success, a negative gaot_status on failure;
 *     gaot_last_error() returns a thread-local message for the last failure.
 *   - all pointers are DEVICE pointers owned by the caller; the library never allocates, frees or
 *     synchronises (graph-capture safe).  `stream` is a hipStream_t passed as void*.
 *   - dense tensors are row-major fp32; index arrays are int32 unless the name says i64.
 *   - "CSR" = the reference's neighbour dict (neighbor_search.py:139-140): index[E] holds, for each
 *     query segment, the indices into the SOURCE point set; splits[Q+1] the segment bounds.
 */
#ifndef GAOT_HIP_H
#define GAOT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* gaot_stream_t;

enum gaot_status {
    GAOT_OK = 0,
    GAOT_ERR_BAD_ARG = -1,     /* shape / pointer / alignment contract violated */
    GAOT_ERR_LAUNCH = -2,      /* hipLaunch / hipGetLastError failed            */
    GAOT_ERR_UNSUPPORTED = -3  /* valid request this build does not implement   */
};

enum gaot_act {
    GAOT_ACT_NONE = 0,
    GAOT_ACT_GELU = 1,      /* exact erf GELU, mlp.py:307-337 (F.gelu default)            */
    GAOT_ACT_RELU = 2,      /* gemb.py:54-59                                              */
    GAOT_ACT_GELU_BWD = 3,  /* out = acc * gelu'(aux)   (aux = saved pre-activation)      */
    GAOT_ACT_RELU_BWD = 4,  /* out = acc * (aux > 0)    (aux = saved post-activation)     */
    /* SwiGLU gate of the transformer FFN (attn.py:150-156) fused into the two GEMMs either side of it:            */
    GAOT_ACT_SWIGLU_BWD = 5,/* acc = dg [M,N]; aux_in = u = [u1|u3] [M,2N]; C [M,2N] = [dg*u3*silu'(u1) | dg*silu(u1)] */
    GAOT_ACT_SWIGLU = 6     /* B = [w1;w3] (2F rows, N = 2F, k-major); C [M,F] = silu(u1)*u3; aux_out [M,2F] = u (optional).
                               Needs K % 32 == 0, F % 4 == 0, no bias/rowbias/rowscale/residual/split_k.             */
};

int gaot_abi_version(void);
const char* gaot_last_error(void);

/* A "magnitude word" (gaot_gemm_desc.a_absmax ..., gaot_absmax_grouped) is GAOT_AMAX_SLOTS slots, one float at the head of each
 * 128-byte line (GAOT_AMAX_SLOTS * GAOT_AMAX_STRIDE floats in all, 128-byte aligned): producers publish max |x| into the slot their wave
 * index selects (atomic max on the float bit pattern), consumers take the maximum of the slots. */
#define GAOT_AMAX_SLOTS 32
#define GAOT_AMAX_STRIDE 32

/* ------------------------------------------------------------------------------------------
 * fp32 GEMM on the matrix cores (v_mfma_f32_32x32x2_f32), fused prologue/epilogue.
 * Replaces every nn.Linear / Conv1d(k=1) on the path and their backward products:
 *   ChannelMLP mlp.py:283-305, LinearChannelMLP mlp.py:329-337, q/k/v/o_proj attn.py:92-117,
 *   FFN attn.py:150-156, skip_proj attn.py:225-227, patch_linear gaot.py:208.
 *
 *   acc[m,n] = sum_k Aop[m,k] * Bop[k,n]
 *     Aop[m,k] = a_kmajor ? A[m*lda + k] : A[k*lda + m];  for k >= k_split (if A2): A2 with k-k_split
 *     Bop[k,n] = b_kmajor ? B[n*ldb + k] : B[k*ldb + n]   (b_kmajor=1 is an nn.Linear weight [N,K])
 *   v = acc (+ bias[n]) (+ rowbias[(m % rowbias_period)*ld_rowbias + n]);  v *= rowscale[m]
 *   if aux_out: aux_out[m*ld_aux + n] = v;      v = act(v)  (or v *= act'(aux_in[m*ld_aux+n]))
 *   v += residual[m*ldr + n];   C[m*ldc + n] = v
 * Optional pointers may be NULL.  split_k > 1 needs workspace >= split_k*(M*N + M) floats.
 * ------------------------------------------------------------------------------------------ */
typedef struct gaot_gemm_desc {
    int32_t M, N, K;
    const float* A;  int64_t lda;  int32_t a_kmajor;
    const float* A2; int64_t lda2; int32_t k_split;
    const float* B;  int64_t ldb;  int32_t b_kmajor;
    float* C;        int64_t ldc;
    const float* bias;
    const float* rowbias; int32_t rowbias_period; int64_t ld_rowbias;
    const float* rowscale;
    int32_t act;
    const float* aux_in; float* aux_out; int64_t ld_aux;
    const float* residual; int64_t ldr;
    int32_t split_k; float* workspace;
    float* colsum;   /* optional, a_kmajor = 0 only: colsum[m] = sum_k Aop[m,k] (bias gradient fused into dW = dY^T X) */
    /* precision of the product where the split-bf16 tile kernels run it (per call, no global state): 0 or 3 = every fp32 operand as
     * THREE bf16 pieces, six piece products: exact to fp32 rounding (error vs float64 ~ 2e-7, like the fp32 MFMA); 2 = TWO pieces, both
     * rounded to nearest (x = h + m + e, |e| <= 2^-18 |x|, unbiased), three piece products: 16 significant bits per operand, half the
     * matrix-pipe work.  4 = TWO fp16 pieces of the SCALED operand, both rounded to nearest, three piece products on
     * v_mfma_f32_32x32x16_f16: with s = the power of two that puts the operand's largest magnitude into [2^13, 2^14), s x = h + m + e,
     * |e| <= 2^-23 |s x| (at most the operand's last bit; zero for three values in four) for every element within 2^-16 of the largest, an absolute
     * 2^-39 of the largest below that: fp32-level products (measured error vs float64 at or below the three-piece products') at the
     * matrix-pipe work of the two-piece ones.  Needs a_absmax / b_absmax; without them, and on kernels off the split tiles, it means 3.
     * Kernels on the fp32 MFMA / vector pipe ignore the field.  Anything else: GAOT_ERR_INVALID. */
    int32_t pieces;
    /* pieces = 4: device magnitude words (above) holding max |Aop| and max |Bop| (an upper bound is as good: one
     * binade of slack costs nothing), read by the kernel -- no host value, so a captured launch follows the data.  gaot_absmax_grouped
     * computes them; producers publish them. */
    const float* a_absmax; const float* b_absmax;
    /* optional (pieces = 4): Bop ALREADY split into the two fp16 pieces the kernel would form from it (weights: once per pass by
     * gaot_split_f16_planes_grouped from the SAME b_absmax word, instead of once per workgroup per k-tile).  Piece q of Bop[k,n] is the
     * 16-bit word b_planes[n * ld_bplanes + (k / 16) * 32 + q * 16 + k % 16] (k-contiguous in groups of 16 whatever b_kmajor says: the
     * two pieces of a group are one 64-byte segment).  b_plane_stride must be 16.  Bit-identical products; kernels that do not take
     * the pieces ignore the field.  ld_bplanes a multiple of 8, 16-byte aligned. */
    const void* b_planes; int64_t ld_bplanes; int64_t b_plane_stride;
    /* optional (any pieces): the magnitude word of C as this call stores it (atomic max per slot): the word must be ZERO (or hold a
     * running maximum of the same tensor) before the call.  The next product's a_absmax.  Published from the tile kernels' vector
     * epilogue, from the split-K reduce launch, or -- skinny / scalar-epilogue kernels -- by one gaot_absmax_grouped launch over C. */
    float* c_absmax;
    /* pieces = 4 with A2: the magnitude word of A2 (the kernel scales both halves of the concatenated operand by the larger word) */
    const float* a2_absmax;
    /* 1 (with split_k > 1 and no epilogue operand: bias / row bias / row scale / activation / aux / residual / colsum): leave the
     * product as its K slabs -- workspace[z * M * N + m * N + n], z < gaot_gemm_slab_count(K, split_k) -- and launch no reduce; the
     * consumer sums them (gaot_rmsnorm_bwd_slabs).  C is not written and may be null; c_absmax is ignored. */
    int32_t raw_slabs;
} gaot_gemm_desc;

/* the number of K slabs gaot_gemm_f32 cuts a reduction of K into when asked for split_k (32-wide k-tiles, no empty slab) */
int32_t gaot_gemm_slab_count(int32_t K, int32_t split_k);

int gaot_gemm_f32(const gaot_gemm_desc* d, gaot_stream_t stream);
/* which kernel family gaot_gemm_f32 WOULD run this product on: 1 = fp32-MFMA tiles, 2 = skinny vector kernels, 3 = split tiles on the
 * bf16 / fp16 matrix pipe (the only ones that read pieces / *_absmax); launches nothing; < 0 on a bad descriptor */
int gaot_gemm_path(const gaot_gemm_desc* d);
/* fp16 pieces of weight matrices for gaot_gemm_desc.b_planes, n matrices per launch: item i reads src[r * ld + c] (rows x cols), scales
 * by the power of two its magnitude word `absmax` selects, and writes piece q (0: h = rn16(s x), 1: m = rn16(s x - h)) of element (r, c)
 * to planes_k[r * 2 cols + (c / 16) * 32 + q * 16 + c % 16] (as stored: the k-contiguous B operand of x W^T, ld_bplanes = 2 cols) and to
 * planes_t[c * 2 rows + (r / 16) * 32 + q * 16 + r % 16] (transposed: the k-contiguous B operand of the input-gradient product dY W,
 * ld_bplanes = 2 rows); either may be NULL.  2 rows cols 16-bit words each; rows, cols multiples of 16.
 * The launch also writes the planes' RANGE VERDICT into the two floats that follow the head of the word's FIRST slot: float [1] (rows of the
 * matrix as stored) and float [2] (its columns) = the eighth-largest, over the 64-wide groups of a 64 x 64 tile, of log2(tensor maximum /
 * group maximum), maximised over tiles (an atomic max on the float's bits).  The fp16-piece tile kernels read both beside b_planes: a product
 * whose operands' spread exceeds what one power of two per tensor carries is computed again on the fp32 MFMA (csrc/gemm_split.hip).  They
 * must be ZERO before the launch -- zero the whole first slot line together with the slot heads; gaot_absmax_grouped never touches them -- and
 * a word that is reused for another tensor must be zeroed again: stale verdicts cause needless second passes, garbage there is read as a
 * (clamped) number of binades. */
typedef struct gaot_f16_planes_item {
    const float* src; int64_t ld; int32_t rows, cols; const float* absmax; void* planes_k; void* planes_t;
} gaot_f16_planes_item;
int gaot_split_f16_planes_grouped(const gaot_f16_planes_item* items, int32_t n, gaot_stream_t stream);
/* magnitude words for pieces = 4: max over the slots of out_i = max(that, max |x_i[r * ld + c]|) over rows x cols, n matrices in ONE
 * launch (atomic max per slot: zero the words first, or let them accumulate over pieces of one tensor).  NaNs are ignored. */
typedef struct gaot_absmax_item { const float* x; int64_t ld; int32_t rows, cols; float* out; } gaot_absmax_item;
int gaot_absmax_grouped(const gaot_absmax_item* items, int32_t n, gaot_stream_t stream);

/* Grouped weight-gradient products: ONE launch over n products  out_i[M_i,N_i] = g_i[K_i,M_i]^T x_i[K_i,N_i]  (+ colsum_i[m] =
 * sum_k g_i[k,m], the bias gradient), i.e. dW = dY^T X (and db) of every nn.Linear / Conv1d(k=1) whose backward has been
 * reached (mlp.py:283-305, attn.py:92-117,150-156,225-227, gaot.py:208: autograd runs them one by one, each a long reduction
 * over all tokens into a small matrix).  Products on the bf16 / fp16 matrix pipe with `pieces` (0 / 3, 4 or 2) as gaot_gemm_desc.pieces (4 falls back to 3 unless every item carries its magnitude words); K slabs (4 096 rows; the matrix pipe accumulates runs of 1 024, the running sums live in registers) are summed in slab order by
 * the last workgroup to finish a tile (deterministic, no atomics on data).
 * Needs M, N % 4 == 0, K % 32 == 0, ld* % 4 == 0, 16-byte aligned pointers.  `workspace`: >= gaot_gemm_tn_grouped_workspace()
 * floats; `counters`: >= *n_counters int32, ZERO before the first call (every call leaves them zero again). */
typedef struct gaot_wgrad_item {
    const float* g;  int64_t ldg;     /* [K, M] row-major: the output gradient dY (tokens x out_features)  */
    const float* x;  int64_t ldx;     /* [K, N] row-major: the layer input X  (tokens x in_features)       */
    float* out;      int64_t ldo;     /* [M, N]: dW                                                        */
    float* colsum;                    /* optional [M]: db                                                  */
    int32_t M, N, K;
    const float* g_absmax; const float* x_absmax;   /* pieces = 4: magnitude words of g and x (as gaot_gemm_desc.a_absmax); else may be NULL */
} gaot_wgrad_item;
int64_t gaot_gemm_tn_grouped_workspace(const gaot_wgrad_item* items, int32_t n, int32_t* n_counters);
int gaot_gemm_tn_grouped(const gaot_wgrad_item* items, int32_t n, int32_t pieces, float* workspace, int32_t* counters, gaot_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Geometry plan pieces (once per mesh geometry; torch_scatter / repeat_interleave call sites
 * agno.py:131-141,206-207, gemb.py:110-143 become precomputed index arrays).
 * ------------------------------------------------------------------------------------------ */
/* int64 CSR -> int32 CSR + per-edge query id.  index32[E], splits32[Q+1], edge_query[E]. */
int gaot_csr_prepare(const int64_t* index_i64, const int64_t* splits_i64, int32_t Q, int32_t E, int32_t n_src,
                     int32_t* index32, int32_t* splits32, int32_t* edge_query, int32_t* status_flag,
                     gaot_stream_t stream);
/* transposed (by-source) CSR: t_splits[n_src+1], t_edge[E] (edge ids ascending inside a source: counting sort by source, then a
 * per-row sort -- thread-level for short rows, a workgroup-level rank sort for the long rows of skewed meshes).
 * scratch: n_src + 1 + E int32. */
/* out = [parts[0] + offsets[0] | parts[1] + offsets[1] | ...]: block-diagonal union of per-sample CSR plans (vx mode) without
 * re-deriving them.  parts / lens / offsets are HOST arrays of n_parts entries (device pointers inside parts). */
int gaot_concat_offset(const int32_t* const* parts, const int32_t* lens, const int32_t* offsets, int32_t n_parts, int32_t* out,
                       gaot_stream_t stream);
int gaot_csr_transpose(const int32_t* index32, int32_t E, int32_t n_src,
                       int32_t* t_splits, int32_t* t_edge, int32_t* scratch, gaot_stream_t stream);
/* The same union from a DEVICE-side table, into STATIC buffers sized for a capacity e_cap >= the union's edge count: what a captured
 * (hipGraph) vx training step replays while the batch composition changes every step -- the reference's shuffling loader,
 * data_utils.py:272-294 with static_trainer.py:180-202; magno.py:356-413 / 694-751 loop over the samples in Python.  The caller uploads one
 * gaot_union_part per sample (device pointers of that sample's plan arrays and coordinates, its first edge's position in the union and its
 * edge count) before the launch; every sample has q_each query rows and n_src_each sources.  Writes index32 / edge_query / t_edge [e_cap],
 * splits32 [n_parts * q_each + 1], t_splits [n_parts * n_src_each + 1], the stacked coordinates src [n_parts * n_src_each, dim_src] and
 * dst [n_parts * q_each, dim_dst] (either may be NULL: not copied) and *e_real = the union's edge count.  The edges e_real..e_cap-1 ("pads")
 * belong to no row of either CSR; they carry source 0, query 0 and t_edge = their own id, so kernels that walk the flat edge list over
 * e_cap entries stay in bounds, and they must be given an edge scale of 0 (gaot_edge_zero_pads, gaot_edge_inv_degree). */
typedef struct gaot_union_part {
    const int32_t* index;        /* [e_count]  source ids of the sample's CSR                     */
    const int32_t* edge_query;   /* [e_count]  query id per edge                                   */
    const int32_t* t_edge;       /* [e_count]  edge ids of the transposed CSR                      */
    const int32_t* splits;       /* [q_each + 1]                                                   */
    const int32_t* t_splits;     /* [n_src_each + 1]                                               */
    const float* src;            /* [n_src_each, dim_src] source coordinates (NULL with src = NULL) */
    const float* dst;            /* [q_each, dim_dst]     query coordinates                        */
    int32_t e_begin;             /* position of the sample's first edge in the union               */
    int32_t e_count;
} gaot_union_part;
int gaot_union_compose(const gaot_union_part* parts_dev, int32_t n_parts, int32_t q_each, int32_t n_src_each, int32_t dim_src,
                       int32_t dim_dst, int32_t e_cap, int32_t* index32, int32_t* edge_query, int32_t* t_edge, int32_t* splits32,
                       int32_t* t_splits, float* src, float* dst, int32_t* e_real, gaot_stream_t stream);
/* Training-time neighbour sub-sampling on the device (edge_drop.py:54-99: 'ratio' = mode 1 keeps every edge with probability sample_ratio,
 * 'max_neighbors' = mode 2 keeps a uniformly random subset of max_neighbors edges of every row that has more; the reference draws with torch's
 * generator and rebuilds the CSR with boolean indexing / bincount / cumsum per step).  Here: a compaction of a plan's arrays (int32 CSR +
 * transposed CSR as gaot_csr_prepare / gaot_csr_transpose / gaot_union_compose leave them; e_real_in = that plan's device edge count or NULL
 * for E) into out_* buffers of the SAME capacity E, *e_real_out = the kept count, pads as gaot_union_compose writes them.  Nothing about the
 * draw is a launch argument: `seed` is a device word (gaot_attention_seed_next advances it), so a captured step draws anew on every replay.
 * Both CSRs stay sorted (order-preserving compaction).  scratch: gaot_edge_drop_scratch(E) int32. */
int64_t gaot_edge_drop_scratch(int32_t E);
int gaot_edge_drop(const int32_t* index32, const int32_t* edge_query, const int32_t* t_edge, const int32_t* splits32,
                   const int32_t* t_splits, int32_t Q, int32_t n_src, int32_t E, const int32_t* e_real_in, int32_t mode,
                   float sample_ratio, int32_t max_neighbors, const uint64_t* seed, int32_t* out_index, int32_t* out_edge_query,
                   int32_t* out_t_edge, int32_t* out_splits, int32_t* out_t_splits, int32_t* e_real_out, int32_t* scratch,
                   gaot_stream_t stream);
/* The same union straight from the callers' int64 CSR lists -- the per-sample dicts the reference's trainer uploads anew every step
 * (move_to_device, static_trainer.py:192-193) -- without building a plan per sample first: index32 / edge_query [e_cap] (edge_query by a
 * binary search in the sample's row splits), splits32, stacked coordinates, *e_real, pads as above.  *status_flag |= 1 (row splits not
 * monotone 0..e_count) | 2 (index outside [0, n_src_each)): the contract of gaot_csr_prepare, checked on the device (what is stored is clamped);
 * zero it once, read it whenever convenient.  The transposed CSR of the result: gaot_csr_transpose_dev. */
typedef struct gaot_union_part_raw {
    const int64_t* index;        /* [e_count]     neighbors_index of the sample        */
    const int64_t* splits;       /* [q_each + 1]  neighbors_row_splits of the sample   */
    const float* src;            /* [n_src_each, dim_src]                              */
    const float* dst;            /* [q_each, dim_dst]                                  */
    int32_t e_begin;
    int32_t e_count;
    int64_t reserved[3];         /* (64 bytes per entry, like gaot_union_part: one table serves both)  */
} gaot_union_part_raw;
int gaot_union_compose_raw(const gaot_union_part_raw* parts_dev, int32_t n_parts, int32_t q_each, int32_t n_src_each, int32_t dim_src,
                           int32_t dim_dst, int32_t e_cap, int32_t* index32, int32_t* edge_query, int32_t* splits32, float* src, float* dst,
                           int32_t* e_real, int32_t* status_flag, gaot_stream_t stream);
/* gaot_csr_transpose for a padded list: only the first *e_real of the E entries are edges (t_splits[n_src] = *e_real), the pads get
 * t_edge = their own id.  scratch: gaot_csr_transpose_dev_scratch(E, n_src) int32. */
int64_t gaot_csr_transpose_dev_scratch(int32_t E, int32_t n_src);
int gaot_csr_transpose_dev(const int32_t* index32, int32_t E, const int32_t* e_real, int32_t n_src, int32_t* t_splits, int32_t* t_edge,
                           int32_t* scratch, gaot_stream_t stream);
/* out[e] = 1 / max(deg(query(e)), 1) -- the 'mean' reduction of agno.py:264 as a per-edge scale -- and 0 for e >= *e_real (e_real may be NULL) */
int gaot_edge_inv_degree(const int32_t* splits32, const int32_t* edge_query, int32_t E, const int32_t* e_real, float* out,
                         gaot_stream_t stream);
/* a[b, e, :] = 0 for *e_real <= e < E, a [B, E, width]: the pads of a padded union -- under whatever per-edge scale the transform uses, and in
 * per-edge gradient rows a flat kernel filled before a reduction over all E rows reads them.  max_pads: an upper bound of E - *e_real known on
 * the host (sizes the launch; E if unknown). */
int gaot_edge_zero_pads(float* a, int32_t B, int32_t E, int32_t width, const int32_t* e_real, int32_t max_pads, gaot_stream_t stream);
/* Content guard of the per-geometry caches.  The reference trainer uploads the (unchanged) coordinates anew every step
 * (static_trainer.py:167-170), i.e. NEW tensors with the OLD bytes.  The plan keeps a copy of the bytes its cached arrays were
 * computed from:
 *   gaot_guard_begin(flag)                      *flag = 0
 *   gaot_guard_compare(current, kept, n, flag)  *flag |= (current != kept)        (any number of operands)
 *   gaot_guard_update(current, kept, n, flag)   if (*flag) kept = current
 * and the plan kernels below take `guard` = that flag: with a non-null guard whose value is 0 they return immediately and the
 * cached output stays; guard = NULL always computes.  Nothing is read back by the host (no synchronisation). */
int gaot_guard_begin(int32_t* flag, gaot_stream_t stream);
int gaot_guard_compare(const void* current, const void* kept, int64_t nbytes, int32_t* flag, gaot_stream_t stream);
int gaot_guard_update(const void* current, void* kept, int64_t nbytes, const int32_t* flag, gaot_stream_t stream);
/* two (current, kept) pairs in ONE launch: flag |= (any word differs), then kept := current (what the captured-step path of
 * autograph.py does with the latent and physical coordinates before every replay) */
int gaot_guard_sync2(const void* cur0, void* kept0, int64_t nbytes0, const void* cur1, void* kept1, int64_t nbytes1,
                     int32_t* flag, gaot_stream_t stream);
/* cosine edge attention + segment softmax (agno.py:112-146, 211-224): attn[E].
 * src [n_src,dim], qry [Q,dim] are the (possibly node_pos_encoded) kernel coordinates. */
int gaot_edge_attention_cosine(const float* src, const float* qry, int32_t dim,
                               const int32_t* index32, const int32_t* splits32, int32_t Q,
                               float* attn, const int32_t* guard, gaot_stream_t stream);
/* segment softmax of given scores (dot-product attention, agno.py:215-217,224): fwd and bwd.
 * bwd: dscore = attn * (dattn - sum_seg(attn*dattn)). */
int gaot_segment_softmax_fwd(const float* score, const int32_t* splits32, int32_t Q, float* attn, gaot_stream_t stream);
int gaot_segment_softmax_bwd(const float* attn, const float* dattn, const int32_t* splits32, int32_t Q,
                             float* dscore, gaot_stream_t stream);
/* kernel-MLP input rows [y_j , x_i] (agno.py:188,206-207,229): feat[E, 2*dim]. */
int gaot_edge_features(const float* src, const float* qry, int32_t dim,
                       const int32_t* index32, const int32_t* edge_query, int32_t E,
                       float* feat, const int32_t* guard, gaot_stream_t stream);
/* GeometricEmbedding statistics, standardised (gemb.py:83-171): stats[Q, 3+2*dim], dim in {2,3}.  The Q rows form `groups`
 * equal groups (vx mode: one per sample of a block-diagonal union), each standardised on its own as the reference does per
 * sample.  scratch: 34*(3+2*dim)*groups doubles (per group: 2 x 16 per-workgroup partial column sums, added in a fixed order, + mean and
 * divisor). */
int gaot_geo_stats(const float* geom, const float* qry, int32_t dim,
                   const int32_t* index32, const int32_t* splits32, int32_t Q,
                   float* stats, double* scratch, const int32_t* guard, int32_t groups, gaot_stream_t stream);

/* radius graph by cell list (replaces NeighborSearch backends, neighbor_search.py:65-335; `dist <= r` inclusive,
 * unbounded degree, ascending data index per query like `_native_neighbor_search`).  origin[dim] / dims[dim] are HOST
 * arrays describing a uniform grid with cell size `cell` >= radius that covers the data points.
 *   1. gaot_cells_build : cell_start[ncell+1], cell_points[n]           (scratch: 2 * n + ncell + 1 int32)
 *   2. gaot_radius_count: deg[m], splits[m+1] (int64; caller reads splits[m] = E to size the index array)
 *   3. gaot_radius_fill : index[E] (int64)      (scratch: E int64, used by rows of more than 4096 neighbours; n_data = number of
 *      data points: with more data points than queries a wave works per query -- dense cells of a skewed cloud -- else a thread)
 * max_neighbors > 0 with strict = 1 reproduces the `torch_cluster` backend (neighbor_search.py:148-175: torch_cluster.radius
 * with its default max_num_neighbors = 32): squared distance strictly below r^2, and of a query's neighbours only the
 * max_neighbors with the SMALLEST data indices (its kernel scans the data in index order and stops at the cap).
 * max_neighbors = 0, strict = 0: the in-repo backends (native / chunked / grid). */
int gaot_cells_build(const float* data, int32_t n, int32_t dim, const float* origin, float cell, const int32_t* dims,
                     int32_t* cell_start, int32_t* cell_points, int32_t* scratch, gaot_stream_t stream);
int gaot_radius_count(const float* queries, int32_t m, const float* data, int32_t n_data, int32_t dim, float radius,
                      const float* origin, float cell, const int32_t* dims, const int32_t* cell_start,
                      const int32_t* cell_points, int32_t* deg, int64_t* splits, int32_t max_neighbors, int32_t strict,
                      gaot_stream_t stream);
int gaot_radius_fill(const float* queries, int32_t m, const float* data, int32_t n_data, int32_t dim, float radius,
                     const float* origin, float cell, const int32_t* dims, const int32_t* cell_start,
                     const int32_t* cell_points, const int64_t* splits, int64_t* index, int64_t* scratch, int32_t max_neighbors,
                     int32_t strict, gaot_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * GNO integral transform = gather / per-edge weight / CSR segment reduce (agno.py:198,245-271).
 *   out[b,r,:] = sum_{t in [splits[r],splits[r+1])} escale[edge(t)] * w[edge(t),:] * src[b, col(t), :]
 *     edge(t) = edge_map ? edge_map[t] : t ;  col(t) = cols[edge(t)] when edge_map else cols[t]
 * forward  : splits/cols = CSR, edge_map = NULL, src = f_y                 -> out [B,Q,C]
 * backward : splits = t_splits, edge_map = t_edge, cols = edge_query, src = dOut -> dF [B,n_src,C]
 * w may be NULL (all ones).  escale[E] may be NULL; it carries the edge attention weight (agno.py:249-250)
 * or 1/deg(query) for the 'mean' reduction (agno.py:264).  Any C (16-byte lanes when C % 4 == 0).
 * ------------------------------------------------------------------------------------------ */
int gaot_gno_gather_reduce(const float* w, const float* src, int32_t B, int32_t n_src_rows, int32_t C,
                           const int32_t* splits, const int32_t* cols, const int32_t* edge_map,
                           int32_t n_out_rows, const float* escale, float* out, gaot_stream_t stream);
/* Encoder variant with the point-wise LINEAR lifting (magno.py:334, one Conv1d(k=1): f = Wl pn + bl, c_in <= 4) folded in:
 *   out[b,q,:] = sum_ci Wl[:,ci] * (sum_e a_e pn[b,j(e),ci] k_e) + bl * (sum_e a_e k_e)
 * -- identical to gaot_gno_gather_reduce(k, f) on the lifted f, which is never materialised.  pn [B,n_src,c_in], wl [C,c_in].
 * Backward: dk [E,C] plus gaot_gno_lift_edge_grad_parts(E,C) partial rows [(c_in+1)*C] = [dWl^T (c_in x C) | dbl (C)]
 * that the caller sums (gaot_colsum); pn gets no gradient (raw input data).  out_absmax (optional): out's magnitude word
 * (gaot_gemm_desc.c_absmax conventions: zero before the launch). */
int gaot_gno_lift_gather_reduce(const float* k, const float* pn, const float* wl, const float* bl, int32_t B, int32_t n_src,
                                int32_t c_in, int32_t C, const int32_t* splits, const int32_t* cols, int32_t Q,
                                const float* escale, float* out, float* out_absmax, gaot_stream_t stream);
int32_t gaot_gno_lift_edge_grad_parts(int32_t E, int32_t C);
int gaot_gno_lift_edge_grad(const float* dout, const float* k, const float* pn, const float* wl, const float* bl, int32_t B,
                            int32_t Q, int32_t n_src, int32_t c_in, int32_t C, const int32_t* index32,
                            const int32_t* edge_query, int32_t E, const float* escale, float* dk, float* partial,
                            gaot_stream_t stream);
/* Decoder variant with the point-wise LINEAR maps that follow the transform (recovery + projection, magno.py:640-668, folded
 * by the caller into weff [OC,C] and a per-query row bias [Q,OC]; OC = out_channels <= 4) applied inside the kernel:
 *   y[b,q,o] = sum_ch weff[o,ch] * (sum_e a_e k[e,ch] f[b,j(e),ch]) + rowbias[q,o] + bias[o]        (rowbias / bias may be NULL)
 * The [B,Q,C] transform output is never written.  Backward from dy [B,Q,OC]: dk [E,C], dF [B,n_src,C] (df may be NULL),
 * and gaot_gno_lift_edge_grad_parts(E,C) partial rows [OC*C] of dweff (caller sums them with gaot_colsum). */
int gaot_gno_proj_gather_reduce(const float* k, const float* f, const float* weff, const float* rowbias, const float* bias,
                                int32_t B, int32_t n_src, int32_t C, int32_t out_channels, const int32_t* splits,
                                const int32_t* cols, int32_t Q, const float* escale, float* y, gaot_stream_t stream);
int gaot_gno_proj_backward(const float* dy, const float* k, const float* f, const float* weff, int32_t B, int32_t Q,
                           int32_t n_src, int32_t C, int32_t out_channels, const int32_t* index32, const int32_t* edge_query,
                           int32_t E, const int32_t* t_splits, const int32_t* t_edge, const float* escale, float* dk,
                           float* dweff_partial, float* df, gaot_stream_t stream);
/* dW[e,:] = escale[e] * sum_b dOut[b, edge_query[e], :] * f[b, index[e], :]   (escale may be NULL) */
int gaot_gno_edge_grad(const float* dout, const float* f, int32_t B, int32_t Q, int32_t n_src, int32_t C,
                       const int32_t* index32, const int32_t* edge_query, int32_t E,
                       const float* escale, float* dw, gaot_stream_t stream);
/* per-edge batched variants used by the 'nonlinear' transforms (agno.py:230-246):
 *   prod[b,e,:] = k[b,e,:] * f[b,index[e],:] (mul_f) * attn[e]      and the segment sum over e. */
int gaot_gno_segment_sum(const float* x, int32_t B, int32_t E, int32_t C, const int32_t* splits, int32_t Q,
                         const float* rowscale, float* out, gaot_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * processor pieces (attn.py)
 * ------------------------------------------------------------------------------------------ */
/* RMSNorm attn.py:161-172.  y = x * rstd * w ; rstd[M] saved for backward.  y_absmax (optional): the magnitude word of y
 * (gaot_gemm_desc.c_absmax conventions: zero before the launch) -- y is the A operand of the q|k|v / w1|w3 products. */
int gaot_rmsnorm_fwd(const float* x, const float* w, int32_t M, int32_t D, float eps,
                     float* y, float* rstd, float* y_absmax, gaot_stream_t stream);
/* dx = rstd*(w*dy - x*rstd^2*mean(w*dy*x)) (+ dx_add) (+ dx_add2) ; dw_partial[P,D] column partials (P returned rows =
 * gaot_rmsnorm_bwd_partials(M)); caller sums them.  dx_add / dx_add2 (optional): gradients reaching the SAME tensor by other
 * routes -- the block's residual branch (attn.py:228-233) and the long-range skip (attn.py:281-299) -- added here instead of by
 * separate elementwise launches. */
int gaot_rmsnorm_bwd_partials(int32_t M);
int gaot_rmsnorm_bwd(const float* x, const float* w, const float* rstd, const float* dy, const float* dx_add, const float* dx_add2,
                     int32_t M, int32_t D, float* dx, float* dw_partial, float* dx_absmax /* optional, as y_absmax */, gaot_stream_t stream);
/* The same with dy = dy_slabs[0] + ... + dy_slabs[n_slabs - 1] (slab_stride floats apart) (+ dy_add): dy is the output of a split-K
 * product left as raw K slabs (gaot_gemm_desc.raw_slabs) and of its fused residual -- the slab sum happens here, in slab order,
 * instead of in a reduce launch of its own.  D = 256, 384 or 512, 16-byte aligned pointers (else GAOT_ERR_INVALID: reduce, then
 * gaot_rmsnorm_bwd). */
int gaot_rmsnorm_bwd_slabs(const float* x, const float* w, const float* rstd, const float* dy_slabs, int32_t n_slabs, int64_t slab_stride,
                           const float* dy_add, const float* dx_add, const float* dx_add2, int32_t M, int32_t D, float* dx,
                           float* dw_partial, float* dx_absmax, gaot_stream_t stream);
/* SwiGLU gate attn.py:151: u = [u1 | u3] ([M,2F]); g = silu(u1)*u3 ; bwd writes du [M,2F]. */
int gaot_swiglu_fwd(const float* u, int32_t M, int32_t F, float* g, gaot_stream_t stream);
int gaot_swiglu_bwd(const float* u, const float* dg, int32_t M, int32_t F, float* du, gaot_stream_t stream);
/* out[i] = g[i] * act'(z[i]), act = GAOT_ACT_GELU (z = the saved pre-activation) or GAOT_ACT_RELU (z = the saved output): the derivative of
 * the LAST activation of an MLP chain (mlp.py:283-305 with a non-default final non-linearity), which has no following product to ride on */
int gaot_act_bwd(const float* g, const float* z, int64_t n, int32_t act, float* out, gaot_stream_t stream);
/* softmax attention, no mask, scale 1/sqrt(head_dim) (attn.py:98-116, F.scaled_dot_product_attention).
 * q/k/v are strided views: element (b,s,h,d) at ptr[(b*S+s)*ld + h*head_dim + d]; kv head = h / (H/Hkv).
 * o has the same addressing with ldo.  lse[B,H,S] (natural-log-sum-exp of scaled scores) saved for bwd.
 * head_dim <= 128, S arbitrary (head_dim <= 64: split-bf16 kernels where they apply; above: fp32 MFMA).
 * pieces: precision of the products on the matrix pipe, as gaot_gemm_desc.pieces: 0 / 3 = every fp32 operand (Q, K, V, dO and the
 * probabilities P / dS) as three bf16 pieces, exact to fp32 rounding; 2 = two rounded bf16 pieces each (16 significant bits); 4 = two
 * fp16 pieces of the scaled operand (head_dim 32; elsewhere, and without the words, it means 3): Q / K / V scaled through
 * qkv_absmax -- ONE magnitude word for the three of them (they are column blocks of one projection output; any bound works) --
 * dO through dout_absmax, P by 2^13, dS per 32 x 32 tile by the tile's own maximum. */
int gaot_attention_fwd(const float* q, const float* k, const float* v, int64_t ldq, int64_t ldk, int64_t ldv,
                       int32_t B, int32_t S, int32_t H, int32_t Hkv, int32_t head_dim,
                       float* o, int64_t ldo, float* lse, int32_t pieces, const float* qkv_absmax, gaot_stream_t stream);
/* [ABI 9] the same with a workspace (gaot_attention_fwd_workspace floats, 16-byte aligned; 0 floats / a null pointer: exactly
 * gaot_attention_fwd).  With pieces = 4, 32 < head_dim <= 64 and 128 .. 255 blocks of 256 queries x batch x heads (1 x 4 096 tokens x 8
 * heads of 48) the keys of a query block are shared between two workgroups and a second launch joins the halves
 * (o = w1 o1 + w2 o2 with w_i = exp(lse_i - lse)): the same softmax, eight waves per CU instead of four. */
int64_t gaot_attention_fwd_workspace(int32_t B, int32_t S, int32_t H, int32_t head_dim);
int gaot_attention_fwd_ws(const float* q, const float* k, const float* v, int64_t ldq, int64_t ldk, int64_t ldv,
                          int32_t B, int32_t S, int32_t H, int32_t Hkv, int32_t head_dim,
                          float* o, int64_t ldo, float* lse, int32_t pieces, const float* qkv_absmax, float* workspace, gaot_stream_t stream);
/* workspace floats needed by gaot_attention_bwd */
/* attention dropout (attn.py:110-114: dropout_p of F.scaled_dot_product_attention while training): the softmax output is
 * multiplied by keep(b,h,q,k) / (1 - p) before the product with V.  keep = (splitmix64(seed + ((b*H + h)*S + q)*S + k) >> 32) <
 * (1 - p) * 2^32 with seed read from DEVICE memory, so captured graphs draw a new mask at every replay:
 *   gaot_attention_seed_next : used[0] = mix(state[0], state[1] + 1, salt); state[1] += 1   (state = {seed, counter})
 *   gaot_attention_fwd_dropout / _bwd_dropout : as gaot_attention_fwd / _bwd (fp32-MFMA kernels, head_dim <= 128) with p_drop in (0, 1) and the
 *   device word written by gaot_attention_seed_next; the backward regenerates the forward's mask from the same word. */
int gaot_attention_seed_next(uint64_t* state, uint64_t salt, uint64_t* used, gaot_stream_t stream);
int gaot_attention_fwd_dropout(const float* q, const float* k, const float* v, int64_t ldq, int64_t ldk, int64_t ldv,
                               int32_t B, int32_t S, int32_t H, int32_t Hkv, int32_t head_dim, float* o, int64_t ldo,
                               float* lse, float p_drop, const uint64_t* seed, gaot_stream_t stream);
int gaot_attention_bwd_dropout(const float* q, const float* k, const float* v, int64_t ldq, int64_t ldk, int64_t ldv,
                               const float* o, const float* dout, int64_t ldo, const float* lse, int32_t B, int32_t S,
                               int32_t H, int32_t Hkv, int32_t head_dim, float* dq, float* dk, float* dv, int64_t lddq,
                               int64_t lddk, int64_t lddv, float* workspace, float p_drop, const uint64_t* seed,
                               gaot_stream_t stream);
int64_t gaot_attention_bwd_workspace(int32_t B, int32_t S, int32_t H, int32_t head_dim);
int gaot_attention_bwd(const float* q, const float* k, const float* v, int64_t ldq, int64_t ldk, int64_t ldv,
                       const float* o, const float* dout, int64_t ldo, const float* lse,
                       int32_t B, int32_t S, int32_t H, int32_t Hkv, int32_t head_dim,
                       float* dq, float* dk, float* dv, int64_t lddq, int64_t lddk, int64_t lddv,
                       float* workspace, int32_t pieces, const float* qkv_absmax, const float* dout_absmax,
                       float* dqkv_absmax /* optional: ONE magnitude word (zero before) for everything stored to dq, dk and dv */,
                       gaot_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * reductions / layout
 * ------------------------------------------------------------------------------------------ */
/* out[n] = sum_m x[m*ld + n]  (bias gradients).  scratch: gaot_colsum_scratch(M,N) floats. */
int64_t gaot_colsum_scratch(int32_t M, int32_t N);
int gaot_colsum(const float* x, int64_t ld, int32_t M, int32_t N, float* out, float* scratch, gaot_stream_t stream);
/* out[r,:] = sum_b x[b,r,:]   (row-periodic bias gradients; x is [B,R,N] contiguous) */
/* Grouped column sums: out_i[n] = sum_m x_i[m*ld_i + n] for n small partial-row matrices in ONE launch (the per-workgroup partial
 * rows behind the norm-weight gradients attn.py:161-172 and the lifting gradient magno.py:273-274, reduced at the end of a backward
 * pass).  N % 4 == 0, ld % 4 == 0, 16-byte aligned pointers; fixed summation order (deterministic). */
typedef struct gaot_colsum_item {
    const float* x; int64_t ld; float* out; int32_t M, N;
    int32_t out_cols; int64_t out_ld;   /* 0, 0: out is N contiguous floats; else the N sums are an [N / out_cols, out_cols] matrix stored with
                                           row stride out_ld (a column block of a wider gradient matrix); out_cols % 4 == 0 */
} gaot_colsum_item;
int gaot_colsum_grouped(const gaot_colsum_item* items, int32_t n, gaot_stream_t stream);
int gaot_batchsum(const float* x, int32_t B, int64_t RN, float* out, gaot_stream_t stream);
/* Fused row-wise MLP, every width 64: the kernel MLP of the integral transform (LinearChannelMLP, mlp.py:307-337, called per
 * edge at agno.py:229-231; act = GAOT_ACT_GELU) and the statistical geometry embedding followed by its recovery block
 * (gemb.py:54-59 + magno.py:345-350; act = GAOT_ACT_RELU):
 *   out[e,:] = W_L(... act(W_2 act(W_1 x[e,:] + b_1) + b_2) ...) + b_L,   `act` after every layer but the last.
 * x [E, c_in] (c_in <= 16), n_layers = 2..4, W_1 [64, c_in], every other W_i [64, 64], b_i [64]; w / b are HOST arrays of
 * n_layers device pointers.  Forward: one launch, activations never leave registers.  Backward: recomputes the chain,
 * returns the packed parameter gradient  grads = [dW_2 | .. | dW_L | dW_1 (64 x c_in) | db_1 | db_2 | .. | db_L]
 * ((n_layers-1)*4096 + 64*c_in + 64*n_layers floats) through `workspace` (gaot_kernel_mlp_bwd_workspace() floats:
 * per-workgroup partials summed in fixed order).  x gets no gradient (edge coordinates). */
int gaot_kernel_mlp_fwd(const float* x, int32_t E, int32_t c_in, int32_t n_layers, const float* const* w,
                        const float* const* b, int32_t act, float* out, gaot_stream_t stream);
int64_t gaot_kernel_mlp_bwd_workspace(int32_t E, int32_t c_in, int32_t n_layers);
/* rows of per-workgroup partial gradients the backward leaves in `workspace` (row stride = the parameter block size); passing
 * grads == workspace skips the final row sum: the caller reduces column blocks itself (gaot_colsum_grouped). */
int32_t gaot_kernel_mlp_bwd_rows(int32_t E);
int gaot_kernel_mlp_bwd(const float* x, int32_t E, int32_t c_in, int32_t n_layers, const float* const* w,
                        const float* const* b, int32_t act, const float* dk, float* grads, float* workspace, gaot_stream_t stream);
/* the same pair for layers narrower than 64: widths[i] = output width of layer i (multiples of 4 in 4..64, e.g. 48 lifting
 * channels); weight i is [widths[i]][widths[i-1]] row-major (layer 0: [widths[0]][cin]), out / dk are [E, widths[n-1]].  The chain
 * runs zero-padded at width 64; the gradient vector keeps its layout of 64 x 64 blocks, of which the leading
 * widths[i] x widths[i-1] corner is meaningful (the rest is zero).
 * ldw (optional, NULL = dense): ldw[i] = row stride of weight i in floats (0 = dense; a column block of a wider matrix -- the
 * geoembed half of the recovery weight, magno.py:345-350 -- is read in place).
 * pieces: precision of the 64-wide layers' products, as gaot_gemm_desc.pieces (0 / 3 = exact three-piece products, with the weight
 * gradient on the fp32 MFMA; 2 = two rounded pieces per operand everywhere, the weight gradient on the bf16 pipe as well). */
int gaot_kernel_mlp_fwd_w(const float* x, int32_t E, int32_t cin, int32_t n_layers, const float* const* w, const float* const* b,
                          int32_t act, const int32_t* widths, const int32_t* ldw, int32_t pieces, float* out, gaot_stream_t stream);
int gaot_kernel_mlp_bwd_w(const float* x, int32_t E, int32_t cin, int32_t n_layers, const float* const* w, const float* const* b,
                          int32_t act, const int32_t* widths, const int32_t* ldw, int32_t pieces, const float* dk, float* grads,
                          float* workspace, gaot_stream_t stream);
/* TWO chains in ONE launch each way: chain `a` = the kernel MLP of an integral transform (agno.py:229-231 -> mlp.py:307-337: 4 layers,
 * GELU, over the E edge rows), chain `b` = the geometry-embedding chain of the same transform with its half of the recovery block
 * (gemb.py:54-59 + magno.py:345-350: 3 layers, ReLU, over the Q query rows).  Neither reads the other's result, and on its own chain b is
 * a launch of 32 .. 128 workgroups on 256 CUs.  Arguments per chain as gaot_kernel_mlp_fwd_w / _bwd_w (forward: `out`; backward: `dk`,
 * `grads`, `workspace`, with grads == workspace leaving the per-workgroup partial rows to the caller).  Pairs the fused kernels do not cover
 * (other depths / activations / precisions) run as the two single launches: same results either way. */
typedef struct gaot_kmlp_desc {
    const float* x; int32_t E; int32_t cin; int32_t n_layers;
    const float* const* w; const float* const* b;
    int32_t act; const int32_t* widths; const int32_t* ldw; int32_t pieces;
    float* out;
    const float* dk; float* grads; float* workspace;
} gaot_kmlp_desc;
int gaot_kernel_mlp_fwd_pair(const gaot_kmlp_desc* a, const gaot_kmlp_desc* b, gaot_stream_t stream);
int gaot_kernel_mlp_bwd_pair(const gaot_kmlp_desc* a, const gaot_kmlp_desc* b, gaot_stream_t stream);
/* nn.MSELoss() with mean reduction (the reference trainers' loss, base_trainer.py:71): loss[0] = mean((pred - target)^2) over
 * n elements through `partial` (>= 256 floats; fixed-order two-stage sum, deterministic); backward
 * dpred = 2 (pred - target) / n * grad_loss[0] with grad_loss a DEVICE scalar (so the launch replays inside a hipGraph). */
int gaot_mse_loss_fwd(const float* pred, const float* target, int64_t n, float* partial, float* loss, gaot_stream_t stream);
int gaot_mse_loss_bwd(const float* pred, const float* target, int64_t n, const float* grad_loss, float* dpred, gaot_stream_t stream);
/* forward and the gradient for a unit seed (d loss = 1) in ONE launch: loss[0] = mean (pred - target)^2 with the bits gaot_mse_loss_fwd
 * gives, dpred = 2 (pred - target) / n.  partial: 256 floats of scratch; ticket: one int32, ZERO before the first call (the kernel
 * returns it to zero); tick (optional): a device-resident optimizer step counter advanced by one by this launch (the tick of
 * gaot_adamw_step_dev, folded in: follow with gaot_adamw_apply_dev). */
int gaot_mse_loss_fwd_bwd(const float* pred, const float* target, int64_t n, float* partial, int32_t* ticket, float* loss, float* dpred,
                          float* tick, gaot_stream_t stream);
/* One AdamW update over flat fp32 buffers with torch.optim.AdamW semantics (the reference's optimizer, optimizers.py:196;
 * defaults beta = (0.9, 0.999), eps = 1e-8).  step[1] is a DEVICE counter (float) advanced by the call itself, so the launch
 * replays inside a hipGraph. */
int gaot_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                    float eps, float weight_decay, float* step, gaot_stream_t stream);
/* Same update with the hyper-parameters read from DEVICE memory at run time: hyper[5] = {lr, beta1, beta2, eps, weight_decay}.
 * A captured launch therefore follows the reference's per-epoch LR schedulers (optimizers.py:199-245) without re-capture. */
int gaot_adamw_step_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper, float* step,
                        gaot_stream_t stream);
/* The update alone, for a step counter something else has already advanced (gaot_mse_loss_fwd_bwd's `tick`): step is only read. */
int gaot_adamw_apply_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper, const float* step,
                         gaot_stream_t stream);
/* patchify gaot.py:182-185,202-205 and its inverse gaot.py:224-231.  Latent grid H x W (x Dz; Dz = 0 for 2-D).
 * inverse = 0: in = grid[b, (h,w[,z]), c]  -> out = tokens[b, s, (p..., c)];  inverse = 1: the other way. */
int gaot_patchify(const float* in, int32_t B, int32_t H, int32_t W, int32_t Dz, int32_t P, int32_t C,
                  float* out, int32_t inverse, float* out_absmax /* optional: magnitude word of `out` (zero before) */, gaot_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Edge-partitioned forms of the fused integral-transform kernels (csrc/gno_ep.hip): the edge list is cut into equal chunks, a
 * lane group walks one chunk with a running (segmented) sum per sample, rows that span chunks are completed from carry slots
 * by a second small kernel in chunk order.  Same results as the row-parallel forms up to summation order inside a row;
 * the work per lane group no longer depends on the degree distribution (BASELINE config 3: rows of 350+ edges next to empty
 * rows).  ws: gaot_gno_ep_workspace(E, C, B) floats.
 *   gaot_gno_lift_gather_reduce_ep : as gaot_gno_lift_gather_reduce (+ edge_query[E])
 *   gaot_gno_proj_gather_t_ep      : the dF product of gaot_gno_proj_backward over the transposed CSR
 *   gaot_gno_proj_gather_t_ep_w    : [ABI 10] the same, publishing dF's magnitude word (df_absmax: gaot_gemm_desc.c_absmax conventions, zero
 *                                    before the launch; NULL = the plain form) -- dF is the A operand of the processor's last input-gradient
 *                                    product, which otherwise spends a launch on its maximum
 *   (both: e_real, optional DEVICE scalar = the number of edges actually in the list, <= E: the launch, the chunking and the workspace
 *    are sized for E, edges past *e_real are not walked -- the padded unions of gaot_union_compose)
 *   gaot_gno_proj_gather_reduce_bin: gaot_gno_proj_gather_reduce with the batch INSIDE the lane group (each kernel-value row is
 *                                    read once for 4 samples instead of once per sample).  row_order (optional, [Q]): a permutation
 *                                    of the rows that puts rows with common source rows next to each other (e.g. sorted by first
 *                                    neighbour): the rows are walked in that order, a contiguous range per XCD -- same results,
 *                                    the gathered feature rows stay in the L2 that first fetched them
 * ------------------------------------------------------------------------------------------ */
int64_t gaot_gno_ep_workspace(int32_t E, int32_t C, int32_t B);
int gaot_gno_lift_gather_reduce_ep(const float* k, const float* pn, const float* wl, const float* bl, int32_t B, int32_t n_src,
                                   int32_t c_in, int32_t C, const int32_t* splits, const int32_t* cols, const int32_t* edge_query,
                                   int32_t Q, int32_t E, const float* escale, float* out, float* ws, const int32_t* e_real,
                                   gaot_stream_t stream);
int gaot_gno_proj_gather_t_ep(const float* k, const float* dy, const float* weff, int32_t B, int32_t Q, int32_t n_src, int32_t C,
                              int32_t out_channels, const int32_t* index32, const int32_t* edge_query, int32_t E,
                              const int32_t* t_splits, const int32_t* t_edge, const float* escale, float* df, float* ws,
                              const int32_t* e_real, gaot_stream_t stream);
int gaot_gno_proj_gather_t_ep_w(const float* k, const float* dy, const float* weff, int32_t B, int32_t Q, int32_t n_src, int32_t C,
                                int32_t out_channels, const int32_t* index32, const int32_t* edge_query, int32_t E,
                                const int32_t* t_splits, const int32_t* t_edge, const float* escale, float* df, float* ws,
                                const int32_t* e_real, float* df_absmax, gaot_stream_t stream);
int gaot_gno_proj_gather_reduce_bin(const float* k, const float* f, const float* weff, const float* rowbias, const float* bias,
                                    int32_t B, int32_t n_src, int32_t C, int32_t out_channels, const int32_t* splits,
                                    const int32_t* cols, int32_t Q, const float* escale, float* y, const int32_t* row_order,
                                    gaot_stream_t stream);

/* The decoder's output projection folded into the recovery block (magno.py:345-350 then 640-641: two linear maps, nothing in between):
 *   weff[OC, Cout] = W[OC, C] Wr_a[C, Cout],   rproj[Q, OC] = rowb[Q, C] W^T + b        (the transform kernels above apply them)
 * and, in ONE launch, every gradient of that node: drowb = g_rproj W, dW = g_rproj^T rowb + g_weff Wr_a^T, db = column sums of g_rproj,
 * dWr_a = W^T g_weff (any output may be NULL).  OC <= 4, C % 4 == 0, C <= 256.  workspace: gaot_proj_fold_workspace(Q, C, OC) floats;
 * ticket: one int32, ZERO before the first call (the kernel returns it to zero).  Fixed summation orders (deterministic). */
int32_t gaot_proj_fold_workspace(int32_t Q, int32_t C, int32_t out_channels);
int gaot_proj_fold_fwd(const float* hw, int64_t ldh, const float* hb, const float* wa, int64_t lda, const float* rowb, int64_t ldr,
                       int32_t Q, int32_t C, int32_t Cout, int32_t out_channels, float* weff, float* rproj, gaot_stream_t stream);
int gaot_proj_fold_bwd(const float* g_weff, const float* g_rproj, const float* hw, int64_t ldh, const float* wa, int64_t lda,
                       const float* rowb, int64_t ldr, int32_t Q, int32_t C, int32_t Cout, int32_t out_channels, float* drowb,
                       float* dhw, int64_t ld_dhw, float* dhb, float* dwa, int64_t ld_dwa, float* workspace, int32_t* ticket,
                       gaot_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Operator variants off the default configuration (csrc/glue.hip).
 * ------------------------------------------------------------------------------------------ */
/* RoPE on queries and keys (attn.py:106-108; rotary_embedding_torch.RotaryEmbedding(dim=head_dim).rotate_queries_or_keys: the
 * position is the SEQUENCE INDEX, pairs (2i, 2i+1) of a head are rotated by pos * theta^(-2i/head_dim)).  In place on the first
 * n_heads heads of every row of x [B*S, ld] (the fused q|k|v projection output: n_heads = H + H_kv).
 * cos_sin [S, head_dim/2, 2] = (cos, sin) of the angles.  inverse = 1: the transposed rotation (backward pass). */
int gaot_rope_inplace(float* x, int32_t B, int32_t S, int64_t ld, int32_t n_heads, int32_t head_dim, const float* cos_sin,
                      int32_t inverse, gaot_stream_t stream);
/* dot-product attention scores (agno.py:215-217): score[e] = scale * <qn[edge_query[e], :], kn[index[e], :]>, qn [Q,C], kn [n_src,C]. */
int gaot_edge_dot_score(const float* qn, const float* kn, int32_t C, const int32_t* index32, const int32_t* edge_query,
                        int32_t E, float scale, float* score, gaot_stream_t stream);
/* learned attention through the linear transform: with T[e,:] = sum_b dOut[b,eq[e],:] * f[b,index[e],:] (gaot_gno_edge_grad without a
 * scale), da[e] = <T[e,:], k[e,:]> and T[e,:] *= a[e] in place (T becomes dk). */
int gaot_edge_rowdot_scale(float* T, const float* k, const float* a, int32_t E, int32_t C, float* da, gaot_stream_t stream);
/* PointNet pooling (gemb.py:217, scatter_max): out[q,c] = max over the CSR segment of h[e,c], 0 for an empty segment;
 * backward shares the gradient evenly among the edges attaining the maximum. */
int gaot_segment_max_fwd(const float* h, int32_t C, const int32_t* splits32, int32_t Q, float* out, gaot_stream_t stream);
int gaot_segment_max_bwd(const float* h, const float* out, const float* dout, int32_t C, const int32_t* splits32, int32_t Q,
                         float* dh, gaot_stream_t stream);
/* backward of gaot_gno_segment_sum: dx[b,e,:] = rowscale[edge_query[e]] * dout[b, edge_query[e], :]. */
int gaot_segment_broadcast(const float* dout, int32_t B, int32_t E, int32_t C, int32_t Q, const int32_t* edge_query,
                           const float* rowscale, float* dx, gaot_stream_t stream);
/* multiscale mixing (magno.py:291-303): out = sum_i w[q,i] * scales[i][b,q,:] (w NULL: mean over the n <= 8 scales);
 * backward: dscales[i] = w[q,i] * dout (NULL entries skipped), dw[q,i] = sum_{b,c} dout * scales[i] (dw NULL: skipped).
 * `scales` / `dscales` are HOST arrays of device pointers. */
int gaot_scale_mix_fwd(const float* const* scales, int32_t n, const float* w, int32_t B, int32_t Q, int32_t C, float* out,
                       gaot_stream_t stream);
int gaot_scale_mix_bwd(const float* const* scales, float* const* dscales, int32_t n, const float* w, int32_t B, int32_t Q,
                       int32_t C, const float* dout, float* dw, gaot_stream_t stream);
/* 'nonlinear' / 'nonlinear_kernelonly' transforms (agno.py:230-271): kernel values per sample, k [B,E,C].
 *   gaot_edge_cat      x[b,e,:] = [feat[e,:W0], f[b,index[e],:C]]                       (rows of the kernel MLP)
 *   gaot_gno_bk_reduce out[b,q,:] = sum_{e in seg(q)} escale[e] * k[b,e,:] * (mul_f ? f[b,index[e],:] : 1)
 *   gaot_gno_bk_backward  dk[b,e,:] = escale[e] * dout[b,eq[e],:] * (mul_f ? f : 1);
 *                         df[b,j,:] = sum_{e: index[e]=j} (mul_f ? escale[e] k dout : 0) + dx[b,e,W0:]   (dx = gradient of the MLP rows, may be NULL);
 *                         dscale[e] = sum_{b,c} dout * k * (mul_f ? f : 1)              (NULL outputs are skipped) */
int gaot_edge_cat(const float* feat, int32_t W0, const float* f, int32_t B, int32_t n_src, int32_t C, const int32_t* index32,
                  int32_t E, float* x, gaot_stream_t stream);
int gaot_gno_bk_reduce(const float* k, const float* f, int32_t B, int32_t n_src, int32_t C, int32_t E, const int32_t* splits32,
                       const int32_t* index32, int32_t Q, const float* escale, int32_t mul_f, float* out, gaot_stream_t stream);
int gaot_gno_bk_backward(const float* dout, const float* k, const float* f, const float* dx, int32_t W0, int32_t B, int32_t n_src,
                         int32_t C, int32_t E, int32_t Q, const int32_t* index32, const int32_t* edge_query,
                         const int32_t* t_splits, const int32_t* t_edge, const float* escale, int32_t mul_f, float* dk,
                         float* df, float* dscale, gaot_stream_t stream);
/* ConditionedNorm (mlp.py:74-124): y[b,s,:] = x[b,s,:] * scale[b,:] + shift[b,:]; backward writes dx and per-chunk partial sums
 * part[chunks, B, 2*D] = (sum_s dy*x | sum_s dy) with chunks = gaot_cond_affine_bwd_chunks(S). */
int gaot_cond_affine_fwd(const float* x, const float* scale, const float* shift, int32_t B, int64_t S, int32_t D, float* y,
                         gaot_stream_t stream);
int32_t gaot_cond_affine_bwd_chunks(int64_t S);
int gaot_cond_affine_bwd(const float* x, const float* dy, const float* scale, int32_t B, int64_t S, int32_t D, float* dx,
                         float* part, gaot_stream_t stream);

/* autoregressive rollout glue (gaot.py:371-388, 432, 436-476), one pass each over the [rows = B*N] node rows:
 *   gaot_rollout_input : pn[r,:] = [state[r,:U], stat[r,:S], t0n, dtn]   (n_time = 2), or without the dtn column (n_time = 1:
 *                        conditioned-norm models feed the time through `condition`)
 *   gaot_rollout_update: den = de-normalised stepper result (mode 0 'output': pred*u_std+u_mean; 1 'residual': (state*u_std+u_mean)
 *                        + (pred*a_std+a_mean); 2 'time_der': ... + dt*(pred*a_std+a_mean)), then state = (den - u_mean) / u_std in
 *                        place; every operation rounded separately like the reference's tensor expression. */
int gaot_rollout_input(const float* state, int32_t U, const float* stat, int32_t S, float t0n, float dtn, int32_t n_time,
                       int64_t rows, float* pn, gaot_stream_t stream);
int gaot_rollout_update(const float* pred, float* state, int32_t U, const float* u_mean, const float* u_std, const float* a_mean,
                        const float* a_std, float dt, int32_t mode, int64_t rows, float* den_out, gaot_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GAOT_HIP_H */
