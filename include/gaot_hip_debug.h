/*
 * gaot_hip_debug.h -- tuning / ablation hooks of libgaot_hip.so.  NOT part of the data path and NOT part of the drop-in
 * boundary (include/gaot_hip.h): these setters change process-global kernel-selection state, so they are neither thread-safe
 * nor re-entrant across streams.  Only tools/, bench.py's separately reported `--dtype bf16` variant and A/B tests call them;
 * gaot_amd's modules never do.  Every setter returns the previous value.
 */
#ifndef GAOT_HIP_DEBUG_H
#define GAOT_HIP_DEBUG_H

#include "gaot_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* tuning hook (not part of the data path): force the GEMM tile; 0 = heuristic. Returns the previous value. */
int gaot_debug_set_gemm_tile(int cfg);
/* tuning hook: ablate parts of the GEMM kernel (results become WRONG): 1 no in-loop loads, 2 no LDS staging, 4 no stores */
int gaot_debug_set_gemm_ablate(int bits);
/* which kernel family served the calling thread's last gaot_gemm_f32: 1 = fp32 MFMA tiles, 2 = skinny VALU path,
 * 3 = split-bf16 MFMA tiles (fp32 operands split exactly into three bf16 pieces, six bf16 MFMAs per product, fp32-level error) */
int gaot_debug_last_gemm_path(void);
/* tuning hook: 0 = register-staged fp32 tiles only, 1 = + LDS-direct fp32 tiles (2-stage ring), 3 = same with a 3-stage
 * ring, 4 = + split-bf16 tiles where the heuristic picks them (DEFAULT), 5 = split-bf16 wherever eligible, 6 = split-bf16
 * for the SwiGLU-gate product only */
int gaot_debug_set_gemm_glds(int on);
/* 3 (default): fp32-level products from three bf16 pieces per operand; 1: operands rounded to bf16, one piece product, fp32
 * accumulation -- the separately reported `bench.py --dtype bf16` variant only (BASELINE configs[1]); returns the old value. */
int gaot_debug_set_gemm_pieces(int pieces);
/* 0: ignore gaot_gemm_desc.b_planes (same-box A/B of the pre-split weight planes; bit-identical results); returns the old value */
int gaot_debug_set_gemm_planes(int on);
/* all-DMA fp16-piece tiles (gemm_ad.hip): 0 off, 1 per the heuristic (default), 2 / 3: 64- / 128-row tiles wherever eligible; returns the old value */
int gaot_debug_set_gemm_ad(int on);
/* 64 x 64 all-DMA tiles for narrow outputs (N <= 256), bits: 1 (default) launches that 64 x 128 tiles leave half filled (4 096-token
 * batches), 2 also K <= 256 at full launches (off: slower at step level); 0 none (same-box A/B; see the dispatcher in gemm.hip);
 * returns the old value */
int gaot_debug_set_gemm_ad_narrow(int on);
/* attention backward, head_dim 32, fp16 pieces: 1 (default) lets two workgroups share the query tiles of a 256-key block when the launch
 * would otherwise hold 128 .. 255 workgroups (4 x 1 024 tokens x 8 heads); 0: the 4-wave kernel as before (same-box A/B); returns the old value */
int gaot_debug_set_attention_qsplit(int on);
/* fp16-piece products: output tiles (of the split tile kernels and the grouped weight-gradient launch) that took the per-row second pass
 * since the counter was last reset -- a tile whose operand rows span more than 2^13 in magnitude is recomputed with one power-of-two
 * scale per row (gemm_split.hip).  Synchronises the device.  reset != 0: zero the counter after reading it. */
unsigned gaot_debug_split_redo_count(int reset);
/* grouped weight gradients: values of k per workgroup (K slab length; multiple of 32, default 4096) */
/* (k > 0: a fixed K slab for every product of the grouped launch; 0: automatic; -c: the cap of longer products' slabs (default 4 096);
 * -(100000 + c): the same for node-level products, K > 16 384) */
int gaot_debug_set_wgrad_kslab(int k);
/* [r6] grouped weight gradients on fp16 pieces: 256 = 256 x 128 tiles on eight waves (one workgroup per CU) when every product's M is a
 * multiple of 256; 128 = 128 x 128 tiles, two workgroups per CU */
int gaot_debug_set_gemm_ad_flush(int on);        /* [r6] narrow outputs with a reduction of 1 025 .. 2 048: 1 (default) = unsplit on the 64 x 64 all-DMA tiles, which flush their accumulators every 1 024 values of k; 0 = K slabs as before; negative = query; returns the old value */
int gaot_debug_set_wgrad_tile_rows(int bm);
/* [r6] grouped weight gradients, automatic K slabs: 0 (default) = the rule of rounds 3-5, 1 = a cost model (rounds x longest K loop + a per-slab term: measured slower, gemm_split.hip) */
int gaot_debug_set_wgrad_slab_rule(int r);
/* tuning hook: head_dim 32 attention, 1 = split-bf16 MFMA kernels (default), 0 = fp32-MFMA kernels, 2 / 3 = split with the
 * 8-wave / 4-wave forward workgroup forced.  Returns the previous value. */
int gaot_debug_set_attention_split(int on);
int gaot_debug_set_attention_keysplit(int on);  /* [r6] the key-split forward of gaot_attention_fwd_ws: 1 (default) = on for the shapes it is for (head_dim 36 .. 64), 2 = head_dim 32 too (tests), 0 = off (the workspace query then answers 0); returns the old value */
int gaot_debug_set_attention_dh8(int on);       /* [r6] 32 < head_dim <= 64 on fp16 pieces: 1 (default) = the 8-wave 256-key backward (attn_bwd_split8_dh_kernel<true, QS>: query-split at 128 .. 255 key blocks x batch x heads), 0 = the 4-wave 128-key kernel; returns the old value */
int gaot_debug_set_attention_h16(int mode);     /* [r6] the fp16-piece backward at >= 256 key blocks: 8 + 16 VAR = attn_bwd_h16_kernel<8, 1, VAR> (default 0x18), 4 = <4> (two workgroups per CU), 0 = attn_bwd_split8_kernel; returns the old mode */
/* head_dim-32 split attention: pieces of P in the forward and of P / dS in the backward products, as 10 * forward + backward:
 * 22 (default) = two rounded pieces (five piece products instead of six) in both; 33 = exact three-way splits; 32, 23 for A/B runs.
 * Returns the previous value. */
int gaot_debug_set_attention_p_pieces(int n);
/* head_dim-32 split attention: pieces of the Q / K / V / dO operands: 2 (default) = two rounded pieces (with two-piece P / dS: three
 * piece products everywhere), 3 = exact three-way splits.  Returns the previous value. */
int gaot_debug_set_attention_operand_pieces(int n);
/* A/B hook: 1 (default) = the two-piece head_dim-32 backward reads dO^T / Q^T / dS through transposing LDS reads, 0 = separately staged
 * transposed planes.  Same results.  Returns the previous value. */
int gaot_debug_set_attention_tr(int on);
/* tuning hook: 1 = the software-pipelined 8-wave split forward where it applies (S % 64 == 0), 0 (default) = the plain one.
 * Returns the previous value. */
int gaot_debug_set_attention_pipe(int on);
/* tuning hook (results become WRONG): forward kernel 1 = no stores, 2 = no GELU, 4 = no MFMA layers, 8 = no weight staging */
int gaot_debug_set_kernel_mlp_ablate(int bits);
/* A/B hook: 1 (default) = the bf16-split kernel-MLP kernels (fp32-level products on v_mfma_f32_32x32x16_bf16), 0 = the fp32-MFMA
 * kernels.  Same results to fp32 rounding either way.  Returns the previous value. */
int gaot_debug_set_kernel_mlp_split(int on);
int gaot_debug_set_ep_chunk(int edges_per_chunk);      /* tuning only: 0 = default (32) */

#ifdef __cplusplus
}
#endif
#endif
