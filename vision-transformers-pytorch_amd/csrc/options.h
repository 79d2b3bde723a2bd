// Dispatch switches of libvtx.so (include/vtx.h: vtx_set_option / vtx_get_option).
//
// The ONLY process-global mutable state of the library: a table of atomic ints, read once per entry-point call with
// a relaxed load (no getenv caches frozen at first use -- a test flips a switch in-process and compares the kernel
// variants bit for bit).  Initial values come from the VTX_* environment variables of the same name at library load.
#pragma once

enum VtxOptionId {
  VTX_OPT_GEMM_GLDS = 0,        // 1: bf16 forward-layout GEMMs take the LDS-DMA kernel (gemm_glds.hip); 0: register-staged
  VTX_OPT_GLDS_BM = 1,          // 0: tile height per launch (glds_pick_bm) | 64 | 128
  VTX_OPT_GLDS_WAVES = 2,       // 8: 2 x 4 waves per 128-column tile | 4: 2 x 2 waves
  VTX_OPT_WGRAD_GLDS = 3,       // 1: LDS-DMA + transpose-read weight-gradient kernel; 0: register-staged
  VTX_OPT_WG_WAVES = 4,         // 8 | 4 waves per weight-gradient workgroup
  VTX_OPT_WGRAD_BLOCKS = 5,     // split-K target: workgroups per weight-gradient launch (512 = one resident round)
  VTX_OPT_SATTN = 6,            // 1: ViT fast path sattn_* (D = 64, L <= 224); 0: generic attn_*
  VTX_OPT_WATTN_FWD_WAVES = 7,  // persistent window-attention forward: resident waves (4096)
  VTX_OPT_WATTN_BWD_WAVES = 8,  // ... backward (2048)
  VTX_OPT_SRATTN_WGS = 9,       // PVT spatial-reduction attention: target workgroups (2048)
  VTX_OPT_WATTN_XCD_MAJOR = 10,     // 1: window-attention workgroups ordered head-fastest per XCD (the heads sharing a
                                    //    128-byte line run back to back on one L2); 0: all blocks of head 0, then head 1, ...
  VTX_OPT_LN_FIT = 11,              // LayerNorm exact-fit lane groups for C = 384 / 768: bit 0 forward, bit 1 backward
  VTX_OPT_GLDS_EPI = 12,            // LDS-DMA GEMM epilogue (128-column tiles, 8 waves): 1 = wave-private staging, no workgroup barrier (default) | 0 = shared staging passes
  VTX_OPT_SATTN_WAVES = 13,         // ViT attention fast path: 8 waves on single 16-token tiles (default) | 7, 6 | 1 = whichever of 6 / 7 / 8 leaves the fewest idle tile slots (round 6: L = 197 -> 7; measured no faster, profiles/round6_sattn_wave_counts.txt) | 4 waves on pairs of tiles
  VTX_OPT_WATTN_BWD4 = 14,          // 1: bf16 window-attention backward with four waves per problem (wattn_bwd4_kernel) | 0: one wave
  VTX_OPT_GEMM_SKINNY = 15,         // 1: bf16 GEMMs with K = 64 / 96 / 128 over >= 32 768 rows take the weight-resident streaming kernel (gemm_skinny.hip) | 0
  VTX_OPT_GEMM_ASTAT = 16,          // 1: bf16 GEMMs with 192 <= K <= 384 (K % 64 == 0, N % 128 == 0, N >= 256) and >= 2 tiles per CU take the A-stationary kernel (gemm_astat.hip) | 2: any row count | 3: >= 1.25 tiles per CU | 0
  VTX_OPT_TWINS_SUB_LDS = 17,       // 1: the Twins sub-sampling gather / scatter staged through LDS (one workgroup per row of patches) where the geometry allows | 0: element-wise
  VTX_OPT_WGRAD_WIDE = 18,          // 1: grouped weight gradients made of whole 128 x 384 tiles (C = 384 layers) take the wide-tile kernel, one workgroup per CU | 2: the same with its multiplying waves in two groups half a k-step apart | +4 (5 | 6): only those (the round-4 rule); without it 128 x 64 J tiles, J = 6, 5, 3 (+8: 4 too; without it J = 4 only as the >= 75 % fallback of groups whose widths divide by 384, C = 768), for every group whose Kin are multiples of one of them (C = 192 / 320 layers too; C = 256 / 512 with +8), ragged N | bits 4-11: fill threshold in percent instead of 85 (probe: 1209 = 256-column tiles + 75 %) | 0: 128 x 128 tiles
  VTX_OPT_GEMM_PP = 19,             // 1: bf16 GEMMs with N % 192 == 0, K % 64 == 0, long contractions (K >= 1152, or K >= 768 with N <= 384) and >= 3/4 of a CU-filling round of 128 x 192 tiles take the two-group kernel (gemm_pp.hip) | 2: any row count, any K | 10W: forced tile height 32 W | 0
  VTX_OPT_LN_ROWS = 20,             // LayerNorm rows in flight per lane group: bit 0 forward, two rows of the three-vector groups (C = 384 / 768 exact fit) | bit 1 backward, two rows of the one- / two-vector groups
  VTX_OPT_SKINNY_WAVES = 21,        // waves per persistent workgroup of the weight-resident streaming GEMMs (gemm_skinny.hip): 4 | 8 | 16
  VTX_OPT_WATTN_FAST = 22,          // window-attention forward: bit 0, 7 x 7 windows run their 49th query as one row (4 scores per lane) instead of a padded 16-query tile | bit 1, windows of a masked layer whose tokens share one region id take the unmasked instruction stream
  VTX_OPT_WATTN_FWD4 = 23,          // 1: the bf16 window-attention forward shares a problem between the four waves of its workgroup (one 16-token tile per wave, rotating; next-problem prefetch) when a head has >= 4 096 (image, window) problems | 2: always | 0: one wave per problem
  VTX_OPT_MLP_FUSED = 24,          // 1: the MLP of bf16 layers with C = 64 / 96 over >= 32 768 rows runs as ONE launch each way with both weights resident in LDS (mlp_fused.hip: z and h never stored by the forward, recomputed by the backward) inside the one-call layers | 100 b + f: kernel variants (forward f in {4, 8, 9, 12, 16}, backward b in {4, 8}: mlp_fused.hip) | 0: four GEMM launches
  VTX_OPT_LN_FOLD = 25,            // LayerNorm launches folded into their neighbours (round 6): bit 0, the norm_ff backward runs in the epilogue of the fused-MLP backward (mlp_bwd_kernel<.., LNB>: dln2 never stored, dx1 bit-identical to the stand-alone launch; layers that take the fused MLP) | bit 1, the qkv input gradient of a narrow window-attention layer (C = 64 / 96 / 128, >= 32 768 rows) and the norm_attn backward run as one weight-resident streaming launch (gemm_skinny.hip dgrad_ln_kernel: dln1 never stored, dx bit-identical) | bit 2, norm_ff FORWARD on the row operands of the fused-MLP forward (mlp_fwd_kernel<.., LNF>: x1 in, ln2 / mean / rstd / y out) | bit 3, a narrow norm_attn forward on the row operands of the weight-resident streaming qkv GEMM (gemm_skinny_kernel<.., LNF>) | 0: stand-alone launches
  VTX_OPT_COUNT = 26
};

int vtx_opt(int id);   // current value (relaxed atomic load); capi.hip
int vtx_cu_count_cached();   // compute units of the current device (256 on MI355X; 256 when no device can be asked); capi.hip
