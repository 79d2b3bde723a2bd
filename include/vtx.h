/* vtx.h -- C ABI of libvtx.so: hand-written gfx950 (MI355X / CDNA4) kernels for the ViT / Swin
 * training hot path of rosinality/vision-transformers-pytorch.
 *
 * The reference has no FFI layer: its boundary is the Python nn.Module surface (SURVEY.md
 * section 8(b)).  These entry points are what the MI355X-native nn.Modules in
 * vision-transformers-pytorch_amd/models/ call (through ctypes, vtx/_lib.py) in place of the stock
 * PyTorch op compositions of the reference; each declaration cites the reference lines it replaces.
 *
 * Conventions (all entry points):
 *   - return 0 on success, a negative VTX_ERR_* code otherwise (vtx_strerror); never throw, never exit
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch caching allocator), including
 *     workspaces whose size comes from the matching vtx_*_workspace(); no ownership transfer
 *   - kernels are enqueued on `stream` (a hipStream_t passed as void*) and never synchronise; re-entrant across
 *     streams / threads (autograd's backward thread).  Process-global mutable state, all of it DIAGNOSTIC: (1) the table
 *     of dispatch switches behind vtx_set_option (atomic ints, read once per call: a switch flipped while another thread
 *     is inside an entry point takes effect at that thread's next call); (2) the launch timer of vtx_timer_start /
 *     vtx_timer_stop (csrc/layer.hip: one unsynchronised record list -- while a timer is open, vtx_layer_* / vtx_srlayer_*
 *     calls append to it from whichever thread enqueues them, so open it only around single-threaded, single-stream
 *     sections: bench.py's event-sampled steps; with no timer open the layer calls touch no shared state)
 *   - dtype: VTX_F32 (parity mode, exact-fp32 MFMA) or VTX_BF16 (training mode: bf16 storage,
 *     fp32 accumulation / statistics); parameters, their gradients and all statistics are fp32
 *   - activations are row-major [rows, C] (tokens x channels) = NHWC / (B, L, C)
 */
#ifndef VTX_H_
#define VTX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VTX_F32 0
#define VTX_BF16 1

#define VTX_OK 0
#define VTX_ERR_SHAPE (-1)
#define VTX_ERR_DTYPE (-2)
#define VTX_ERR_ALIGN (-3)
#define VTX_ERR_LAUNCH (-4)
#define VTX_ERR_WORKSPACE (-5)
#define VTX_ERR_NULL (-6)

const char* vtx_strerror(int code);
/* ABI version of the library (bumped on any signature change). */
int vtx_abi_version(void);
/* Compute units of the current device (256 on MI355X): what the one-workgroup-per-CU kernels size their grids and the
 * dispatch heuristics their thresholds with; the Python mirrors of those heuristics (vtx/ops.py) ask here. */
int vtx_cu_count(void);
/* Test helper: fill the LDS of every CU with `pattern` (160-KB workgroups, `rounds` per CU): LDS is not cleared between workgroups, and a
 * kernel that reads LDS it never wrote sees what the previous workgroup left there (tests/test_gpu_lds_poison.py). */
int vtx_debug_lds_poison(unsigned pattern, int rounds, void* stream);
/* Measurement helper (bench.py `measured_peaks`, SURVEY.md section 8(d): "ideally a measured MFMA micro-benchmark on the box"): one launch of
 * waves_per_cu / 4 256-thread workgroups per CU (waves_per_cu in 4, 8, .. 32), every wave issuing 4 * iters dense bf16 MFMAs
 * (v_mfma_f32_32x32x16_bf16, 32 768 FLOP each) on register operands, no memory traffic.  *flops receives the FLOPs of the launch; the caller
 * brackets the call with events on `stream`.  `sink`: any device buffer of >= 4 bytes. */
int vtx_debug_mfma_peak(int iters, int waves_per_cu, void* sink, double* flops, void* stream);
/* Measurement helper: one idle wavefront that occupies `stream` for `microseconds` of wall clock (<= 100 000) and touches no memory.  Two of
 * them on two streams finish in one period when the streams run concurrently, in two when they share a hardware queue: vtx.functional picks
 * the side stream of the weight gradients with it (round 6, profiles/round6_side_stream_queue.md). */
int vtx_debug_spin(int microseconds, void* stream);
/* waves per workgroup the ViT attention fast path (csrc/attention_seq.hip) runs sequences of L tokens with under the current SATTN_WAVES option
 * (round 6: 6 / 7 / 8 so that the live 16-token tiles divide among the waves; the bindings name the kernel instantiation with it) */
int vtx_sattn_waves(int L);

/* ---- Dispatch switches (csrc/options.h).  Which kernel variant an entry point launches -- LDS-DMA vs register-staged
 * GEMM, tile height, waves per workgroup, split-K target, fused vs separate split-K reduction, persistent-grid sizes --
 * is a table of ints, initialised from the environment variable of the same name (vtx_option_name: "VTX_GLDS_BM", ...)
 * at library load and changeable in-process: every variant computes the same function (most of them bit-identically,
 * tests/test_gpu_dispatch.py), the switches exist for measurement and for parity tests at a forced dispatch.
 * (The reference has no counterpart: its dispatch is torch's.) */
int vtx_option_count(void);
const char* vtx_option_name(int id);
int vtx_get_option(int id);              /* -1 for an unknown id */
int vtx_set_option(int id, int value);   /* VTX_ERR_SHAPE for an unknown id */

/* ---- LayerNorm (reference: nn.LayerNorm at models/vit.py:13, models/swin_transformer.py:12,
 * 206, 221, 277).  y = (x - mean) * rstd * gamma + beta over the last dim, biased variance.
 * merge != 0: x is (B, H, W, C/4) NHWC and row (b, i, j) is the 2x2 patchify gather of
 * PatchMerge (models/swin_transformer.py:15-22, 224) -- the 4C-wide row is never materialised.
 * Saves mean / rstd [rows] for the backward. */
int vtx_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                      int64_t rows, int C, float eps, int dtype, int merge, int H, int W, void* stream);
/* Over the kept samples of a stochastic-depth branch only (csrc/layer.hip drives these): logical row r is row
 * perm[r / T] * T + r % T of every row-indexed tensor, perm [samples] int32 on the device with the kept samples first. */
int vtx_layernorm_fwd_mapped(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                             int64_t rows, int C, float eps, int dtype, const int* perm, int T, void* stream);
int vtx_layernorm_bwd_mapped(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                             const void* dres, void* dx, void* workspace, size_t ws_bytes, int64_t rows, int64_t live, int C,
                             int dtype, const int* perm, int T, void* stream);
size_t vtx_layernorm_bwd_workspace(int64_t rows, int C);
/* dx = dres + LN'(dy)  (dres optional, plain layout only), dgamma / dbeta fp32 [C] (overwritten).
 * dgamma == dbeta == NULL defers the final column reduce: the per-block partials stay in the workspace as
 * [vtx_layernorm_bwd_blocks(rows, C)][2 C] fp32 and the caller sums them -- together with the layer's other small
 * reductions -- in one vtx_colreduce_multi launch (same summation order, same bits). */
int vtx_layernorm_bwd_blocks(int64_t rows, int C);
int vtx_colreduce_multi(int n, const float* const* part, float* const* out0, float* const* out1, const int* nb,
                        const int* C, const int* ld, void* stream);
int vtx_layernorm_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                      const void* dres, void* dx, float* dgamma, float* dbeta, void* workspace, size_t ws_bytes,
                      int64_t rows, int C, int dtype, int merge, int H, int W, void* stream);

/* ---- Linear layers (reference: nn.Linear / nn.Conv2d-as-GEMM at models/vit.py:23-25, 73,
 * models/swin_transformer.py:34-35, 205, 222, 281, models/layer.py:191-196) with fused epilogues.
 *   mode 0 (forward): C[M,N] = epi( A[M,K] . B[N,K]^T )        A = activations, B = weight
 *   mode 1 (dgrad)  : C[M,N] = epi( A[M,K] . B[K,N]   )        A = dy, B = weight [K = out, N = in]
 *   epi(v): v += bias[col]; act 1: aux_out = z = v, v = silu(z) (MLP, models/layer.py:191-196);
 *           act 2: v *= silu'(aux_in[row,col]);
 *           v *= rowscale[row / rows_per_scale] (DropPath, models/layer.py:172-180);
 *           v += resid[row, col] (residual add, models/vit.py:60-61, swin_transformer.py:194-195)
 * bias / resid / rowscale / aux_out / aux_in may be NULL.  lda/ldb/ldc in elements; N, K and the leading dimensions
 * must be multiples of 8 (16-byte vector access everywhere). */
int vtx_gemm(int mode, int dtype, const void* A, const void* B, void* C, int M, int N, int K, int64_t lda,
             int64_t ldb, int64_t ldc, const float* bias, const void* resid, const float* rowscale,
             int rows_per_scale, void* aux_out, const void* aux_in, int act, void* stream);
/* ---- Fused MLP of the narrow stages (csrc/mlp_fused.hip; reference models/layer.py:186-196 inside models/swin_transformer.py:193-197):
 * bf16, C = 64 / 96, ff % 32 == 0, both weights resident in LDS, >= 32 768 rows (vtx_mlp_fused_ok tells; option MLP_FUSED).
 *   vtx_mlp_fwd: y = resid + rowscale[row / rows_per_scale] * (silu(ln2 . w1^T + b1) . w2^T + b2); z (the bf16 pre-activation) and
 *                h = silu(z) are written only when their pointers are given (the one-call layers pass NULL: nothing ff-wide is stored);
 *   vtx_mlp_bwd: z and h recomputed from ln2; dz = rowscale * (dy . w2) * silu'(z) and h written for the weight gradients
 *                (dW2 = dy^T h, dW1 = dz^T ln2: vtx_wgrad_group), dln2 = dz . w1.
 * w1 [ff][C], w2 [C][ff] bf16 (the forward-layout copies); b1 / b2 / resid / rowscale may be NULL.  Element values: those of the
 * four vtx_gemm launches they replace, bit for bit. */
int vtx_mlp_fused_ok(int dtype, int64_t M, int C, int ff);
int vtx_mlp_fwd(int dtype, const void* ln2, const void* w1, const float* b1, const void* w2, const float* b2, const void* resid,
                const float* rowscale, int rows_per_scale, void* y, void* z, void* h, int64_t M, int C, int ff, void* stream);
int vtx_mlp_bwd(int dtype, const void* ln2, const void* dy, const void* w1, const float* b1, const void* w2, const float* rowscale,
                int rows_per_scale, void* h, void* dz, void* dln2, int64_t M, int C, int ff, void* stream);
/*   vtx_mlp_bwd_ln (round 6, option LN_FOLD): vtx_mlp_bwd with the LayerNorm backward of the block's norm_ff (models/swin_transformer.py:196:
 *                out + drop_path(ff(norm_ff(out)))) in its epilogue -- dx1 = dy + LN'(dln2) with LN' over x1 / mean / rstd / gamma, the bits of
 *                vtx_mlp_bwd followed by vtx_layernorm_bwd(dln2, x1, mean, rstd, gamma, dres = dy); dln2 is never stored.  part: the
 *                [part_rows][2 C] fp32 dgamma | dbeta partial rows of vtx_layernorm_bwd's deferred reduce (vtx_colreduce_multi); part_rows >=
 *                vtx_cu_count(): the launch writes one row per workgroup and zeroes the others. */
int vtx_mlp_bwd_ln(int dtype, const void* ln2, const void* dy, const void* w1, const float* b1, const void* w2, const float* rowscale,
                   int rows_per_scale, void* h, void* dz, const void* x1, const float* mean, const float* rstd, const float* gamma,
                   void* dx1, float* part, int part_rows, int64_t M, int C, int ff, void* stream);
/* ---- Narrow-layer input gradient with the LayerNorm backward in its epilogue (round 6, option LN_FOLD bit 1; csrc/gemm_skinny.hip):
 * dx = dres + LN'(dy . W) -- the bits of vtx_gemm(dy [M, K], wt [C, K] = W^T) followed by vtx_layernorm_bwd(dln, x, mean, rstd, gamma,
 * dres); the intermediate dln [M, C] is never stored.  bf16, C in {64, 96, 128}, K = 3 C (a packed qkv projection, reference
 * models/swin_transformer.py:128,194) or K = C.  part: as in vtx_mlp_bwd_ln. */
int vtx_dgrad_ln(int dtype, const void* dy, const void* wt, const void* x, const float* mean, const float* rstd, const float* gamma,
                 const void* dres, void* dx, float* part, int part_rows, int64_t M, int C, int K, void* stream);
/* ---- LayerNorm FORWARD folded into its consumer (round 6, option LN_FOLD bits 2, 3): the norm runs on the row operands of a row-streaming
 * launch (csrc/ln_fold.h); the normalised rows and the statistics are stored on the side for the backward; outputs bit-identical to
 * vtx_layernorm_fwd followed by the consumer.
 *   vtx_mlp_fwd_ln: LN(x1; gamma, beta, eps) -> ln2, mean, rstd;  y = x1 + rowscale * MLP(ln2)      (vtx_layernorm_fwd + vtx_mlp_fwd)
 *   vtx_ln_gemm:    LN(x) -> ln_out, mean, rstd;  y = ln_out . w^T + bias, w [N][C]                   (vtx_layernorm_fwd + vtx_gemm)
 * bf16; C in {64, 96} (MLP) / {64, 96, 128} (GEMM, N % 32 == 0). */
int vtx_mlp_fwd_ln(int dtype, const void* x1, const float* gamma, const float* beta, float eps, void* ln2, float* mean, float* rstd,
                   const void* w1, const float* b1, const void* w2, const float* b2, const float* rowscale, int rows_per_scale, void* y,
                   int64_t M, int C, int ff, void* stream);
int vtx_ln_gemm(int dtype, const void* x, const float* gamma, const float* beta, float eps, void* ln_out, float* mean, float* rstd,
                const void* w, const float* bias, void* y, int64_t M, int C, int N, void* stream);
size_t vtx_wgrad_workspace(int64_t mtok, int N, int Kin);
/* dW[N,Kin] = sum_m s[m] dy[m,:]^T x[m,:] (fp32, overwritten); dbias[N] = sum_m s[m] dy[m,:] (optional,
 * computed inside the same kernel).  s = rowscale[m / rows_per_scale] or 1.  scale_const > 0 declares that
 * every rowscale value is either 0 or scale_const (DropPath: mask / (1 - p)), which lets the LDS-DMA kernel
 * skip dropped samples' rows instead of scaling; pass 0 for arbitrary scales.
 * Deterministic: split-K over tokens into fp32 slabs in `workspace`, summed in slice order by ONE following reduce launch.
 * dW needs 16-byte alignment (the kernels store 16-byte vectors): VTX_ERR_ALIGN otherwise. */
int vtx_wgrad(int dtype, const void* dy, const void* x, float* dW, float* dbias, int64_t mtok, int N, int Kin,
              int64_t ld_dy, int64_t ld_x, const float* rowscale, int rows_per_scale, float scale_const,
              void* workspace, size_t ws_bytes, void* stream);
/* Grouped form: the weight gradients of nprob <= vtx_wgrad_group_max() linears over the SAME mtok tokens in ONE launch
 * -- the four of a transformer layer's backward (fc2, fc1, proj, qkv: models/vit.py:59-63, swin_transformer.py:193-197,
 * layer.py:191-196).  Split-K only exists to fill the chip, so four problems together need a quarter of the slices (and
 * of the fp32 slab traffic) of one.  bf16, every (N[i], Kin[i]) a multiple of 8 and >= 64 (vtx_wgrad_group_ok tells;
 * otherwise call vtx_wgrad per problem).  Arrays are HOST arrays of nprob entries; dbias / rowscale may be NULL or hold
 * NULL entries; rowscale values in {0, scale_const} (scale_const > 0) as for vtx_wgrad; same determinism contract.
 * ncol (0..4) deferred column reductions (the arguments of vtx_colreduce_multi: a layer's LayerNorm dgamma / dbeta partials,
 * its rel_pos gradient partials) ride in the group's ONE reduce launch -- same bits as vtx_colreduce_multi, one launch less
 * per layer.  They must have been enqueued on `stream` (or be ordered before it) like the operands.
 * accumulate != 0 (needs vtx_wgrad_group_slices() >= 2): every output (dW, dbias, col_out0 / 1) becomes out + result -- a
 * parameter's second gradient inside one backward (DINO's multi-crop backbone, train_dino.py:229-236) lands on the first
 * instead of in a tensor of its own that autograd then has to add (one `add` launch per parameter). */
int vtx_wgrad_group_max(void);
int vtx_wgrad_group_ok(int dtype, int nprob, const int* N, const int* Kin, int64_t mtok, int has_rowscale,
                       int rows_per_scale, float scale_const);
size_t vtx_wgrad_group_workspace(int nprob, const int* N, const int* Kin, int64_t mtok);
int vtx_wgrad_group(int dtype, int nprob, const void* const* dy, const void* const* x, float* const* dW,
                    float* const* dbias, const int* N, const int* Kin, const int64_t* ld_dy, const int64_t* ld_x,
                    const float* const* rowscale, int rows_per_scale, float scale_const, int64_t mtok,
                    void* workspace, size_t ws_bytes, int ncol, const float* const* col_part, float* const* col_out0,
                    float* const* col_out1, const int* col_nb, const int* col_C, const int* col_ld, int accumulate,
                    void* stream);
/* ... over the KEPT samples of stochastic-depth branches (csrc/layer.hip drives it): problem i with perm[i] != NULL
 * contracts over the mtok_kept[i] tokens of its kept samples only, in perm[i] order (logical token t = row
 * perm[i][t / rows_per_scale] * rows_per_scale + t % rows_per_scale of dy[i] and x[i]; dropped samples' rows are never read)
 * and multiplies its result by scale[i]; rowscale[i] must be NULL for it.  perm == NULL: exactly vtx_wgrad_group. */
int vtx_wgrad_group_mapped(int dtype, int nprob, const void* const* dy, const void* const* x, float* const* dW,
                           float* const* dbias, const int* N, const int* Kin, const int64_t* ld_dy, const int64_t* ld_x,
                           const float* const* rowscale, const int* const* perm, const int* mtok_kept, const float* scale,
                           int rows_per_scale, float scale_const, int64_t mtok, void* workspace, size_t ws_bytes, int ncol,
                           const float* const* col_part, float* const* col_out0, float* const* col_out1, const int* col_nb,
                           const int* col_C, const int* col_ld, int accumulate, void* stream);
/* split-K slices such a group runs with (>= 2: its outputs come from the reduce launch, so `accumulate` is available) */
int vtx_wgrad_group_slices(int nprob, const int* N, const int* Kin, int64_t mtok);

/* ---- Attention cores.  qkv is the QKV-projection output [rows, 3*nH*D] with channel order
 * [q|k|v][head][d] (models/vit.py:30-34, models/swin_transformer.py:128); o is [rows, nH*D].
 *   swin == 0 (reference models/vit.py:30-42): L tokens per image, rows = B*L.
 *   swin != 0 (reference models/swin_transformer.py:109-154): (H, W) NHWC feature map, win x win
 *     windows (L = win*win), shift != 0 selects the rolled partition (roll by -win/2 and back are
 *     folded into addressing); bias = [nH][L][L] fp32 from vtx_relpos_bias; mask = local_mask
 *     buffer [nW][L][L] bytes (True = -inf) or NULL.
 * lse [B*nW*nH*L] fp32 (log-sum-exp per query row) is saved for the backward.
 * Global attention (swin == 0, no bias / mask) runs at ANY L: up to 224 tokens on the register-resident kernels, beyond
 * (e.g. 577 = ViT-S/16 at 384 x 384, models/vit.py:153-175) on key-block / online-softmax kernels (csrc/attention_long.hip);
 * their backward needs vtx_attention_bwd_workspace() bytes of workspace (B*nH*L floats). */
int vtx_relpos_bias(const float* rel_pos, const int64_t* pos, float* bias, int L, int nH, void* stream);
int vtx_attention_fwd(const void* qkv, void* o, float* lse, const float* bias, const uint8_t* mask, int B, int L,
                      int nH, int D, int swin, int H, int W, int win, int shift, int dtype, void* stream);
size_t vtx_attention_bwd_workspace(int B, int L, int nH, int swin, int H, int W, int win);
/* global attention over Bk <= B images only (the b-th one is image perm[b]; lse indexed by b): bf16, D = 64, L <= 224 */
int vtx_attention_fwd_mapped(const void* qkv, void* o, float* lse, const int* perm, int Bk, int L, int nH, int D, int dtype,
                             void* stream);
int vtx_attention_bwd_mapped(const void* qkv, const void* o, const void* dout, const float* lse, void* dqkv, const int* perm,
                             int Bk, int L, int nH, int D, int dtype, void* stream);
/* dqkv [rows, 3*nH*D] (fully overwritten).  With bias: drel_pos [ntab, nH] fp32 = dense gradient of
 * the rel_pos embedding (models/swin_transformer.py:46,135), scattered through the CSR (order[L*L],
 * offsets[ntab+1]) of the pos table; deterministic (no atomics). */
int vtx_attention_bwd(const void* qkv, const void* o, const void* dout, const float* lse, const float* bias,
                      const uint8_t* mask, const int* csr_order, const int* csr_offsets, void* dqkv,
                      float* drel_pos, int ntab, void* workspace, size_t ws_bytes, int B, int L, int nH, int D,
                      int swin, int H, int W, int win, int shift, int dtype, void* stream);

/* ---- Dropout of the attention probabilities (reference models/vit.py:39, models/swin_transformer.py:144: F.dropout(attn, p,
 * training) on the softmax output; models/pvt.py:60, models/twins.py:88,147 likewise).  Same arguments as vtx_attention_fwd / _bwd
 * plus (drop_p, seed, keep): cell (problem, query, key) is kept when a counter-based hash of (seed, problem, query * L + key) says
 * so -- the backward regenerates the decision from the same (drop_p, seed), no mask tensor exists in HBM -- and kept
 * probabilities are scaled by 1 / (1 - drop_p).  problem = (image * nW + window) * nH + head.  keep != NULL replaces the hash by
 * an explicit mask [problems][L][L] of bytes (1 = keep): the parity tests pass the mask the reference drew.  0 < drop_p < 1.
 * Global attention of any length (key-block kernels beyond 224 tokens), window attention of <= 64 tokens (head dim 32 | 64) or <= 160
 * tokens (head dim 32).
 * vtx_attn_keep_mask writes the hash's decisions [nprob][Lq][Lk] (what the kernels regenerate) for tests / an external checker. */
int vtx_attention_fwd_drop(const void* qkv, void* o, float* lse, const float* bias, const uint8_t* mask, int B, int L,
                           int nH, int D, int swin, int H, int W, int win, int shift, int dtype, float drop_p, uint64_t seed,
                           const uint8_t* keep, void* stream);
int vtx_attention_bwd_drop(const void* qkv, const void* o, const void* dout, const float* lse, const float* bias,
                           const uint8_t* mask, const int* csr_order, const int* csr_offsets, void* dqkv,
                           float* drel_pos, int ntab, void* workspace, size_t ws_bytes, int B, int L, int nH, int D,
                           int swin, int H, int W, int win, int shift, int dtype, float drop_p, uint64_t seed,
                           const uint8_t* keep, void* stream);
int vtx_attn_keep_mask(uint8_t* out, int64_t nprob, int Lq, int Lk, float drop_p, uint64_t seed, void* stream);

/* ---- Window attention fast path (head dim 32, window <= 7x7): one wavefront per (image, window, head) problem,
 * persistent 4-wave workgroups per head, same semantics as vtx_attention_fwd/bwd with swin != 0
 * (reference models/swin_transformer.py:103-160: roll, partition, q k^T / sqrt(d) + rel_pos(pos) bias,
 * masked_fill(local_mask, -inf), softmax, @ v, inverse partition, roll back).
 *   rel_pos [(2 win - 1)^2][nH] fp32 (the nn.Embedding weight), pos [L][L] int64 (the module's `pos` buffer);
 *   region: NULL for un-shifted layers, else [nW][64] uint8 with
 *           local_mask[n][a][b] == (region[n][a] != region[n][b])   (swin_transformer.py:82-89, 138-141)
 *           -- vtx.tables.mask_regions derives the ids from the module's local_mask buffer and verifies the
 *           identity; a mask without that structure takes vtx_attention_* instead.
 * The head's bias table is gathered into LDS once per workgroup; no per-call table kernels.
 * The backward writes dqkv fully and drel_pos [(2 win - 1)^2][nH] (deterministic). */
int vtx_wattn_fwd(const void* qkv, void* o, float* lse, const float* rel_pos, const int64_t* pos,
                  const uint8_t* region, int B, int L, int nH, int H, int W, int win, int shift, int dtype,
                  void* stream);
int vtx_wattn_fwd_mapped(const void* qkv, void* o, float* lse, const float* rel_pos, const int64_t* pos,
                         const uint8_t* region, const int* perm, int Bk, int L, int nH, int H, int W, int win, int shift,
                         int dtype, void* stream);     /* the b-th image worked on is image perm[b], b < Bk */
int vtx_wattn_bwd_mapped(const void* qkv, const void* o, const void* dout, const float* lse, const float* rel_pos,
                         const int64_t* pos, const uint8_t* region, void* dqkv, void* workspace, size_t ws_bytes,
                         const int* inv_cells, int inv_count, const int* perm, int Bk, int L, int nH, int H, int W, int win,
                         int shift, int dtype, void* stream);
size_t vtx_wattn_bwd_workspace(int B, int nH, int H, int W, int win);
/* drel_pos == NULL defers the rel_pos-gradient reduce: vtx_wattn_bwd_parts() partial rows of stride vtx_wattn_bwd_part_ld(nH)
 * stay in the workspace (columns (2 win - 1)^2 * nH) for vtx_colreduce_multi. */
int vtx_wattn_bwd_parts(int B, int nH, int H, int W, int win);
int vtx_wattn_bwd_part_ld(int nH);
/* inv_cells [inv_count][(2 win - 1)^2] int32 (device) or NULL: the inverse of pos -- inv_cells[t][b] = the t-th cell
 * q * 64 + key (row stride 64) with pos[q][key] == b in ascending (q, key) order, padded with the cell L * 64; inv_count = the
 * largest bin (vtx.tables.pos_inverse).  With it the rel_pos gradient is gathered per bin inside the kernel; NULL: LDS-atomic
 * scatter (~27 us more per launch, same result up to fp32 summation order). */
int vtx_wattn_bwd(const void* qkv, const void* o, const void* dout, const float* lse, const float* rel_pos,
                  const int64_t* pos, const uint8_t* region, void* dqkv, float* drel_pos, void* workspace,
                  size_t ws_bytes, const int* inv_cells, int inv_count, int B, int L, int nH, int H, int W, int win,
                  int shift, int dtype, void* stream);

/* ---- One call per transformer layer (csrc/layer.hip): the pre-LN block of models/vit.py:59-63 and
 * models/swin_transformer.py:193-197 --  x1 = x + s1 * proj(attn(qkv(LN1 x)));  y = x1 + s2 * fc2(silu(fc1(LN2 x1)))  -- as
 * ONE descriptor: vtx_layer_fwd enqueues the 7 launches, vtx_layer_bwd the 8 + 2 of the backward (the entry points above, in
 * the order vtx.functional.TransformerLayerFn issues them: bit-identical results), so that the host pays one call and one
 * activation buffer per layer instead of ~19 calls and ~28 allocations (host time per Swin-S step 13.9 -> see DESIGN).
 * All pointers are device addresses the caller keeps alive; weights (wq, wo, w1, w2: [out][in]) are in the compute dtype,
 * the transposed copies (w?t: [in][out]) may be NULL (then every dgrad takes the register-staged kernel); s1 / s2 are the
 * per-sample DropPath scales (NULL: none) with values in {0, scale_const}.
 *   attn_kind VTX_ATTN_WINDOW: vtx_wattn_fwd / _bwd (rel_pos, pos, region, H, W, win, shift; B images, L = win^2);
 *             VTX_ATTN_GLOBAL: vtx_attention_fwd / _bwd without bias / mask (B images of L tokens, head dim C / nH).
 * Backward: dz / dln2 / dx1 / dout / dqkv / dln1 are scratch; ln?_ws (vtx_layernorm_bwd_workspace), attn_ws
 * (vtx_wattn_bwd_workspace / vtx_attention_bwd_workspace) and wgrad_ws (vtx_wgrad_group_workspace for the problems
 * (C, ff), (ff, C), (C, C), (3C, C)) are workspaces; dW? / db? / dg? / dbe? (/ drel) receive the parameter gradients (fp32).
 * `side_stream` != NULL sends the grouped weight gradient + its reduce there behind an event fork; the caller joins.
 * accumulate: as for vtx_wgrad_group. */
enum { VTX_ATTN_WINDOW = 1, VTX_ATTN_GLOBAL = 2 };
typedef struct VtxLayerFwd {
  int dtype, attn_kind;
  int64_t M;                                   /* tokens */
  int C, ff, nH, L, B, rows_per_scale;
  int H, W, win, shift;
  float eps;
  const void* x;
  const float *ln1_w, *ln1_b, *ln2_w, *ln2_b;
  const void *wq, *wo, *w1, *w2;
  const float *bq, *bo, *b1, *b2;
  const float* rel_pos; const int64_t* pos; const uint8_t* region;
  const float *s1, *s2;
  void *ln1, *qkv, *o, *x1, *ln2, *z, *h, *y;  /* z may be NULL (no backward will run) */
  float *mean1, *rstd1, *mean2, *rstd2, *lse;
  /* stochastic-depth compaction (both NULL: off): perm1 / perm2 [B] int32 on the device list the samples of the attention /
   * MLP branch, kept ones (s? != 0) first, Bk1 / Bk2 >= 1 of them -- each branch is computed for its kept samples only, the
   * others pass through (bf16; window attention, or global attention with head dim 64 and L <= 224; C and ff multiples of
   * 128; rows_per_scale = tokens per sample) */
  const int *perm1, *perm2;
  int Bk1, Bk2;
} VtxLayerFwd;
int vtx_layer_fwd(const VtxLayerFwd* a, void* stream);
typedef struct VtxLayerBwd {
  int dtype, attn_kind;
  int64_t M;
  int C, ff, nH, L, B, rows_per_scale;
  int H, W, win, shift;
  float scale_const;
  int accumulate, inv_count;
  const void *dy, *x, *ln1, *qkv, *o, *x1, *ln2, *z, *h;
  const float *mean1, *rstd1, *mean2, *rstd2, *lse;
  const float *ln1_w, *ln2_w;
  const void *wq, *wo, *w1, *w2, *wqt, *wot, *w1t, *w2t;
  const float* rel_pos; const int64_t* pos; const uint8_t* region; const int* inv_cells;
  const float *s1, *s2;
  void *dz, *dln2, *dx1, *dout, *dqkv, *dln1, *dx;
  void *ln1_ws, *ln2_ws, *attn_ws, *wgrad_ws;
  size_t ln_ws_bytes, attn_ws_bytes, wgrad_ws_bytes;
  float *dWq, *dbq, *dWo, *dbo, *dW1, *db1, *dW2, *db2, *dg1, *dbe1, *dg2, *dbe2, *drel;
  const int *perm1, *perm2;                    /* the forward's (see VtxLayerFwd) */
  int Bk1, Bk2;
  const float* b1;                             /* fc1 bias: a forward that ran the fused MLP (vtx_mlp_fwd) kept no z -- the backward recomputes it */
} VtxLayerBwd;
int vtx_layer_bwd(const VtxLayerBwd* a, void* stream, void* side_stream);
int vtx_layer_desc_bytes(int which);
/* The same for the layers built around the spatial-reduction (sub-sampled) attention: one PVT block (reference
 * models/pvt.py:31-68, 99-103) or the global half of a Twins-SVT layer (models/twins.py:56-93, 201-202):
 *   x1 = x + s1 * proj(sr_attn(q(LN1 x), kv(reduce(LN1 x))))      y = x1 + s2 * fc2(silu(fc1(LN2 x1)))
 * reduce (r > 1) = operand gather (vtx_patchify_fwd, or vtx_twins_subsample_fwd when twins != 0) + GEMM + bias [+ LayerNorm
 * when srn_w != NULL]; splitk != 0: the reduction conv runs as vtx_wgrad on the transposed operand copy + vtx_bias_cast.
 * r == 1: keys / values come from LN1's output directly (Lk = L).  linear_q / linear_kv have no bias.  The caller owns every
 * buffer; the backward's weight gradients run as two grouped launches (the M-token problems with the LayerNorm column
 * reductions; the B*Lk-token problems of the reduction branch) on `side_stream` when given, like vtx_layer_bwd. */
typedef struct VtxSrLayerFwd {
  int dtype, twins;
  int64_t M;                                   /* B * L tokens */
  int C, ff, nH, L, B, rows_per_scale, H, W, r, skip, Lk, splitk;
  float eps;
  const void* x;
  const float *ln1_w, *ln1_b, *ln2_w, *ln2_b, *srn_w, *srn_b;
  const void *wq, *wkv, *wsr, *wsr_t, *wo, *w1, *w2;
  const float *bsr, *bo, *b1, *b2;
  const float *s1, *s2;
  void *ln1, *q, *patches, *patches_t, *red32, *red, *kvin, *kv, *o, *x1, *ln2, *z, *h, *y;
  float *mean1, *rstd1, *mean2, *rstd2, *means, *rstds, *lse;
  void* splitk_ws;
  size_t splitk_ws_bytes;
} VtxSrLayerFwd;
int vtx_srlayer_fwd(const VtxSrLayerFwd* a, void* stream);
typedef struct VtxSrLayerBwd {
  int dtype, twins;
  int64_t M;
  int C, ff, nH, L, B, rows_per_scale, H, W, r, skip, Lk, reserved;
  float scale_const;
  const void *dy, *x, *ln1, *q, *patches, *red, *kvin, *kv, *o, *x1, *ln2, *z, *h;
  const float *mean1, *rstd1, *mean2, *rstd2, *means, *rstds, *lse;
  const float *ln1_w, *ln2_w, *srn_w;
  const void *wq, *wkv, *wsr, *wo, *w1, *w2, *wqt, *wkvt, *wsrt, *wot, *w1t, *w2t;
  const float *s1, *s2;
  void *dz, *dln2, *dx1, *dout, *dq, *dkv, *dkvin, *dred, *dpatches, *dln1, *dx;
  void *ln1_ws, *ln2_ws, *lns_ws, *attn_ws, *wgrad_ws, *wgrad2_ws;
  size_t ln_ws_bytes, lns_ws_bytes, attn_ws_bytes, wgrad_ws_bytes, wgrad2_ws_bytes;
  float *dWq, *dWkv, *dWsr, *dbsr, *dWo, *dbo, *dW1, *db1, *dW2, *db2, *dg1, *dbe1, *dg2, *dbe2, *dgs, *dbs;
  const float* b1;                             /* fc1 bias (the fused MLP's backward recomputes z: see VtxLayerBwd) */
} VtxSrLayerBwd;
int vtx_srlayer_bwd(const VtxSrLayerBwd* a, void* stream, void* side_stream);
/* sizeof the descriptors: 0 / 1 VtxLayerFwd / Bwd, 2 / 3 VtxSrLayerFwd / Bwd */
/* Per-launch HIP-event timing of what vtx_layer_* enqueues (bench.py's roofline block describes the kernels of the timed
 * path): between vtx_timer_start() and vtx_timer_stop() every launch of a layer call is bracketed by two events on its
 * stream; _stop synchronises and returns up to `cap` records.  tag: VTX_T_*; rows = rows / tokens the launch computes;
 * GEMM: C[rows, n] over k; attention: n heads, k tokens per problem; flags: 1 residual, 2 z written, 4 z read, 8 row-mapped
 * (compacted branch), 16 shifted-window mask. */
enum { VTX_T_LN_FWD = 1, VTX_T_GEMM = 2, VTX_T_WATTN_FWD = 3, VTX_T_ATTN_FWD = 4, VTX_T_LN_BWD = 5, VTX_T_WATTN_BWD = 6,
       VTX_T_ATTN_BWD = 7, VTX_T_WGRAD = 8,
       /* vtx_srlayer_*: sub-sampled attention (rows = B*Lq queries, n heads, k keys); its layer's grouped weight gradients over
        * rows tokens: _SR_A = fc2, fc1, proj, q [, kv when flags & 1] with n = C, k = ff; _SR_B = kv + reduction conv with n = C,
        * k = r*r*C; _SPLITK = ONE problem dW[n, k] over rows tokens (the reduction conv's forward on the split-K launch);
        * _GATHER = operand gather / scatter of rows x n elements */
       VTX_T_SRATTN_FWD = 9, VTX_T_SRATTN_BWD = 10, VTX_T_WGRAD_SR_A = 11, VTX_T_WGRAD_SR_B = 12, VTX_T_WGRAD_SPLITK = 13,
       VTX_T_GATHER = 14,
       /* the fused MLP of the narrow stages (vtx_mlp_fwd / vtx_mlp_bwd inside vtx_layer_*): rows, n = C, k = ff */
       VTX_T_MLP_FWD = 15, VTX_T_MLP_BWD = 16,
       /* round 6: a dgrad with the LayerNorm backward in its epilogue (csrc/gemm_skinny.hip dgrad_ln_kernel): rows x C (n) over k */
       VTX_T_DGRAD_LN = 17,
       /* a LayerNorm forward on the row operands of the GEMM that consumes it (gemm_skinny.hip, LNF): rows x n over k = C */
       VTX_T_LN_GEMM = 18 };
typedef struct VtxTimerRec { int tag, n, k, flags; int64_t rows; float ms; } VtxTimerRec;
int vtx_timer_start(void);
int vtx_timer_stop(VtxTimerRec* out, int cap);   /* sizeof(VtxLayerFwd) (0) / sizeof(VtxLayerBwd) (1), for bindings */

/* ---- Spatial-reduction (cross) attention of PVT (csrc/attention_sr.hip; reference models/pvt.py:38-66) and the global
 * sub-sampled attention of Twins-SVT (models/twins.py:56-93): head dim D = 64 | 32, Lq queries against Lk reduced keys per
 * (image, head): up to 64 keys on the register-resident kernels (every 224 x 224 configuration), more (PVT at 256 x 256 stage 4:
 * 65; at 384 x 384: 144 / 145; Twins at 448 x 448: 256) on key-block / online-softmax kernels (csrc/attention_long.hip).
 *   q [B*Lq, nH*D] (= linear_q output), kv [B*Lk, 2*nH*D] (= linear_kv output: k | v halves, pvt.py:51, twins.py:74),
 *   o [B*Lq, nH*D], lse [B*nH*Lq] fp32 (saved for the backward).
 * The backward writes dq and dkv fully; key-side partials are summed in fixed order (deterministic). */
int vtx_srattn_fwd(const void* q, const void* kv, void* o, float* lse, int B, int Lq, int Lk, int nH, int D, int dtype,
                   void* stream);
/* score [B, nH, Lq, Lk] = q k^T / sqrt(D) before the softmax: the second value pvt.MultiHeadedAttention.forward returns
 * (models/pvt.py:53, 69; the PVT layers discard it).  Inference-side helper: no backward. */
int vtx_srattn_scores(const void* q, const void* kv, void* score, int B, int Lq, int Lk, int nH, int D, int dtype,
                      void* stream);
size_t vtx_srattn_bwd_workspace(int B, int Lq, int Lk, int nH, int D);
int vtx_srattn_bwd(const void* q, const void* kv, const void* o, const void* dout, const float* lse, void* dq, void* dkv,
                   void* workspace, size_t ws_bytes, int B, int Lq, int Lk, int nH, int D, int dtype, void* stream);

/* the same with dropout of the attention probabilities (models/pvt.py:60, models/twins.py:88): see vtx_attention_fwd_drop;
 * problem = image * nH + head, cell = query * Lk + key, keep [B*nH][Lq][Lk] */
int vtx_srattn_fwd_drop(const void* q, const void* kv, void* o, float* lse, int B, int Lq, int Lk, int nH, int D, int dtype,
                        float drop_p, uint64_t seed, const uint8_t* keep, void* stream);
int vtx_srattn_bwd_drop(const void* q, const void* kv, const void* o, const void* dout, const float* lse, void* dq, void* dkv,
                        void* workspace, size_t ws_bytes, int B, int Lq, int Lk, int nH, int D, int dtype, float drop_p,
                        uint64_t seed, const uint8_t* keep, void* stream);

/* ---- Halo (blocked local) attention (csrc/halo.hip, csrc/attention_long.hip; reference models/halo_transformer.py:22-115): the
 * queries of a win x win block attend to the (win + 2 halo)^2 neighbourhood around it -- out-of-image positions are ZERO key / value
 * rows that take part in the softmax (F.unfold's padding, halo_transformer.py:70-76) -- plus rel_pos[pos[q][k]][head].
 *   vtx_window_gather : dst [B * nW, (win + 2 halo)^2, nc] from channels [c0, c0 + nc) of the channels-last map [B, H, W, ld]
 *                       (halo = 0: the window partition of the queries);  vtx_window_scatter: its adjoint (sums over the
 *                       neighbourhoods that hold a token, fixed order; halo = 0: the inverse partition); c0, nc, ld multiples of 8.
 *   vtx_table_bias / _bwd: bias[h][cell] = table[pos[cell]][h] for any cell count, and the dense table gradient through the CSR of pos.
 *   vtx_xattn_fwd / _bwd: softmax(q k^T / sqrt(D) + bias) v with q [B * Lq, nH * D], kv [B * Lk, 2 * nH * D] (k | v), bias [nH][Lq][Lk]
 *                       fp32 or NULL, any Lq / Lk, D = 32 | 64 (key blocks of 64, online softmax); the backward also returns
 *                       dbias = sum over the B problems of dS (fixed order). */
int vtx_window_gather(const void* map, void* dst, int B, int H, int W, int64_t ld, int c0, int nc, int win, int halo, int dtype,
                      void* stream);
int vtx_window_scatter(const void* src, void* map, int B, int H, int W, int64_t ld, int c0, int nc, int win, int halo, int dtype,
                       void* stream);
int vtx_table_bias(const float* table, const int64_t* pos, float* bias, int64_t cells, int nH, void* stream);
int vtx_table_bias_bwd(const float* full, const int* csr_order, const int* csr_offsets, float* dtable, int64_t cells, int nH, int ntab,
                       void* stream);
int vtx_xattn_fwd(const void* q, const void* kv, void* o, float* lse, const float* bias, int B, int Lq, int Lk, int nH, int D, int dtype,
                  void* stream);
size_t vtx_xattn_bwd_workspace(int B, int Lq, int nH);
int vtx_xattn_bwd(const void* q, const void* kv, const void* o, const void* dout, const float* lse, const float* bias, void* dq,
                  void* dkv, float* dbias, void* workspace, size_t ws_bytes, int B, int Lq, int Lk, int nH, int D, int dtype,
                  void* stream);

/* vtx_xattn_* with dropout of the attention probabilities (halo_transformer.py:101); arguments as vtx_attention_fwd_drop */
int vtx_xattn_fwd_drop(const void* q, const void* kv, void* o, float* lse, const float* bias, int B, int Lq, int Lk, int nH, int D,
                       int dtype, float drop_p, uint64_t seed, const uint8_t* keep, void* stream);
int vtx_xattn_bwd_drop(const void* q, const void* kv, const void* o, const void* dout, const float* lse, const float* bias, void* dq,
                       void* dkv, float* dbias, void* workspace, size_t ws_bytes, int B, int Lq, int Lk, int nH, int D, int dtype,
                       float drop_p, uint64_t seed, const uint8_t* keep, void* stream);

/* ---- Positional-encoding generator of Twins-SVT (csrc/twins_misc.hip; reference models/twins.py:25-37):
 * y = x + DepthwiseConv3x3(x) on channels-last features x, y [B, H, W, C] (C % 8 == 0, C <= 1024), w = the
 * Conv2d(C, C, 3, padding=1, bias=False, groups=C) weight [C, 1, 3, 3] fp32.  The reference permutes to NCHW, convolves and
 * permutes back (twins.py:32-35); here the taps are read in place.  adjoint != 0 mirrors the taps: vtx_dwconv3_fwd(dy, w, dx,
 * ..., 1, ...) is the input gradient.  _wgrad writes dw [C, 1, 3, 3] fp32 = sum over pixels of dy * shifted x (deterministic:
 * fixed-order partial sums in the workspace). */
int vtx_dwconv3_fwd(const void* x, const float* w, void* y, int B, int H, int W, int C, int adjoint, int dtype, void* stream);
size_t vtx_dwconv3_wgrad_workspace(int B, int H, int W, int C);
int vtx_dwconv3_wgrad(const void* x, const void* dy, float* dw, void* workspace, size_t ws_bytes, int B, int H, int W, int C,
                      int dtype, void* stream);

/* ---- Operand of the sub-sampling convolution of Twins-SVT's global attention (csrc/twins_misc.hip; reference
 * models/twins.py:69-71).  The reference reshapes its 4-D input as ``input.transpose(1, 2).reshape(B, C, H, W)``: the "image"
 * the Conv2d(C, C, r, stride r) sees is a fixed permutation of the feature map's elements (flat index f = c' H W + y W + x of
 * the (W, H, C)-ordered map), kept here as written.  x [B, H, W, C] -> out [B*(H/r)*(W/r), C*r*r], columns (c', py, px) --
 * the Conv2d weight's own memory layout [out][c'][py][px], so weight.view(out, C*r*r) is the GEMM operand and the weight
 * gradient comes out in the parameter's layout; out_t (nullable): a second, transposed copy [C*r*r, B*(H/r)*(W/r)] -- with the
 * transposed weight copy it turns the few-row / long-K convolution of the late stages (128 rows x K = 25 088) into a vtx_wgrad
 * call, i.e. a split-K launch that fills the chip; _bwd is the inverse scatter (accumulate != 0 adds into dx). */
int vtx_twins_subsample_fwd(const void* x, void* out, void* out_t, int B, int H, int W, int C, int r, int dtype, void* stream);
/* out [rows, C] (dtype) = x [rows, C] fp32 + bias [C] fp32 (nullable): fp32 split-K sums -> the compute dtype. */
int vtx_bias_cast(const float* x, const float* bias, void* out, int64_t rows, int C, int dtype, void* stream);
int vtx_twins_subsample_bwd(const void* dout, void* dx, int B, int H, int W, int C, int r, int accumulate, int dtype,
                            void* stream);

/* ---- Non-overlapping patch gather on token-major (NHWC) features: the im2col of Conv2d(C, C', p, stride = p)
 * (PVT patch embeddings of stages 2-4 and the spatial-reduction conv, pvt.py:26-29, 44-46, 112, 129-131).
 *   x [B, skip + H*W, C] (the first `skip` tokens of every image -- PVT's cls token -- are not part of the grid);
 *   out [B*(H/p)*(W/p), p*p*C] with column order (py, px, c); the conv weight is permuted to match by the host.
 * _bwd is the inverse scatter (a permutation); accumulate != 0 adds into dx (the gradient that also arrives
 * through the query path), rows of skipped tokens are left untouched. */
int vtx_patchify_fwd(const void* x, void* out, int B, int H, int W, int C, int p, int skip, int dtype, void* stream);
int vtx_patchify_bwd(const void* dout, void* dx, int B, int H, int W, int C, int p, int skip, int accumulate, int dtype,
                     void* stream);

/* ---- Position-embedding add with optional cls token (PVT PatchEmbedding, pvt.py:133-137):
 *   out[b, 0] = cls + pos[0] (has_cls), out[b, s + t] = x[b, t] + pos[s + t], s = has_cls.
 *   x [B, T, C] of dtype, cls [C] / pos [s + T, C] fp32; _bwd: dx = dout[:, s:], dcls = sum_b dout[b, 0],
 *   dpos = sum_b dout[b] (fp32, deterministic). */
int vtx_add_pos_fwd(const void* x, const float* cls, const float* pos, void* out, int B, int T, int C, int dtype,
                    void* stream);
int vtx_add_pos_bwd(const void* dout, void* dx, float* dcls, float* dpos, int B, int T, int C, int dtype, void* stream);

/* ---- Row-wise L2 normalisation y = x / max(||x||_2, eps) (F.normalize of the DINO head, models/vit.py:258) and its
 * backward dx = (dy - y sum(y o dy)) / max(||x||, eps); nrm [rows] fp32 is saved by the forward. */
int vtx_l2norm_fwd(const void* x, void* y, float* nrm, int64_t rows, int C, float eps, int dtype, void* stream);
int vtx_l2norm_bwd(const void* dy, const void* y, const float* nrm, void* dx, int64_t rows, int C, int dtype, void* stream);

/* ---- Device-side input pipeline (csrc/input.hip): per-sample mixup / cutmix
 * (reference mix_dataset.py:36-90) + Normalize + RandomErasing in its three colour modes (reference transforms.py:309-418;
 * factory.py:177-181 configures mode "pixel") of a device-resident batch in one pass.  The random plan is drawn on the
 * host in the reference's order (vtx.input_pipeline.plan_batch): device array of N records
 *   {int partner, mode (0 none | 1 mixup | 2 cutmix); float ratio; int x1, y1, x2, y2, nrect, top[4], left[4];
 *    short eh[4], ew[4]; int fmode (0 const | 1 rand | 2 pixel), foff[4]}
 *   (vtx_mix_plan_bytes() = 100 bytes each; at most vtx_mix_max_rects() rectangles);
 * fills: fp32 table of the host-drawn normal values (Tensor.normal_ with the reference's shapes), rectangle r of a sample
 *   reads [C] (rand) or [C][eh][ew] (pixel) floats at foff[r]; may be NULL when every fmode is 0.
 * x [N, C, H, W] uint8 (in_u8: scaled by 1/255 like ToTensor) or fp32, W % 4 == 0;
 * out: [N, C, H, W] fp32 (out_nhwc_bf16 == 0: the reference's model input) or [N, H, W, C] bf16 (C in {1, 3, 4}) -- the
 *   layout vtx_patch_gather_nhwc consumes, so the patch-embedding input is written once, in place of train.py:267's
 *   fp32 NCHW batch + the permute / patchify copies of swin_transformer.py:371, 15-22. */
size_t vtx_mix_plan_bytes(void);
int vtx_mix_max_rects(void);
int vtx_mix_normalize_erase(const void* x, int in_u8, const void* plan, const float* mean, const float* stdv,
                            const float* fills, void* out, int out_nhwc_bf16, int N, int C, int H, int W, void* stream);

/* ---- Fused optimizer tail (csrc/optim.hip): nn.utils.clip_grad_norm_ + torch.optim.AdamW.step of the reference's
 * train step (train.py:285-299) as two multi-tensor passes.  Tensors are given as HOST arrays of n device pointers
 * (fp32, any 4-byte alignment) and element counts; the addresses travel in kernel arguments (64 tensors per launch).
 *   vtx_grad_sqnorm: norm_out[0] = sum over all tensors of g^2, norm_out[1] = its square root (deterministic: fixed
 *                    summation order, no atomics); partial = sum_i ceil(numel_i / vtx_opt_chunk()) floats of workspace.
 *   vtx_adamw_step:  step t >= 1 of AdamW (decoupled decay first, bias-corrected, eps added to sqrt(v)/sqrt(1-b2^t)),
 *                    per-tensor lr / weight decay, on g * min(1, max_norm / (norm[1] + 1e-6)) when max_norm > 0
 *                    (norm = vtx_grad_sqnorm's device output); gradients are not modified. */
int vtx_opt_chunk(void);
int vtx_grad_sqnorm(int n, const float* const* g, const int64_t* numel, float* partial, float* norm_out, void* stream);
int vtx_adamw_step(int n, float* const* p, const float* const* g, float* const* m, float* const* v,
                   const int64_t* numel, const float* lr, const float* wd, const float* norm, float max_norm,
                   float beta1, float beta2, float eps, int t, void* stream);

/* Momentum (EMA) update of the DINO teacher / an EMA model: p_i = m * p_i + (1 - m) * g_i for n fp32 tensors given as
 * HOST arrays of device pointers (train_dino.py:258-263, train_util.py:70-76). */
int vtx_ema_update(int n, float* const* p, const float* const* g, const int64_t* numel, float m, void* stream);

/* ---- DINO loss (csrc/dino.hip; reference loss.py:122-152): forward value and gradient w.r.t. the student logits in
 * one sweep, plus the teacher column sums of update_center.
 *   student [n_crop*B, K], teacher [2*B, K] (dtype), center [K] fp32 (read only); K % 8 == 0;
 *   loss_rows [n_crop*B] fp32 with loss = sum(loss_rows) / ((2 n_crop - 2) B);
 *   dstudent [n_crop*B, K] (dtype) = gscale * d loss / d student;  batch_center [K] fp32 = sum of the teacher rows;
 *   workspace: vtx_dino_loss_workspace(B, K) bytes. */
size_t vtx_dino_loss_workspace(int B, int K);
int vtx_dino_loss(const void* student, const void* teacher, const float* center, void* workspace, size_t ws_bytes,
                  float* loss_rows, void* dstudent, float* batch_center, int n_crop, int B, int K, float student_temp,
                  float teacher_temp, float gscale, int dtype, void* stream);

/* ---- MixLoss of the supervised train step (csrc/dino.hip; reference loss.py:53-86, train.py:274-276): label-smoothed KL
 * between log_softmax(logits) and r * smooth(label1) + (1 - r) * smooth(label2), reduction "mean"; value and gradient
 * in one sweep.  logits / dlogits [B, K] (dtype), labels [B] int64, ratio [B] fp32; loss = sum(loss_rows) / B. */
int vtx_mix_loss(const void* logits, const int64_t* label1, const int64_t* label2, const float* ratio, void* dlogits,
                 float* loss_rows, int B, int K, float eps, float gscale, int dtype, void* stream);

/* ---- Multi-tensor weight cast (csrc/cast.hip): all fp32 Linear / Conv weights of a model -> bf16, plain [out][in]
 * and transposed [in][out], in one launch per forward.  This is the per-call weight cast of the reference's bf16
 * autocast (torch.cuda.amp.autocast around model(input), train.py:273-274) done once for the whole model.
 * desc: device array of nmat records {const float* src; int64 off; int rows, cols, tile0, tiles_c}
 * (vtx_cast_desc_bytes() each); ntiles = sum ceil(rows/64)*ceil(cols/64); matrix i occupies
 * [off_i, off_i + rows_i*cols_i) of both flat bf16 outputs. */
size_t vtx_cast_desc_bytes(void);
int vtx_cast_weights(const void* desc, int nmat, int ntiles, void* dst, void* dst_t, void* stream);

/* ---- Patch gather: NCHW fp32 image -> patch matrix [B*(H/p)*(W/p), Kp] of dtype, columns >= 3*p*p zero.
 *   order 0: column (py, px, c)  -- Swin: permute(0,2,3,1) + patchify(4) (models/swin_transformer.py:15-22,
 *            208-213, 371); the Linear(48, C) then runs as vtx_gemm on the padded K = Kp
 *   order 1: column (c, py, px)  -- ViT: Conv2d(3, C, p, stride=p) as an im2col GEMM (models/vit.py:73, 76) */
int vtx_patch_gather(const float* x, void* out, int B, int Cin, int H, int W, int p, int Kp, int order, int dtype,
                     void* stream);
/* Same patch matrix from a bf16 NHWC image [B, H, W, Cin] (vtx_mix_normalize_erase's out_nhwc_bf16 output), W % 8 == 0. */
int vtx_patch_gather_nhwc(const void* x, void* out, int B, int Cin, int H, int W, int p, int Kp, int order, int dtype,
                          void* stream);
/* ---- Token mean over Tn tokens: AdaptiveAvgPool2d(1)+Flatten of the Swin classifier
 * (models/swin_transformer.py:281, 376-377) on NHWC features. */
int vtx_token_mean_fwd(const void* x, void* y, int B, int Tn, int C, int dtype, void* stream);
int vtx_token_mean_bwd(const void* dy, void* dx, int B, int Tn, int C, int dtype, void* stream);
/* ---- ViT token assembly (models/vit.py:140-143): out[b,0] = cls + pos[0]; out[b,1+t] = patches[b,t] + pos[1+t].
 * L = n_patch + 1; cls [C], pos [L, C] fp32.  Backward: dpatches, dcls [C], dpos [L, C] (fp32, batch sums). */
int vtx_vit_assemble_fwd(const void* patches, const float* cls, const float* pos, void* out, int B, int L, int C,
                         int dtype, void* stream);
int vtx_vit_assemble_bwd(const void* dx, void* dpatches, float* dcls, float* dpos, int B, int L, int C, int dtype,
                         void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VTX_H_ */
