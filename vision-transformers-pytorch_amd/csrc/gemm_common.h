// Shared pieces of the GEMM family (gemm.hip: register-staged kernels for every mode / dtype;
// gemm_glds.hip: LDS-DMA kernel for the bf16 forward-layout GEMMs).
#pragma once
#include "vtx_common.h"

// phase time stamps for tools/probe/gemm_trace.hip (compiled out of the library)
#ifndef VTX_TRACE
#define VTX_TRACE(id)
#endif
// phase ablation of the LDS-DMA GEMM for IN-MODEL timing (tools/probe/build_ablate.sh builds separate libraries, selected
// through VTX_LIBVTX; results are garbage, durations are what is measured): 1 no main loop | 4 no epilogue stores |
// 8 no epilogue operand loads | 16 SiLU applied to the A fragments after the LDS read (cost probe of "fc2 reads z").  0 in the library.
#ifndef GLDS_ABLATE
#define GLDS_ABLATE 0
#endif

struct GemmArgs {
  const void* A; const void* B; void* C;
  int M, N, K;
  int64_t lda, ldb, ldc;
  const float* bias;        // [N] fp32 or null
  const void* resid;        // T [M, ldc] or null: C = resid + rowscale * (acc + bias)
  const float* rowscale;    // per-sample DropPath scale on OUTPUT rows (index row / rows_per_scale) or null
  int rows_per_scale;
  void* aux_out;            // T [M, ldc] or null: pre-activation z when act == 1
  const void* aux_in;       // T [M, ldc]: z when act == 2
  int act;                  // 0 none | 1 (3) aux_out = z = acc + bias, C = silu (gelu)(z) | 2 (4) C = acc * silu' (gelu')(aux_in)
  const float* kscale;      // per-sample scale along the CONTRACTION index of a transposed A (wgrad through DropPath)
  int k_per_scale;
  float kscale_const;       // > 0: every kscale value is 0 or this constant (rows are masked, the constant scales the accumulators)
  int kchunk;               // contraction length per grid.z slice (multiple of the LDS k-tile)
  float* ksum_out;          // TA only: [grid.z][M] fp32 = sum over the contraction of opA (bias gradient), or null
  // Stochastic-depth compaction (round 3; LDS-DMA kernels with the wave-private epilogue only): the M logical rows are the
  // samples of `perm` in order -- logical row r is row perm[r / map_T] * map_T + r % map_T of EVERY row-indexed operand
  // (A, C, resid, aux_in, aux_out; rowscale is indexed by perm[r / map_T]).  The kept samples come first: rows [0, Mk) are
  // computed, rows [Mk, M) belong to dropped samples and are copy-only (C = resid; only launched when there is a resid).
  const int* perm;          // [samples] int32 or null (no map: logical row == row, Mk == M)
  int map_T;                // rows per sample
  int Mk;
  unsigned map_magic;       // ceil(2^32 / map_T): r / map_T == __umulhi(r, map_magic) for r * map_T < 2^32 (no runtime division)
};
inline void gemm_args_nomap(GemmArgs& a) { a.perm = nullptr; a.map_T = 1; a.Mk = a.M; a.map_magic = 0; }
inline unsigned vtx_div_magic(int T) { return (unsigned)((0x100000000ull + (unsigned)T - 1) / (unsigned)T); }


// Global operands of the epilogue, requested ahead of it (EpiOperands::load) -- by the LDS-DMA kernel before its main
// loop, so they land under the first k-tile's wait (vmcnt retires in order: they are older than every DMA piece).
// Issued inside the store loop they cannot be hoisted above the previous iteration's stores (possible aliasing), and
// with 8-12 waves per CU each iteration then exposed a full L2 / HBM latency -- measured on the stage-2 fc1 forward:
// 81 us of a 139 us launch with the main loop AND the stores switched off.
//   bias: per lane the WN columns of its accumulator tiles (added while staging);  z (act') or the residual: one
//   16-byte vector per store iteration, both passes (never both on the hot path: a residual next to act' is loaded late)
// NWN = waves along N (2: 2 x 2 waves, 4: 2 x 4 waves); the workgroup has 128 * NWN threads
template <typename T, int BM, int BN, int NWN = 2> struct EpiOperands {
  static constexpr int NT = 128 * NWN;
  static constexpr int WN = BN / (16 * NWN);
  static constexpr int VROW = BN / 8;                       // 8-element vectors per staged row
  static constexpr int NVEC = (BM / 2) * VROW;              // vectors per pass
  static constexpr int NIT = (NVEC + NT - 1) / NT;          // iterations per pass
  float bcol[WN];
  float rsc[2 * NIT];
  Vec8<T> ein[2 * NIT];

  // (row, col) of the 8-vector this thread stores in iteration it of a pass; false when it has none
  static __device__ __forceinline__ bool where(const GemmArgs& p, int m0, int n0, int pass, int it, int& row, int& col) {
    const int v = threadIdx.x + NT * it;
    const int lr = v / VROW, cv = v - lr * VROW;
    const int w2 = lr / (BM / 4), rem = lr - w2 * (BM / 4);
    row = m0 + w2 * (BM / 2) + pass * (BM / 4) + rem;
    col = n0 + cv * 8;
    return (NVEC % NT == 0 || v < NVEC) && row < p.M && col < p.N;
  }

  __device__ __forceinline__ void load(const GemmArgs& p, int m0, int n0, int wn, int c_) {
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int col = n0 + wn * (BN / NWN) + j * 16 + c_;
      bcol[j] = (p.bias && col < p.N) ? p.bias[col] : 0.f;
    }
    const bool act_bwd = p.act == 2 || p.act == 4;
    const T* __restrict__ esrc = act_bwd ? (const T*)p.aux_in : (const T*)p.resid;
#pragma unroll
    for (int q = 0; q < 2 * NIT; ++q) {
      int row, col;
      const bool ok = where(p, m0, n0, q / NIT, q % NIT, row, col);
      ein[q] = vec8_zero<T>();
      rsc[q] = 1.f;
      if (ok && !(GLDS_ABLATE & 8)) {
        if (esrc) ein[q] = load8<T>(esrc + (int64_t)row * p.ldc + col);
        if (p.rowscale) rsc[q] = p.rowscale[row / p.rows_per_scale];
      }
    }
  }
};

// acc[i][j][r] = C[m0 + wm*(BM/2) + 16 i + 4 g + r][n0 + wn*(BN/NWN) + 16 j + c]; lds_raw: >= (BM/2)*(BN+4)*4 bytes,
// no longer read by anybody when this is called (callers barrier after their last operand read).
// The tile goes through LDS in two passes of BM/2 rows so that every lane stores 16 contiguous bytes.
template <typename T, typename TO, int BM, int BN, int NWN = 2>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, f32x4 (&acc)[BM / 32][BN / (16 * NWN)], unsigned char* lds_raw,
                                              int m0, int n0, int tz, int wm, int wn, int c_, int g_,
                                              const EpiOperands<T, BM, BN, NWN>& eo) {
  constexpr int WM = BM / 32, WN = BN / (16 * NWN);
  constexpr int NT = 128 * NWN;
  constexpr int CSTR = BN + 4;                // fp32 C-staging row stride (floats)
  using EO = EpiOperands<T, BM, BN, NWN>;
  constexpr int VROW = EO::VROW, NIT = EO::NIT;
  TO* __restrict__ Cout = (TO*)p.C + (int64_t)tz * p.M * p.ldc;
  const T* __restrict__ resid = (const T*)p.resid;
  T* __restrict__ aux_out = (T*)p.aux_out;
  float* cbuf = reinterpret_cast<float*>(lds_raw);
  const bool act_fwd = p.act == 1 || p.act == 3, act_bwd = p.act == 2 || p.act == 4;

#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
    for (int ii = 0; ii < WM / 2; ++ii) {
      const int i = pass * (WM / 2) + ii;
#pragma unroll
      for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          cbuf[(wm * (BM / 4) + ii * 16 + g_ * 4 + r) * CSTR + wn * (BN / NWN) + j * 16 + c_] = acc[i][j][r] + eo.bcol[j];
    }
    __syncthreads();
    VTX_TRACE(3 + 2 * pass);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int q = pass * NIT + it;
      int row, col;
      if (!EO::where(p, m0, n0, pass, it, row, col)) continue;
      const int64_t off = (int64_t)row * p.ldc + col;
      const int v = threadIdx.x + NT * it;
      const int lr = v / VROW, cv = v - lr * VROW;
      const float* cp = cbuf + lr * CSTR + cv * 8;
      f32x4 lo = *reinterpret_cast<const f32x4*>(cp), hi = *reinterpret_cast<const f32x4*>(cp + 4);
      float val[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      if (act_fwd) {
        Vec8<T> z;
#pragma unroll
        for (int e = 0; e < 8; ++e) z.set(e, val[e]);  // activation of the ROUNDED pre-activation (what the backward sees)
        if (p.act == 1) {
#pragma unroll
          for (int e = 0; e < 8; ++e) val[e] = silu_f(z.get(e));
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) val[e] = gelu_f(z.get(e));
        }
        if (aux_out && !(GLDS_ABLATE & 4)) store8<T>(aux_out + off, z);
      } else if (act_bwd) {
        if (p.act == 2) {
#pragma unroll
          for (int e = 0; e < 8; ++e) val[e] *= dsilu_f(eo.ein[q].get(e));
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) val[e] *= dgelu_f(eo.ein[q].get(e));
        }
      }
      Vec8<T> rv = eo.ein[q];
      if (act_bwd) rv = resid ? load8<T>(resid + off) : vec8_zero<T>();
      Vec8<TO> o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o.set(e, val[e] * eo.rsc[q] + rv.get(e));
      if (!(GLDS_ABLATE & 4) || o.get(0) == 12345.678f) store8<TO>(Cout + off, o);
    }
    VTX_TRACE(4 + 2 * pass);
    if (pass == 0) __syncthreads();
  }
}

// LDS-DMA (global_load_lds) bf16 kernel for C = epi(A W^T) with K % 64 == 0 (gemm_glds.hip)
int gemm_glds_launch(const GemmArgs& a, hipStream_t st);
bool gemm_glds_enabled();
bool gemm_glds_ok(int N, int K);
// gemm_skinny.hip: weight-resident streaming kernel for K = 64 / 96 / 128 over >= 32 768 rows (bf16, A . W^T)
bool gemm_skinny_ok(const GemmArgs& a);
int gemm_skinny_launch(const GemmArgs& a, hipStream_t st);
// gemm_astat.hip: A-stationary kernel for 128 <= K <= 384 (one workgroup per CU keeps its 128-row strip of A in LDS for all column tiles)
bool gemm_astat_ok(const GemmArgs& a);
int gemm_astat_launch(const GemmArgs& a, hipStream_t st);
// gemm_pp.hip: BM x 192 tiles, one workgroup per CU, two wave groups half a k-step apart (N % 192 == 0, K % 64 == 0)
bool gemm_pp_ok(const GemmArgs& a);
int gemm_pp_launch(const GemmArgs& a, hipStream_t st);
// mlp_fused.hip: the whole MLP of a C = 64 / 96 layer in one launch each way (both weights resident in LDS)
bool mlp_fused_ok(int dtype, int64_t M, int C, int ff);
int mlp_fused_fwd(const void* ln2, const void* w1, const float* b1, const void* w2, const float* b2, const void* resid,
                  const float* rowscale, int rows_per_scale, void* y, void* z, void* h, int64_t M, int C, int ff, hipStream_t st);
int mlp_fused_bwd(const void* ln2, const void* dy, const void* w1, const float* b1, const void* w2, const float* rowscale,
                  int rows_per_scale, void* h, void* dz, void* dln2, int64_t M, int C, int ff, hipStream_t st);
// the same with the LayerNorm backward of the layer's norm_ff in its epilogue (option LN_FOLD bit 0): dx1 = dy + LN'(dln2) out, dln2 never stored;
// part: the [part_rows][2 C] workspace of vtx_layernorm_bwd's deferred dgamma / dbeta partial rows (part_rows = vtx_layernorm_bwd_blocks)
bool mlp_fused_ln_ok(int dtype, int64_t M, int C, int ff);
// LayerNorm FORWARD folds (option LN_FOLD bits 2, 3): norm_ff on the row operands of the fused-MLP forward; a narrow norm on the row operands
// of the weight-resident streaming GEMM that consumes it (gemm_skinny.hip)
bool mlp_fused_lnf_ok(int dtype, int64_t M, int C, int ff);
int mlp_fused_fwd_ln(const void* x1, const float* gamma, const float* beta, float eps, void* ln2, float* mean, float* rstd, const void* w1,
                     const float* b1, const void* w2, const float* b2, const float* rowscale, int rows_per_scale, void* y, int64_t M, int C,
                     int ff, hipStream_t st);
bool ln_gemm_ok(int dtype, int64_t M, int C, int N);
int ln_gemm_launch(const void* x, const float* gamma, const float* beta, float eps, void* ln_out, float* mean, float* rstd, const void* w,
                   const float* bias, void* y, int64_t M, int C, int N, hipStream_t st);
int mlp_fused_bwd_ln(const void* ln2, const void* dy, const void* w1, const float* b1, const void* w2, const float* rowscale,
                     int rows_per_scale, void* h, void* dz, const void* x1, const float* mean, const float* rstd, const float* gamma,
                     void* dx1, float* part, int part_rows, int64_t M, int C, int ff, hipStream_t st);
// dx = dres + LN'(dy . W) in one launch (gemm_skinny.hip dgrad_ln_kernel; option LN_FOLD bit 1): wt = the transposed bf16 weight copy [C][K]
bool dgrad_ln_ok(int dtype, int64_t M, int C, int K);
bool dgrad_ln_shape_ok(int C, int K);
int dgrad_ln_launch(const void* dy, const void* wt, const void* x, const float* mean, const float* rstd, const float* gamma, const void* dres,
                    void* dx, float* part, int part_rows, int64_t M, int C, int K, hipStream_t st);
extern "C" int vtx_layernorm_bwd_blocks(int64_t rows, int C);
extern "C" size_t vtx_layernorm_bwd_workspace(int64_t rows, int C);
// mapped (compacted) launch: bf16, mode 0, N % 128 == 0, K % 64 == 0, wave-private epilogue -- else VTX_ERR_SHAPE
int gemm_glds_launch_mapped(const GemmArgs& a, hipStream_t st);

// LDS-DMA + transpose-read weight-gradient kernel (gemm_wgrad_glds.hip): bf16, N and Kin multiples of 8 and >= 64,
// rowscale values restricted to {0, scale_const}.  Grouped: up to wgrad_glds_max_problems() weight gradients over the
// same tokens in one launch, split-K partials left as fp32 slabs (the caller reduces them in one launch).
struct WgradProbHost {
  const void* dy; const void* x;
  float* slab; float* out; float* ksum_part; float* ksum_out;
  const float* rowscale;
  int64_t ld_dy, ld_x;
  int N, Kin;
  int live_only;      // 1: rowscale only says which samples' rows exist (dropped ones are skipped); dy already carries the scale
  const int* perm;    // stochastic-depth compaction: contract over the Mtok tokens of the kept samples only, in perm order
  int Mtok;
  float scale;        // accumulator scale of a mapped problem
};
bool wgrad_glds_ok(int dtype, int N, int Kin, const float* rowscale, float scale_const);
int wgrad_glds_resident();
int wgrad_glds_max_problems();
int wgrad_glds_tiles(int N, int Kin);
int wgrad_wide_tiles(int nprob, const int* N, const int* Kin, int* J = nullptr);   // 128 x 64 J tiles of the group (J = 3 .. 6), or 0: the group stays on 128 x 128 tiles
int wgrad_wide_tiles_any(int nprob, const int* N, const int* Kin, int* J = nullptr);   // the same whatever option WGRAD_WIDE says (workspace sizing)
int wgrad_glds_slices(int64_t mtok, int ntiles, bool wide = false);
int wgrad_glds_group_launch(int nprob, const WgradProbHost* hp, int64_t mtok, int rows_per_scale, float scale_const,
                            int nz, int kchunk, hipStream_t st, int wide = 0);
