// Weight-resident streaming GEMM for gfx950 (bf16): C[M,N] = epilogue(A[M,K] . W[N,K]^T) with a SHORT contraction
// (K = 64 / 96 / 128: Swin-S stage 1, PVT stages 1-2) over MANY rows (M >= 32 768).
//
// These launches are pure HBM streams (Swin stage 1, M = 401 408, K = 96: qkv forward reads 77 MB and writes 231 MB; fc1
// forward writes its 308-MB output twice), yet the tiled kernels ran them at 0.4-0.6 of the stream rate: a 128 x 128 (or 64 x
// 128) tile has a 1-3 k-tile main loop, i.e. it is all prologue and epilogue, the A panel is fetched once per column tile, and
// the register-staged kernel that takes N = 96 / 288 pays ds_write for every operand byte.  Here
//   * the WHOLE weight (N x K bf16, <= 139 KB) is copied into LDS once per persistent workgroup (one per CU, 4 waves);
//   * a wave streams 32-row blocks of A straight from global memory into MFMA operand registers (16 B per lane and k-step,
//     the next block's fragments requested before the current block is multiplied) -- no LDS traffic for A at all;
//   * the product is taken TRANSPOSED, D[n][m] with W rows as the MFMA A operand in the row order n(8 g + e) <-> operand row
//     16 (e >> 2) + 4 g + (e & 3) (the permutation of attention_win.hip's wa_store_t): after two 16 x 16 tiles a lane holds 8
//     CONSECUTIVE output columns of one row -> the epilogue works on 16-byte vectors in the layout of its operands (residual,
//     z) with no staging pass, and every output byte is written once;
//   * same element-wise epilogue expressions as gemm_glds.hip (bias, SiLU / GELU + z, silu' / gelu' x z, DropPath scale,
//     residual), same k order inside the MFMA: bit-identical to the tiled kernels.
// Measured (tools/probe/skinny_gemm.hip, M = 401 408, K = 96): N = 288 74 us (tiled 117), N = 384 with two outputs 168 us
// (217), N = 96 26 us; floors at 6.3 TB/s: 49 / 110 / 24.5 us.
#include "gemm_common.h"
#include "ln_fold.h"
#include "options.h"

constexpr int SK_ROWS = 32;        // rows of A per wave iteration (two 16-row MFMA tiles)
// waves per persistent workgroup (one workgroup per CU): option SKINNY_WAVES (4 | 8 | 16)

// K = 32 KS; ACT = GemmArgs::act, VEC = the epilogue reads a vector per output vector (z for act', else the residual) --
// both compile-time: the per-pair loop of a plain / activation forward carries no loads, no selects, no dead operands
// LNF (round 6, option LN_FOLD bit 3): A holds the RAW rows x of a LayerNorm whose output is this product's row operand (K = C: the qkv
// projection behind norm_attn, models/swin_transformer.py:128,194); the rows are normalised in registers (ln_fold.h LnFwdFold) and the
// normalised rows / mean / rstd stored on the side (workgroups of column chunk 0 only) -- the stand-alone LayerNorm launch and its
// read of x disappear; outputs bit-identical.
struct SkLnArgs {
  const float* gamma;
  const float* beta;
  float eps;
  bf16* ln_out;         // [M][K]
  float* mean;
  float* rstd;
};

template <int KS, int ACT, bool VEC, int SK_WAVES, bool LNF = false>
__global__ __launch_bounds__(64 * SK_WAVES) void gemm_skinny_kernel(GemmArgs p, int nchunk, SkLnArgs q) {
  constexpr int K = 32 * KS, WSTR = K + 8;             // LDS row stride of W (elements): 16 B of padding per row
  extern __shared__ __attribute__((aligned(16))) unsigned char sk_smem[];
  bf16* ws = reinterpret_cast<bf16*>(sk_smem);
  // A weight too large for LDS is split into `nchunk` (1, 2 or 4) column chunks: workgroup b keeps chunk b % nchunk and streams
  // every nchunk-th share of the row blocks (A is then read nchunk times -- it is the small operand of these launches)
  const int chunk = blockIdx.x % nchunk, wgc = blockIdx.x / nchunk, nwgc = gridDim.x / nchunk;
  const int N = p.N / nchunk, M = p.M;                  // columns of this workgroup
  const int ncol0 = chunk * N;
  const bf16* __restrict__ A = (const bf16*)p.A;
  const bf16* __restrict__ W = (const bf16*)p.B + (int64_t)ncol0 * p.ldb;
  float* bs = reinterpret_cast<float*>(sk_smem + (size_t)N * WSTR * 2);
  for (int i = threadIdx.x; i < N * (K / 8); i += 64 * SK_WAVES) {
    const int n = i / (K / 8), q = i - n * (K / 8);
    *reinterpret_cast<bf16x8*>(ws + n * WSTR + q * 8) = *reinterpret_cast<const bf16x8*>(W + (int64_t)n * p.ldb + q * 8);
  }
  for (int i = threadIdx.x; i < N; i += 64 * SK_WAVES) bs[i] = p.bias ? p.bias[ncol0 + i] : 0.f;
  float* lgs = bs + N;                                  // LNF: gamma | beta [2][K]
  if constexpr (LNF)
    for (int i = threadIdx.x; i < K; i += 64 * SK_WAVES) { lgs[i] = q.gamma[i]; lgs[K + i] = q.beta[i]; }
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const int nrb = (M + SK_ROWS - 1) / SK_ROWS;
  const int stride = nwgc * SK_WAVES;
  bf16* __restrict__ Cout = (bf16*)p.C + ncol0;
  bf16* __restrict__ aux_out = p.aux_out ? (bf16*)p.aux_out + ncol0 : nullptr;
  const bf16* __restrict__ resid = p.resid ? (const bf16*)p.resid + ncol0 : nullptr;
  const bf16* __restrict__ aux_in = p.aux_in ? (const bf16*)p.aux_in + ncol0 : nullptr;
  constexpr int act = ACT;
  constexpr bool act_fwd = act == 1 || act == 3, act_bwd = act == 2 || act == 4;
  const int npairs = N >> 5;

  bf16x8 an[2][KS];
  auto load_a = [&](int rb) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int row = min(rb * SK_ROWS + mt * 16 + c, M - 1);   // (rows past M are never stored)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) an[mt][ks] = *reinterpret_cast<const bf16x8*>(A + (int64_t)row * p.lda + ks * 32 + g * 8);
    }
  };
  int rb = wgc * SK_WAVES + wave;
  if (rb < nrb) load_a(rb);
  for (; rb < nrb; rb += stride) {
    bf16x8 a[2][KS];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) a[mt][ks] = an[mt][ks];
    if (rb + stride < nrb) load_a(rb + stride);
    int row[2];
    float rsc[2];
    bool ok[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      row[mt] = rb * SK_ROWS + mt * 16 + c;
      ok[mt] = row[mt] < M;
      rsc[mt] = (ok[mt] && p.rowscale) ? p.rowscale[row[mt] / p.rows_per_scale] : 1.f;
    }
    if constexpr (LNF) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        bf16x8 yv[KS];
        float mu, rs;
        LnFwdFold<KS>::row(a[mt], lgs, lgs + K, q.eps, g, yv, mu, rs);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) a[mt][ks] = yv[ks];
        if (chunk == 0 && ok[mt]) {
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) *reinterpret_cast<bf16x8*>(q.ln_out + (int64_t)row[mt] * K + ks * 32 + g * 8) = yv[ks];
          if (g == 0) { q.mean[row[mt]] = mu; q.rstd[row[mt]] = rs; }
        }
      }
    }
    // epilogue vectors of the first column pair (z for act', else the residual); the next pair's are requested one pair ahead
    const bf16* __restrict__ vsrc = act_bwd ? aux_in : resid;
    Vec8<bf16> ev[2], evn[2];
    if constexpr (VEC) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) evn[mt] = load8<bf16>(vsrc + (int64_t)min(row[mt], M - 1) * p.ldc + 8 * g);
    }
#pragma unroll 3
    for (int np = 0; np < npairs; ++np) {
      if constexpr (VEC) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) ev[mt] = evn[mt];
        const int npn = min(np + 1, npairs - 1);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) evn[mt] = load8<bf16>(vsrc + (int64_t)min(row[mt], M - 1) * p.ldc + npn * 32 + 8 * g);
      } else {
        ev[0] = vec8_zero<bf16>(); ev[1] = ev[0];
      }
      f32x4 acc[2][2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[mt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = np * 32 + 8 * (c >> 2) + 4 * j + (c & 3);       // the W row this lane supplies as MFMA A-operand row c
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          Vec8<bf16> wf;
          wf.v = *reinterpret_cast<const bf16x8*>(ws + n * WSTR + ks * 32 + g * 8);
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) {
            Vec8<bf16> af;
            af.v = a[mt][ks];
            mma16(wf, af, acc[mt][j]);
          }
        }
      }
      // acc[mt][j][r] = (A . W^T)[row = 32 rb + 16 mt + c][col = 32 np + 8 g + 4 j + r]
      const int col = np * 32 + 8 * g;
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(bs + col), b1 = *reinterpret_cast<const f32x4*>(bs + col + 4);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        if (!ok[mt]) continue;
        const int64_t off = (int64_t)row[mt] * p.ldc + col;
        float val[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) { val[r] = acc[mt][0][r] + b0[r]; val[4 + r] = acc[mt][1][r] + b1[r]; }
        if (act_fwd) {
          Vec8<bf16> z;
#pragma unroll
          for (int e = 0; e < 8; ++e) z.set(e, val[e]);
          if (act == 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) val[e] = silu_f(z.get(e));
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) val[e] = gelu_f(z.get(e));
          }
          if (aux_out) store8<bf16>(aux_out + off, z);
        } else if (act_bwd) {
          if (act == 2) {
#pragma unroll
            for (int e = 0; e < 8; ++e) val[e] *= dsilu_f(ev[mt].get(e));
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) val[e] *= dgelu_f(ev[mt].get(e));
          }
        }
        Vec8<bf16> rv = ev[mt];
        if (act_bwd) rv = resid ? load8<bf16>(resid + off) : vec8_zero<bf16>();
        Vec8<bf16> o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o.set(e, val[e] * rsc[mt] + rv.get(e));
        store8<bf16>(Cout + off, o);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// DGRAD + LayerNorm BACKWARD in one launch (round 6, option LN_FOLD bit 1): dx = dres + LN'(dy . W) for a narrow layer -- the qkv input
// gradient of Swin-S stage 1 (dln1 = dqkv . Wqkv: K = 3 C = 288 -> N = C = 96 over 401 408 rows) followed by the norm_attn backward
// (models/swin_transformer.py:194: out = input + drop_path(attn(norm_attn(input)))).  Stand-alone these are a register-staged GEMM that
// writes dln1 (83 us) and ln_bwd_kernel reading it back with x and the residual-stream gradient (76 us).  Here the weight (the
// transposed bf16 copy [C][K], <= 57 KB) is LDS-resident as above, a wave streams 32-row blocks of dqkv into operand registers
// (K / 32 k-steps), all C output columns of the block end up in accumulators in the row-operand layout, and ln_fold.h runs the
// LayerNorm backward on them: dln1 never exists in memory.  Same k order as the tiled kernels, bf16 rounding of dln1 kept: dx
// bit-identical to the two launches; dgamma / dbeta to fp32 summation order.
struct DgradLnArgs {
  const bf16* x;        // [M][C] the norm's input
  const float* mean;    // [M]
  const float* rstd;    // [M]
  const float* gamma;   // [C]
  const bf16* dres;     // [M][C] gradient of the residual stream that bypasses the norm
  bf16* dx;             // [M][C]
  float* part;          // [part_rows][2 C]
  int part_rows;
};

template <int KS, int NPR, int SK_WAVES>
__global__ __launch_bounds__(64 * SK_WAVES) void dgrad_ln_kernel(GemmArgs p, DgradLnArgs q) {
  constexpr int K = 32 * KS, C = 32 * NPR, WSTR = K + 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char sk_smem[];
  bf16* ws = reinterpret_cast<bf16*>(sk_smem);
  const int M = p.M;
  const bf16* __restrict__ A = (const bf16*)p.A;
  const bf16* __restrict__ W = (const bf16*)p.B;
  for (int i = threadIdx.x; i < C * (K / 8); i += 64 * SK_WAVES) {
    const int n = i / (K / 8), kq = i - n * (K / 8);
    *reinterpret_cast<bf16x8*>(ws + n * WSTR + kq * 8) = *reinterpret_cast<const bf16x8*>(W + (int64_t)n * p.ldb + kq * 8);
  }
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const int nrb = (M + SK_ROWS - 1) / SK_ROWS;
  const int stride = gridDim.x * SK_WAVES;
  LnFold<NPR> lnf;
  lnf.init(q.gamma, g);

  bf16x8 an[2][KS];
  auto load_a = [&](int rb) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int row = min(rb * SK_ROWS + mt * 16 + c, M - 1);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) an[mt][ks] = *reinterpret_cast<const bf16x8*>(A + (int64_t)row * p.lda + ks * 32 + g * 8);
    }
  };
  int rb = blockIdx.x * SK_WAVES + wave;
  if (rb < nrb) load_a(rb);
  for (; rb < nrb; rb += stride) {
    bf16x8 a[2][KS];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) a[mt][ks] = an[mt][ks];
    if (rb + stride < nrb) load_a(rb + stride);
    int row[2];
    bool ok[2];
    float mu[2], rs[2];
    bf16x8 xr[2][NPR], dr[2][NPR];                        // the LayerNorm's row operands: requested in front of the products
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      row[mt] = rb * SK_ROWS + mt * 16 + c;
      ok[mt] = row[mt] < M;
      const int rc = min(row[mt], M - 1);
      mu[mt] = q.mean[rc];
      rs[mt] = q.rstd[rc];
#pragma unroll
      for (int tp = 0; tp < NPR; ++tp) {
        xr[mt][tp] = *reinterpret_cast<const bf16x8*>(q.x + (int64_t)rc * C + tp * 32 + g * 8);
        dr[mt][tp] = *reinterpret_cast<const bf16x8*>(q.dres + (int64_t)rc * C + tp * 32 + g * 8);
      }
    }
    Vec8<bf16> dl[2][NPR];
#pragma unroll
    for (int np = 0; np < NPR; ++np) {
      f32x4 acc[2][2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[mt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = np * 32 + 8 * (c >> 2) + 4 * j + (c & 3);       // the W row this lane supplies as MFMA A-operand row c
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          Vec8<bf16> wf;
          wf.v = *reinterpret_cast<const bf16x8*>(ws + n * WSTR + ks * 32 + g * 8);
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) {
            Vec8<bf16> af;
            af.v = a[mt][ks];
            mma16(wf, af, acc[mt][j]);
          }
        }
      }
      // acc[mt][j][r] = (A . W^T)[row = 32 rb + 16 mt + c][col = 32 np + 8 g + 4 j + r]; the plain epilogue of the tiled kernels
      // (no bias, scale 1, no residual: `v * 1 + 0`) and the bf16 rounding of the dln store
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { dl[mt][np].set(r, (acc[mt][0][r] + 0.f) * 1.f + 0.f); dl[mt][np].set(4 + r, (acc[mt][1][r] + 0.f) * 1.f + 0.f); }
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) lnf.row(dl[mt], xr[mt], dr[mt], mu[mt], rs[mt], ok[mt], q.dx + (int64_t)row[mt] * C + 8 * g);
  }
  lnf.template finish<SK_WAVES>(reinterpret_cast<float*>(sk_smem), q.part, q.part_rows, wave, c, g);
}

// dx = dres + LN'(dy . W): dy [M][K] (ld = K), wt = the transposed bf16 weight copy [C][K]
bool dgrad_ln_ok(int dtype, int64_t M, int C, int K) {
  if ((vtx_opt(VTX_OPT_LN_FOLD) & 2) == 0 || dtype != VTX_BF16) return false;
  if (C != 64 && C != 96 && C != 128) return false;
  if (!dgrad_ln_shape_ok(C, K) || M < 32768 || M > 0x7fffffff) return false;
  return vtx_layernorm_bwd_blocks(M, C) >= vtx_cu_count_cached() && (size_t)C * (K + 8) * 2 <= 150 * 1024;
}

template <int KS, int NPR> static int dgrad_ln_launch_k(const GemmArgs& a, const DgradLnArgs& q, hipStream_t st) {
  const size_t smem = (size_t)32 * NPR * (32 * KS + 8) * 2;
  auto kern = dgrad_ln_kernel<KS, NPR, 4>;
  if (smem > 64 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return VTX_ERR_LAUNCH;
  hipLaunchKernelGGL(kern, dim3(vtx_cu_count_cached()), dim3(256), smem, st, a, q);
  return vtx_check_launch();
}

int dgrad_ln_launch(const void* dy, const void* wt, const void* x, const float* mean, const float* rstd, const float* gamma, const void* dres,
                    void* dx, float* part, int part_rows, int64_t M, int C, int K, hipStream_t st) {
  if (!dy || !wt || !x || !mean || !rstd || !gamma || !dres || !dx || !part) return VTX_ERR_NULL;
  if (part_rows < vtx_cu_count_cached()) return VTX_ERR_WORKSPACE;
  GemmArgs a = {};
  a.A = dy; a.B = wt; a.M = (int)M; a.N = C; a.K = K; a.lda = K; a.ldb = K; a.ldc = C;
  DgradLnArgs q = {(const bf16*)x, mean, rstd, gamma, (const bf16*)dres, (bf16*)dx, part, part_rows};
  // (K = 3 C: the packed qkv projection; K = C: a plain projection)
  if (C == 96 && K == 288) return dgrad_ln_launch_k<9, 3>(a, q, st);
  if (C == 64 && K == 192) return dgrad_ln_launch_k<6, 2>(a, q, st);
  if (C == 128 && K == 384) return dgrad_ln_launch_k<12, 4>(a, q, st);
  if (C == 96 && K == 96) return dgrad_ln_launch_k<3, 3>(a, q, st);
  if (C == 64 && K == 64) return dgrad_ln_launch_k<2, 2>(a, q, st);
  if (C == 128 && K == 128) return dgrad_ln_launch_k<4, 4>(a, q, st);
  return VTX_ERR_SHAPE;
}
bool dgrad_ln_shape_ok(int C, int K) { return (K == 3 * C || K == C) && (C == 64 || C == 96 || C == 128); }

extern "C" {
/* Test / tool entry of the fold (the layer calls take it through csrc/layer.hip): dx = dres + LN'(dy . W), the bits of
 * vtx_gemm(dy, wt) followed by vtx_layernorm_bwd(.., dres); part as in vtx_mlp_bwd_ln. */
int vtx_dgrad_ln(int dtype, const void* dy, const void* wt, const void* x, const float* mean, const float* rstd, const float* gamma,
                 const void* dres, void* dx, float* part, int part_rows, int64_t M, int C, int K, void* stream) {
  if (dtype != VTX_BF16) return VTX_ERR_DTYPE;
  if (M <= 0 || M > 0x7fffffff || !dgrad_ln_shape_ok(C, K)) return VTX_ERR_SHAPE;
  return dgrad_ln_launch(dy, wt, x, mean, rstd, gamma, dres, dx, part, part_rows, M, C, K, (hipStream_t)stream);
}
}  // extern "C"

static size_t skinny_smem(int N, int K) { return (size_t)N * (K + 8) * 2 + (size_t)N * 4 + (size_t)2 * K * 4; }
// column chunks (1, 2, 4) so that a chunk of the weight fits LDS; 0: none does
static int skinny_chunks(int N, int K) {
  for (int nc = 1; nc <= 4; nc *= 2)
    if (N % (32 * nc) == 0 && skinny_smem(N / nc, K) <= 150 * 1024) return nc;
  return 0;
}

bool gemm_skinny_ok(const GemmArgs& a) {
  if (vtx_opt(VTX_OPT_GEMM_SKINNY) == 0 || a.perm != nullptr) return false;
  // (launches whose epilogue READS a vector per output vector -- residual, z of act' -- stay on the tiled kernels unless option
  //  GEMM_SKINNY = 2: with 4 waves per CU the loads are exposed; measured fc2 dgrad 160 -> 215 us, proj forward 44 -> 47 us)
  if ((a.resid != nullptr || a.act == 2 || a.act == 4) && vtx_opt(VTX_OPT_GEMM_SKINNY) != 2) return false;
  // (K = 192, Swin stage 2, was built and measured: 6 k-steps per output pair on 4 waves per CU -- qkv forward 56.4 -> 54.5 us, fc1
  //  forward 125 -> 121 us: within noise of the tiled kernels, not routed here)
  if (a.K != 64 && a.K != 96 && a.K != 128) return false;
  if (a.N < 32 || a.M < 32768) return false;
  if ((a.lda % 8) || (a.ldb % 8) || (a.ldc % 8)) return false;
  return skinny_chunks(a.N, a.K) != 0;
}

template <int KS, int ACT, bool VEC, int WAVES> static int skinny_launch_kavw(const GemmArgs& a, hipStream_t st) {
  const int nc = skinny_chunks(a.N, a.K);
  const size_t smem = skinny_smem(a.N / nc, a.K);
  auto kern = gemm_skinny_kernel<KS, ACT, VEC, WAVES>;
  if (smem > 64 * 1024 &&
      hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
    return VTX_ERR_LAUNCH;
  hipLaunchKernelGGL(kern, dim3(vtx_cu_count_cached()), dim3(64 * WAVES), smem, st, a, nc, SkLnArgs{});   // one persistent workgroup per CU
  return vtx_check_launch();
}

// y = LN(x) . W^T + bias with the LayerNorm in the row-operand path (LNF): x [M][K = C], W [N][K]; ln_out / mean / rstd on the side
bool ln_gemm_ok(int dtype, int64_t M, int C, int N) {
  if ((vtx_opt(VTX_OPT_LN_FOLD) & 8) == 0 || dtype != VTX_BF16 || vtx_opt(VTX_OPT_GEMM_SKINNY) == 0) return false;
  if (C != 64 && C != 96 && C != 128) return false;
  if (N < 32 || N % 32 != 0 || M < 32768 || M > 0x7fffffff) return false;
  return skinny_chunks(N, C) != 0;
}
template <int KS> static int ln_gemm_launch_k(const GemmArgs& a, const SkLnArgs& q, hipStream_t st) {
  const int nc = skinny_chunks(a.N, a.K);
  const size_t smem = skinny_smem(a.N / nc, a.K);
  auto kern = gemm_skinny_kernel<KS, 0, false, 4, true>;
  if (smem > 64 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return VTX_ERR_LAUNCH;
  hipLaunchKernelGGL(kern, dim3(vtx_cu_count_cached()), dim3(256), smem, st, a, nc, q);
  return vtx_check_launch();
}
int ln_gemm_launch(const void* x, const float* gamma, const float* beta, float eps, void* ln_out, float* mean, float* rstd, const void* w,
                   const float* bias, void* y, int64_t M, int C, int N, hipStream_t st) {
  if (!x || !gamma || !beta || !ln_out || !mean || !rstd || !w || !y) return VTX_ERR_NULL;
  if (skinny_chunks(N, C) == 0) return VTX_ERR_SHAPE;
  GemmArgs a = {};
  a.A = x; a.B = w; a.C = y; a.M = (int)M; a.N = N; a.K = C; a.lda = C; a.ldb = C; a.ldc = N; a.bias = bias; a.rows_per_scale = 1;
  SkLnArgs q = {gamma, beta, eps, (bf16*)ln_out, mean, rstd};
  if (C == 64) return ln_gemm_launch_k<2>(a, q, st);
  if (C == 96) return ln_gemm_launch_k<3>(a, q, st);
  if (C == 128) return ln_gemm_launch_k<4>(a, q, st);
  return VTX_ERR_SHAPE;
}
extern "C" {
/* vtx_layernorm_fwd(x -> ln, mean, rstd) + vtx_gemm(ln, w, bias -> y) in one launch (round 6, option LN_FOLD bit 3): the same bits. */
int vtx_ln_gemm(int dtype, const void* x, const float* gamma, const float* beta, float eps, void* ln_out, float* mean, float* rstd,
                const void* w, const float* bias, void* y, int64_t M, int C, int N, void* stream) {
  if (dtype != VTX_BF16) return VTX_ERR_DTYPE;
  if (M <= 0 || M > 0x7fffffff || (C != 64 && C != 96 && C != 128) || N < 32 || N % 32) return VTX_ERR_SHAPE;
  return ln_gemm_launch(x, gamma, beta, eps, ln_out, mean, rstd, w, bias, y, M, C, N, (hipStream_t)stream);
}
}  // extern "C"
template <int KS, int ACT, bool VEC> static int skinny_launch_kav(const GemmArgs& a, hipStream_t st) {
  switch (vtx_opt(VTX_OPT_SKINNY_WAVES)) {
    case 16: return skinny_launch_kavw<KS, ACT, VEC, 16>(a, st);
    case 8: return skinny_launch_kavw<KS, ACT, VEC, 8>(a, st);
    default: return skinny_launch_kavw<KS, ACT, VEC, 4>(a, st);
  }
}

template <int KS> static int skinny_launch_k(const GemmArgs& a, hipStream_t st) {
  const bool vec = a.resid != nullptr;
  switch (a.act) {
    case 0: return vec ? skinny_launch_kav<KS, 0, true>(a, st) : skinny_launch_kav<KS, 0, false>(a, st);
    case 1: return vec ? skinny_launch_kav<KS, 1, true>(a, st) : skinny_launch_kav<KS, 1, false>(a, st);
    case 3: return vec ? skinny_launch_kav<KS, 3, true>(a, st) : skinny_launch_kav<KS, 3, false>(a, st);
    case 2: return skinny_launch_kav<KS, 2, true>(a, st);           // (act': z is always read)
    default: return skinny_launch_kav<KS, 4, true>(a, st);
  }
}

int gemm_skinny_launch(const GemmArgs& a, hipStream_t st) {
  if (a.K == 64) return skinny_launch_k<2>(a, st);
  if (a.K == 96) return skinny_launch_k<3>(a, st);
  return skinny_launch_k<4>(a, st);
}
