// FUSED MLP of the narrow stages (round 5): y = x1 + s * (silu(ln2 . W1^T + b1) . W2^T + b2) with C = 64 / 96 and both weights resident
// in LDS -- Swin-S stage 1 (C = 96, ff = 384, 401 408 rows), PVT-Small / Twins-SVT-S stage 1 (C = 64).
// Reference: models/layer.py:186-196 (PositionwiseFeedForward), the MLP half of models/swin_transformer.py:193-197.
//
// Why: at these widths every launch of the MLP is an HBM stream (profiles/round5_shape_table_swin_s.md: 3.5-4.4 TB/s on all of them) and
// most of the bytes are the two ff-wide intermediates -- z and h are written by fc1 (8 units of rows x C x 2 bytes), h is read by fc2
// (4), z by the fc2 dgrad (4), dz written by it (4) and read by the fc1 dgrad (4): 24 of the 29 units of those four launches.
//   forward  (mlp_fwd_kernel): ln2 and x1 in, y out -- 3 units; z and h never leave the registers;
//   backward (mlp_bwd_kernel): ln2 and dy in; z, h and dh are RECOMPUTED (two K = C products per hidden column), dz = s dh silu'(z),
//            dln2 = dz . W1 accumulated over the hidden columns in registers; h and dz go out ONCE, for the layer's grouped weight
//            gradient (dW2 = dy^T h, dW1 = dz^T ln2) -- 2 units in, 9 out, against 9 + 5 of the two dgrad launches, and nothing kept from
//            the forward but ln2.
// Structure = gemm_skinny.hip's: one persistent workgroup per CU, a wave streams 32-row blocks of the row operands from global memory
// straight into MFMA operand registers, products taken TRANSPOSED (weight rows are the MFMA's A operand, in the row order that leaves a
// lane 8 consecutive output columns of one row).  That layout is also the B-operand layout of the NEXT product's k-slots: lane (c, g)
// ends the first product with hidden columns 32 np + 8 g .. + 7 of row c, which is what it must supply for k-step np of h . W2^T (and of
// dz . W1) -- the intermediates go from accumulator registers to operand registers through the bf16 rounding the unfused path stores.
// LDS: W1 [ff][C + 8] + (forward) W2 [C][ff + 8] | (backward) W2^T [ff][C + 8] (transposed while it is copied in: the backward does not
// depend on the transposed weight copies of a weight scope); the backward's third product contracts over the ROWS of W1 -- its A
// fragments come out of the same W1 image through ds_read_b64_tr_b16 (lane p of a 16-lane group points at row p >> 2, 4-column piece
// p & 3 of a [4 hidden][16 column] block and receives block column p: pieces at columns 8 q + 4 j give lane p column 8 (p >> 2) + 4 j +
// (p & 3), the row order above).
// Element values: the same products in the same k order and the same epilogue expressions as the launches they replace (z rounded to
// bf16 before the activation, h and dz rounded to bf16 before the next product): bit-identical to the unfused path
// (tests/test_gpu_mlp_fused.py).
#include <type_traits>

#include "gemm_common.h"
#include "ln_fold.h"
#include "options.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short mf_s16x4;
typedef __attribute__((address_space(3))) mf_s16x4 mf_lds_s16x4;

struct MlpArgs {
  const bf16* a;        // ln2 [M][C]
  const bf16* w1;       // [ff][C]
  const bf16* w2;       // [C][ff]
  const float* b1;      // [ff]
  const float* b2;      // [C]                      (forward)
  const bf16* resid;    // x1 [M][C]                (forward)
  bf16* y;              // [M][C]                   (forward)
  const bf16* dy;       // [M][C]                   (backward)
  bf16* h;              // [M][ff]                  (backward: out; forward: optional out)
  bf16* z;              // [M][ff]                  (forward: optional out)
  bf16* dz;             // [M][ff]                  (backward: out)
  bf16* dx;             // dln2 [M][C]              (backward: out)
  const float* rowscale;
  int rows_per_scale;
  int M, ff;
  // LayerNorm-backward fold (backward, LNB): dx receives dx1 = dy + LN2'(dln2) instead of dln2
  const bf16* x1;       // [M][C]: the LayerNorm's input
  const float* mean;    // [M]
  const float* rstd;    // [M]
  const float* gamma;   // [C]
  float* part;          // [part_rows][2 C] fp32: dgamma | dbeta partial rows (the stand-alone launch's workspace layout)
  int part_rows;        // rows the deferred column reduce will sum: this launch writes gridDim.x of them and zeroes the rest
  // LayerNorm-forward fold (forward, LNF): `a` is not read -- the rows come from `resid` (x1), are normalised in registers with
  // gamma / beta / eps and stored to ln_out / mean_out / rstd_out on the side
  const float* beta;
  float eps;
  bf16* ln_out;
  float* mean_out;
  float* rstd_out;
};

constexpr int MF_ROWS = 32;

template <int KS> __device__ __forceinline__ void mf_load_rows(bf16x8 (&f)[2][KS], const bf16* __restrict__ src, int rb, int M, int c, int g) {
  constexpr int C = 32 * KS;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int row = min(rb * MF_ROWS + mt * 16 + c, M - 1);     // (rows past M are never stored)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) f[mt][ks] = *reinterpret_cast<const bf16x8*>(src + (int64_t)row * C + ks * 32 + g * 8);
  }
}

// ------------------------------------------------------------------------------------------------------------------ forward
// PF: the next block's row operands are requested before the current block is multiplied and the residual vectors before the hidden loop
// (one or two waves per SIMD: nobody else hides the latency); !PF (three or four waves per SIMD): loads where they are used, 24 + 24
// registers less per lane
// ZH: z and h are also written (tests; the layer calls never ask).  Rows past M (the last block only) are row M - 1 again -- operands and
// addresses: the same bits stored to the same place, no exec-mask branch anywhere in the hidden loop.
// LNF (round 6, option LN_FOLD bit 2): the LayerNorm forward of norm_ff runs on the row operands (ln_fold.h LnFwdFold): the kernel reads
// x1 -- which it reads anyway, as the residual -- instead of ln2, and writes ln2 / mean / rstd on the side for the backward and the
// weight gradient: 3 units (x1 in, ln2 and y out) against the 5 of the two launches; ln2, mean, rstd, y bit-identical.
template <int KS, int WAVES, bool PF, bool ZH, bool LNF = false>
__global__ __launch_bounds__(64 * WAVES) void mlp_fwd_kernel(MlpArgs p) {
  constexpr int C = 32 * KS, S1 = C + 8, NT = 64 * WAVES, MF_UNR = PF ? 2 : 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char mf_smem[];
  const int ff = p.ff, S2 = ff + 8, M = p.M;
  bf16* w1s = reinterpret_cast<bf16*>(mf_smem);                       // [ff][C + 8]
  bf16* w2s = w1s + (size_t)ff * S1;                                  // [C][ff + 8]
  float* b1s = reinterpret_cast<float*>(w2s + (size_t)C * S2);        // [ff]
  float* b2s = b1s + ff;                                              // [C]
  for (int i = threadIdx.x; i < ff * (C / 8); i += NT) {
    const int n = i / (C / 8), q = i - n * (C / 8);
    *reinterpret_cast<bf16x8*>(w1s + n * S1 + q * 8) = *reinterpret_cast<const bf16x8*>(p.w1 + (int64_t)n * C + q * 8);
  }
  for (int i = threadIdx.x; i < C * (ff / 8); i += NT) {
    const int n = i / (ff / 8), q = i - n * (ff / 8);
    *reinterpret_cast<bf16x8*>(w2s + n * S2 + q * 8) = *reinterpret_cast<const bf16x8*>(p.w2 + (int64_t)n * ff + q * 8);
  }
  for (int i = threadIdx.x; i < ff; i += NT) b1s[i] = p.b1 ? p.b1[i] : 0.f;
  for (int i = threadIdx.x; i < C; i += NT) b2s[i] = p.b2 ? p.b2[i] : 0.f;
  float* lgs = b2s + C;                                               // LNF: gamma | beta [2][C]
  if constexpr (LNF)
    for (int i = threadIdx.x; i < C; i += NT) { lgs[i] = p.gamma[i]; lgs[C + i] = p.beta[i]; }
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const int nrb = (M + MF_ROWS - 1) / MF_ROWS;
  const int stride = gridDim.x * WAVES;
  const int npairs = ff >> 5;
  const int nperm = 8 * (c >> 2) + (c & 3);              // weight row this lane supplies as MFMA operand row c (+ 32 pair + 4 j)
  const bf16* __restrict__ rows_src = LNF ? p.resid : p.a;

  bf16x8 an[2][KS];
  int rb = blockIdx.x * WAVES + wave;
  if (PF && rb < nrb) mf_load_rows<KS>(an, rows_src, rb, M, c, g);
  for (; rb < nrb; rb += stride) {
    bf16x8 a[2][KS];
    if constexpr (PF) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) a[mt][ks] = an[mt][ks];
      if (rb + stride < nrb) mf_load_rows<KS>(an, rows_src, rb + stride, M, c, g);
    } else {
      mf_load_rows<KS>(a, rows_src, rb, M, c, g);
    }
    if constexpr (LNF) {
      // raw rows -> normalised rows, in place: these are the first product's operands; ln2 / mean / rstd go out on the side (rows past M
      // are row M - 1 again: the same bits to the same place)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int rc = min(rb * MF_ROWS + mt * 16 + c, M - 1);
        bf16x8 yv[KS];
        float mu, rs;
        LnFwdFold<KS>::row(a[mt], lgs, lgs + C, p.eps, g, yv, mu, rs);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          a[mt][ks] = yv[ks];
          *reinterpret_cast<bf16x8*>(p.ln_out + (int64_t)rc * C + ks * 32 + g * 8) = yv[ks];
        }
        if (g == 0) { p.mean_out[rc] = mu; p.rstd_out[rc] = rs; }
      }
    }
    int row[2];
    float rsc[2];
    Vec8<bf16> rv[2][KS];                                  // residual vectors of the output pairs (PF: requested now, used behind the loop)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      row[mt] = min(rb * MF_ROWS + mt * 16 + c, M - 1);
      if constexpr (PF || ZH) rsc[mt] = p.rowscale ? p.rowscale[row[mt] / p.rows_per_scale] : 1.f;
      if constexpr (PF) {
#pragma unroll
        for (int tp = 0; tp < KS; ++tp)
          rv[mt][tp] = p.resid ? load8<bf16>(p.resid + (int64_t)row[mt] * C + tp * 32 + 8 * g) : vec8_zero<bf16>();
      }
    }
    f32x4 oacc[2][2 * KS];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int t = 0; t < 2 * KS; ++t) oacc[mt][t] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll MF_UNR
    for (int np = 0; np < npairs; ++np) {
      // ---- z = ln2 . W1^T for hidden columns 32 np + 8 g .. + 7 of rows (mt, c)
      f32x4 zacc[2][2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int j = 0; j < 2; ++j) zacc[mt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = np * 32 + nperm + 4 * j;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          Vec8<bf16> wf;
          wf.v = *reinterpret_cast<const bf16x8*>(w1s + n * S1 + ks * 32 + g * 8);
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) {
            Vec8<bf16> af;
            af.v = a[mt][ks];
            mma16(wf, af, zacc[mt][j]);
          }
        }
      }
      const int col = np * 32 + 8 * g;
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(b1s + col), b1 = *reinterpret_cast<const f32x4*>(b1s + col + 4);
      Vec8<bf16> hv[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        float val[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) { val[r] = zacc[mt][0][r] + b0[r]; val[4 + r] = zacc[mt][1][r] + b1[r]; }
        Vec8<bf16> zv;
#pragma unroll
        for (int e = 0; e < 8; ++e) zv.set(e, val[e]);         // activation of the ROUNDED pre-activation (what the backward recomputes)
#pragma unroll
        for (int e = 0; e < 8; ++e) hv[mt].set(e, silu_f(zv.get(e)) + 0.f);      // (+ 0: the unfused epilogue's `v * scale + residual` turns -0 into +0)
        if constexpr (ZH) {
          store8<bf16>(p.z + (int64_t)row[mt] * ff + col, zv);
          store8<bf16>(p.h + (int64_t)row[mt] * ff + col, hv[mt]);
        }
      }
      // ---- y^T += W2[:, 32 np ..] . h^T: k-slots (g, e) <-> hidden column 32 np + 8 g + e, which is what hv holds
#pragma unroll
      for (int tp = 0; tp < KS; ++tp)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int n = tp * 32 + nperm + 4 * j;
          Vec8<bf16> wf;
          wf.v = *reinterpret_cast<const bf16x8*>(w2s + n * S2 + np * 32 + g * 8);
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) mma16(wf, hv[mt], oacc[mt][2 * tp + j]);
        }
    }
    // oacc[mt][2 tp + j][r] = (h . W2^T)[row (mt, c)][32 tp + 8 g + 4 j + r]
    if constexpr (!PF) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        if constexpr (!ZH) {       // (three waves per SIMD, 168 registers: nothing of the output side is alive across the hidden loop -- no spill)
          row[mt] = min(rb * MF_ROWS + mt * 16 + c, M - 1);
          rsc[mt] = p.rowscale ? p.rowscale[row[mt] / p.rows_per_scale] : 1.f;
        }
#pragma unroll
        for (int tp = 0; tp < KS; ++tp)
          rv[mt][tp] = p.resid ? load8<bf16>(p.resid + (int64_t)row[mt] * C + tp * 32 + 8 * g) : vec8_zero<bf16>();
      }
    }
#pragma unroll
    for (int tp = 0; tp < KS; ++tp) {
      const int col = tp * 32 + 8 * g;
      const f32x4 c0 = *reinterpret_cast<const f32x4*>(b2s + col), c1 = *reinterpret_cast<const f32x4*>(b2s + col + 4);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        float val[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) { val[r] = oacc[mt][2 * tp][r] + c0[r]; val[4 + r] = oacc[mt][2 * tp + 1][r] + c1[r]; }
        Vec8<bf16> o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o.set(e, val[e] * rsc[mt] + rv[mt][tp].get(e));
        store8<bf16>(p.y + (int64_t)row[mt] * C + col, o);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------ backward
// NTS: h and dz (616 MB per Swin-S stage-1 launch, read next by the weight-gradient launch only) are stored non-temporally.
// ABL: timing probes (results garbage): 1 no h / dz stores | 2 no silu / silu' arithmetic | 3 no dln2 product
// PAIR: the h / dz vectors of an even 32-column step are stored together with the following odd step's: every row's 128-byte line is
// written by two instructions issued back to back (ff % 64 == 0).
// LNB (round 6, option LN_FOLD bit 0): the LayerNorm backward that follows this launch in the layer (models/swin_transformer.py:196,
// layer.py: x1 + ff(norm_ff(x1))) runs in its epilogue.  The block's dln2 sits in the accumulators in the layout of the dy operand -- lane
// (c, g) holds columns 32 tp + 8 g .. + 7 of rows (mt, c), i.e. whole 8-element vectors v = 4 tp + g of the stand-alone kernel's lane
// groups <16 lanes, 1 vector> -- so dx1 = dy + rstd (g - mean(g) - xhat mean(g xhat)), g = bf16(dln2) gamma, needs x1 (one more row
// operand, requested in front of the hidden loop) and two row sums over the four lanes c, c + 16, c + 32, c + 48.  dln2 is neither
// written nor read back (2 of the stand-alone launch's 4 units), and that launch is gone (Swin-S stage 1: 76 us of 401 408 x 96).
// Bit-identical dx1: the element expressions are the stand-alone kernel's (vtx_common.h ln_bwd_elem_*), dln2 goes through its bf16
// rounding, and the row sums are formed in group_sum<16>'s association -- per-vector partials in element order, then
// ((v0 + v1) + (v2 + v3)) per quad, (q0 + q1) + (q2 + q3) with the quads past C / 32 zero.  dgamma / dbeta: per-lane register sums over
// the rows a lane visits, DPP sums over c, one partial row per workgroup in the stand-alone workspace layout (the rows beyond
// gridDim.x zeroed): another row grouping than the stand-alone launch, i.e. fp32 summation order, not bits.
template <int KS, int WAVES, bool PF, bool NTS = false, int ABL = 0, bool PAIR = false, bool LNB = false>
__global__ __launch_bounds__(64 * WAVES) void mlp_bwd_kernel(MlpArgs p) {
  constexpr int C = 32 * KS, S1 = C + 8, NT = 64 * WAVES, MF_UNR = (PF && WAVES <= 4) ? 2 : 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char mf_smem[];
  const int ff = p.ff, M = p.M;
  bf16* w1s = reinterpret_cast<bf16*>(mf_smem);                       // [ff][C + 8]: W1
  bf16* w2t = w1s + (size_t)ff * S1;                                  // [ff][C + 8]: W2^T
  float* b1s = reinterpret_cast<float*>(w2t + (size_t)ff * S1);       // [ff]
  for (int i = threadIdx.x; i < ff * (C / 8); i += NT) {
    const int n = i / (C / 8), q = i - n * (C / 8);
    *reinterpret_cast<bf16x8*>(w1s + n * S1 + q * 8) = *reinterpret_cast<const bf16x8*>(p.w1 + (int64_t)n * C + q * 8);
  }
  for (int i = threadIdx.x; i < C * (ff / 8); i += NT) {
    const int n = i % C, q = i / C;                                   // (consecutive threads: consecutive LDS columns of 8 rows)
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(p.w2 + (int64_t)n * ff + q * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) w2t[(q * 8 + e) * S1 + n] = v[e];
  }
  for (int i = threadIdx.x; i < ff; i += NT) b1s[i] = p.b1 ? p.b1[i] : 0.f;
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const int nrb = (M + MF_ROWS - 1) / MF_ROWS;
  const int stride = gridDim.x * WAVES;
  const int npairs = ff >> 5;
  const int nperm = 8 * (c >> 2) + (c & 3);
  // transpose-read address of this lane inside a [32 hidden][32 columns] block of the W1 image: row 8 g + (c >> 2) (+ 4 for the
  // second half of the k-slots), 4-column piece at column 8 (c & 3) (+ 4 j)
  const int tr_off = (8 * g + (c >> 2)) * S1 + 8 * (c & 3);

  LnFold<LNB ? KS : 2> lnf;                                // (ln_fold.h)
  if constexpr (LNB) lnf.init(p.gamma, g);

  bf16x8 an[2][KS], dn[2][KS];
  int rb = blockIdx.x * WAVES + wave;
  if (PF && rb < nrb) { mf_load_rows<KS>(an, p.a, rb, M, c, g); mf_load_rows<KS>(dn, p.dy, rb, M, c, g); }
  for (; rb < nrb; rb += stride) {
    bf16x8 a[2][KS], d[2][KS];
    if constexpr (PF) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) { a[mt][ks] = an[mt][ks]; d[mt][ks] = dn[mt][ks]; }
      if (rb + stride < nrb) { mf_load_rows<KS>(an, p.a, rb + stride, M, c, g); mf_load_rows<KS>(dn, p.dy, rb + stride, M, c, g); }
    } else {
      mf_load_rows<KS>(a, p.a, rb, M, c, g);
      mf_load_rows<KS>(d, p.dy, rb, M, c, g);
    }
    int row[2];
    float rsc[2];
    bool ok[2];
    // (the stores stay under `ok`: with unconditional stores of clamped rows -- as in the forward -- this kernel measured 252 instead of
    //  229-232 us at 401 408 x 96 x 384, profiles/round5_mlp_fused.txt)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      row[mt] = rb * MF_ROWS + mt * 16 + c;
      ok[mt] = row[mt] < M;
      rsc[mt] = (ok[mt] && p.rowscale) ? p.rowscale[row[mt] / p.rows_per_scale] : 1.f;
    }
    bf16x8 xr[LNB ? 2 : 1][LNB ? KS : 1];                   // LNB: the LayerNorm's input rows and statistics, used behind the hidden loop
    float lmu[2], lrs[2];
    if constexpr (LNB) {
      mf_load_rows<KS>(xr, p.x1, rb, M, c, g);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) { const int rc = min(row[mt], M - 1); lmu[mt] = p.mean[rc]; lrs[mt] = p.rstd[rc]; }
    }
    f32x4 xacc[2][2 * KS];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int t = 0; t < 2 * KS; ++t) xacc[mt][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    Vec8<bf16> hheld[2], dzheld[2];                       // PAIR: an even pair's h / dz, stored with the odd pair's (whole 128-byte lines back to back)
    auto step = [&](int np, auto PH_) __attribute__((always_inline)) {
      constexpr int PH = decltype(PH_)::value;             // 0 store now | 1 hold | 2 store the held vectors and these
      // ---- z = ln2 . W1^T (recomputed) and dh = dy . W2 for hidden columns 32 np + 8 g .. + 7 of rows (mt, c)
      f32x4 zacc[2][2], hacc[2][2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int j = 0; j < 2; ++j) { zacc[mt][j] = f32x4{0.f, 0.f, 0.f, 0.f}; hacc[mt][j] = zacc[mt][j]; }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = np * 32 + nperm + 4 * j;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          Vec8<bf16> wf, vf;
          wf.v = *reinterpret_cast<const bf16x8*>(w1s + n * S1 + ks * 32 + g * 8);
          vf.v = *reinterpret_cast<const bf16x8*>(w2t + n * S1 + ks * 32 + g * 8);
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) {
            Vec8<bf16> af, df;
            af.v = a[mt][ks];
            df.v = d[mt][ks];
            mma16(wf, af, zacc[mt][j]);
            mma16(vf, df, hacc[mt][j]);
          }
        }
      }
      const int col = np * 32 + 8 * g;
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(b1s + col), b1 = *reinterpret_cast<const f32x4*>(b1s + col + 4);
      Vec8<bf16> dzv[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        float zval[8], dval[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          zval[r] = zacc[mt][0][r] + b0[r]; zval[4 + r] = zacc[mt][1][r] + b1[r];
          dval[r] = hacc[mt][0][r]; dval[4 + r] = hacc[mt][1][r];
        }
        Vec8<bf16> zv, hv;
#pragma unroll
        for (int e = 0; e < 8; ++e) zv.set(e, zval[e]);
        if constexpr (ABL == 2) {
          hv = zv;
#pragma unroll
          for (int e = 0; e < 8; ++e) dzv[mt].set(e, dval[e] * rsc[mt]);
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) hv.set(e, silu_f(zv.get(e)) + 0.f);
#pragma unroll
          for (int e = 0; e < 8; ++e) { float v = dval[e]; v *= dsilu_f(zv.get(e)); dzv[mt].set(e, v * rsc[mt] + 0.f); }
        }
        if constexpr (PH == 1) {
          hheld[mt] = hv; dzheld[mt] = dzv[mt];
        } else if (ok[mt] && (ABL != 1 || hv.get(0) == 12345.678f)) {
          bf16* hp = p.h + (int64_t)row[mt] * ff + col;
          bf16* zp = p.dz + (int64_t)row[mt] * ff + col;
          if constexpr (NTS) {
            __builtin_nontemporal_store(hv.v, reinterpret_cast<bf16x8*>(hp));
            __builtin_nontemporal_store(dzv[mt].v, reinterpret_cast<bf16x8*>(zp));
          } else {
            if constexpr (PH == 2) { store8<bf16>(hp - 32, hheld[mt]); store8<bf16>(hp, hv); store8<bf16>(zp - 32, dzheld[mt]); store8<bf16>(zp, dzv[mt]); }
            else { store8<bf16>(hp, hv); store8<bf16>(zp, dzv[mt]); }
          }
        }
      }
      // ---- dln2^T += W1[32 np .., :]^T . dz^T: the A fragments are COLUMNS of the W1 image (transpose reads)
      const bf16* blk = w1s + np * 32 * S1 + tr_off;
      if constexpr (ABL == 3) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) xacc[mt][0][0] += dzv[mt].get(0);
        return;
      }
#pragma unroll
      for (int tp = 0; tp < KS; ++tp)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const bf16* ap = blk + tp * 32 + 4 * j;
          const mf_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((mf_lds_s16x4*)ap);
          const mf_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((mf_lds_s16x4*)(ap + 4 * S1));
          typedef __attribute__((ext_vector_type(8))) short s16x8;
          const s16x8 w8 = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);     // (whole-vector bitcast only: see wg_frag)
          Vec8<bf16> wf;
          wf.v = __builtin_bit_cast(bf16x8, w8);
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) mma16(wf, dzv[mt], xacc[mt][2 * tp + j]);
        }
    };
    if constexpr (PAIR) {
      for (int np = 0; np < npairs; np += 2) { step(np, std::integral_constant<int, 1>{}); step(np + 1, std::integral_constant<int, 2>{}); }
    } else {
#pragma unroll MF_UNR
      for (int np = 0; np < npairs; ++np) step(np, std::integral_constant<int, 0>{});
    }
    // xacc[mt][2 tp + j][r] = (dz . W1)[row (mt, c)][32 tp + 8 g + 4 j + r]
    if constexpr (!LNB) {
#pragma unroll
      for (int tp = 0; tp < KS; ++tp) {
        const int col = tp * 32 + 8 * g;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          if (!ok[mt]) continue;
          Vec8<bf16> o;
#pragma unroll
          for (int r = 0; r < 4; ++r) { o.set(r, xacc[mt][2 * tp][r]); o.set(4 + r, xacc[mt][2 * tp + 1][r]); }
          store8<bf16>(p.dx + (int64_t)row[mt] * C + col, o);
        }
      }
    } else {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        Vec8<bf16> dl[KS];
#pragma unroll
        for (int tp = 0; tp < KS; ++tp)
#pragma unroll
          for (int r = 0; r < 4; ++r) { dl[tp].set(r, xacc[mt][2 * tp][r]); dl[tp].set(4 + r, xacc[mt][2 * tp + 1][r]); }      // (the bf16 the stand-alone path stores)
        // (dres: the residual-stream gradient that bypasses the norm is dy itself, already here as the second product's row operand)
        lnf.row(dl, xr[mt], d[mt], lmu[mt], lrs[mt], ok[mt], p.dx + (int64_t)row[mt] * C + 8 * g);
      }
    }
  }
  // dgamma / dbeta partial row of the workgroup; the scratch is the (now dead) W1 image
  if constexpr (LNB) lnf.template finish<WAVES>(reinterpret_cast<float*>(mf_smem), p.part, p.part_rows, wave, c, g);
}

size_t mlp_fwd_smem(int C, int ff) { return (size_t)ff * (C + 8) * 2 + (size_t)C * (ff + 8) * 2 + (size_t)(ff + C) * 4 + (size_t)2 * C * 4; }
size_t mlp_bwd_smem(int C, int ff) { return (size_t)ff * (C + 8) * 4 + (size_t)ff * 4; }

template <int KS, int WAVES, bool PF> int mlp_fwd_launch_k(const MlpArgs& a, hipStream_t st) {
  const size_t smem = mlp_fwd_smem(32 * KS, a.ff);
  if ((a.z == nullptr) != (a.h == nullptr)) return VTX_ERR_NULL;             // (both or neither)
  const bool lnf = a.ln_out != nullptr;
  if (lnf && (a.z != nullptr || !a.resid || !a.gamma || !a.beta || !a.mean_out || !a.rstd_out)) return VTX_ERR_NULL;
  auto kern = lnf ? mlp_fwd_kernel<KS, WAVES, PF, false, true> : (a.z != nullptr ? mlp_fwd_kernel<KS, WAVES, PF, true> : mlp_fwd_kernel<KS, WAVES, PF, false>);
  if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return VTX_ERR_LAUNCH;
  hipLaunchKernelGGL(kern, dim3(vtx_cu_count_cached()), dim3(64 * WAVES), smem, st, a);
  return vtx_check_launch();
}
template <int KS, int WAVES, bool PF, bool NTS = false, int ABL = 0, bool PAIR = false, bool LNB = false> int mlp_bwd_launch_k(const MlpArgs& a, hipStream_t st) {
  const size_t smem = mlp_bwd_smem(32 * KS, a.ff);
  if (PAIR && a.ff % 64 != 0) return VTX_ERR_SHAPE;
  if (LNB && (!a.x1 || !a.mean || !a.rstd || !a.gamma || !a.part || a.part_rows < vtx_cu_count_cached())) return VTX_ERR_NULL;
  auto kern = mlp_bwd_kernel<KS, WAVES, PF, NTS, ABL, PAIR, LNB>;
  if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return VTX_ERR_LAUNCH;
  hipLaunchKernelGGL(kern, dim3(vtx_cu_count_cached()), dim3(64 * WAVES), smem, st, a);
  return vtx_check_launch();
}
// option MLP_FUSED: 1 = the defaults below; 100 b + f = variant codes (tools / tests): forward f in {4, 8 (prefetching), 9 (8 waves, loads in
// place), 12, 16}, backward b in {4 (4 waves, prefetching), 6 (the same with paired stores: the default), 7 (8 waves, prefetching), 8 (loads in place), 5 / 9 (the same with non-temporal h / dz stores)}
constexpr int MF_FWD_DEFAULT = 12, MF_BWD_DEFAULT = 6;      // (profiles/round5_mlp_fused.txt)
int mf_fwd_code() { const int o = vtx_opt(VTX_OPT_MLP_FUSED); return o >= 100 ? o % 100 : MF_FWD_DEFAULT; }
int mf_bwd_code() { const int o = vtx_opt(VTX_OPT_MLP_FUSED); return o >= 100 ? o / 100 : MF_BWD_DEFAULT; }
template <int KS> int mlp_fwd_launch(const MlpArgs& a, hipStream_t st) {
  if (a.ln_out != nullptr) return mlp_fwd_launch_k<KS, 12, false>(a, st);      // LayerNorm-forward fold: the default variant only
  switch (mf_fwd_code()) {
    case 4: return mlp_fwd_launch_k<KS, 4, true>(a, st);
    case 8: return mlp_fwd_launch_k<KS, 8, true>(a, st);
    case 9: return mlp_fwd_launch_k<KS, 8, false>(a, st);
    case 12: return mlp_fwd_launch_k<KS, 12, false>(a, st);
    case 16: return mlp_fwd_launch_k<KS, 16, false>(a, st);
    default: return VTX_ERR_SHAPE;
  }
}
template <int KS> int mlp_bwd_launch(const MlpArgs& a, hipStream_t st) {
  if (a.x1 != nullptr) {                                     // LayerNorm-backward fold: the default variant (four waves, prefetching) only
    if (a.ff % 64 != 0) return mlp_bwd_launch_k<KS, 4, true, false, 0, false, true>(a, st);
    return mlp_bwd_launch_k<KS, 4, true, false, 0, true, true>(a, st);
  }
  switch (mf_bwd_code()) {
    case 4: return mlp_bwd_launch_k<KS, 4, true>(a, st);
    case 5: return mlp_bwd_launch_k<KS, 4, true, true>(a, st);
    case 6:                                                  // (paired stores take two 32-column pairs per step: ff = 32 x odd runs the
      if (a.ff % 64 != 0) return mlp_bwd_launch_k<KS, 4, true>(a, st);   //  unpaired variant of the same kernel, bit-identical -- mlp_fused_ok admits ff % 32 == 0)
      return mlp_bwd_launch_k<KS, 4, true, false, 0, true>(a, st);
    case 7: return mlp_bwd_launch_k<KS, 8, true>(a, st);
    case 8: return mlp_bwd_launch_k<KS, 8, false>(a, st);
    case 9: return mlp_bwd_launch_k<KS, 8, false, true>(a, st);
#ifdef VTX_MLP_ABLATE
    case 41: return mlp_bwd_launch_k<KS, 4, true, false, 1>(a, st);
    case 42: return mlp_bwd_launch_k<KS, 4, true, false, 2>(a, st);
    case 43: return mlp_bwd_launch_k<KS, 4, true, false, 3>(a, st);
    case 81: return mlp_bwd_launch_k<KS, 8, false, false, 1>(a, st);
    case 82: return mlp_bwd_launch_k<KS, 8, false, false, 2>(a, st);
#endif
    default: return VTX_ERR_SHAPE;
  }
}

}  // namespace

// The layer calls (csrc/layer.hip) ask here, forward and backward with the same arguments: a forward that took the fused kernel kept
// neither z nor h, so its backward must take the fused one as well.
bool mlp_fused_ok(int dtype, int64_t M, int C, int ff) {
  if (vtx_opt(VTX_OPT_MLP_FUSED) == 0 || dtype != VTX_BF16) return false;
  if (C != 64 && C != 96) return false;
  if (ff % 32 != 0 || ff < 32 || M < 32768 || M > 0x7fffffff) return false;
  const size_t lim = 160 * 1024;
  return mlp_fwd_smem(C, ff) <= lim && mlp_bwd_smem(C, ff) <= lim;
}

int mlp_fused_fwd(const void* ln2, const void* w1, const float* b1, const void* w2, const float* b2, const void* resid,
                  const float* rowscale, int rows_per_scale, void* y, void* z, void* h, int64_t M, int C, int ff, hipStream_t st) {
  if (!ln2 || !w1 || !w2 || !y) return VTX_ERR_NULL;
  MlpArgs a = {};
  a.a = (const bf16*)ln2; a.w1 = (const bf16*)w1; a.w2 = (const bf16*)w2; a.b1 = b1; a.b2 = b2; a.resid = (const bf16*)resid;
  a.y = (bf16*)y; a.z = (bf16*)z; a.h = (bf16*)h; a.rowscale = rowscale; a.rows_per_scale = rows_per_scale > 0 ? rows_per_scale : 1;
  a.M = (int)M; a.ff = ff;
  if (C == 96) return mlp_fwd_launch<3>(a, st);
  if (C == 64) return mlp_fwd_launch<2>(a, st);
  return VTX_ERR_SHAPE;
}

int mlp_fused_bwd(const void* ln2, const void* dy, const void* w1, const float* b1, const void* w2, const float* rowscale,
                  int rows_per_scale, void* h, void* dz, void* dln2, int64_t M, int C, int ff, hipStream_t st) {
  if (!ln2 || !dy || !w1 || !w2 || !h || !dz || !dln2) return VTX_ERR_NULL;
  MlpArgs a = {};
  a.a = (const bf16*)ln2; a.dy = (const bf16*)dy; a.w1 = (const bf16*)w1; a.w2 = (const bf16*)w2; a.b1 = b1;
  a.h = (bf16*)h; a.dz = (bf16*)dz; a.dx = (bf16*)dln2; a.rowscale = rowscale; a.rows_per_scale = rows_per_scale > 0 ? rows_per_scale : 1;
  a.M = (int)M; a.ff = ff;
  if (C == 96) return mlp_bwd_launch<3>(a, st);
  if (C == 64) return mlp_bwd_launch<2>(a, st);
  return VTX_ERR_SHAPE;
}

bool mlp_fused_lnf_ok(int dtype, int64_t M, int C, int ff) { return (vtx_opt(VTX_OPT_LN_FOLD) & 4) != 0 && mlp_fused_ok(dtype, M, C, ff); }

int mlp_fused_fwd_ln(const void* x1, const float* gamma, const float* beta, float eps, void* ln2, float* mean, float* rstd, const void* w1,
                     const float* b1, const void* w2, const float* b2, const float* rowscale, int rows_per_scale, void* y, int64_t M, int C,
                     int ff, hipStream_t st) {
  if (!x1 || !gamma || !beta || !ln2 || !mean || !rstd || !w1 || !w2 || !y) return VTX_ERR_NULL;
  MlpArgs a = {};
  a.w1 = (const bf16*)w1; a.w2 = (const bf16*)w2; a.b1 = b1; a.b2 = b2; a.resid = (const bf16*)x1;
  a.y = (bf16*)y; a.rowscale = rowscale; a.rows_per_scale = rows_per_scale > 0 ? rows_per_scale : 1;
  a.M = (int)M; a.ff = ff;
  a.gamma = gamma; a.beta = beta; a.eps = eps; a.ln_out = (bf16*)ln2; a.mean_out = mean; a.rstd_out = rstd;
  if (C == 96) return mlp_fwd_launch<3>(a, st);
  if (C == 64) return mlp_fwd_launch<2>(a, st);
  return VTX_ERR_SHAPE;
}

bool mlp_fused_ln_ok(int dtype, int64_t M, int C, int ff) {
  return (vtx_opt(VTX_OPT_LN_FOLD) & 1) != 0 && mlp_fused_ok(dtype, M, C, ff) && vtx_layernorm_bwd_blocks(M, C) >= vtx_cu_count_cached();
}

int mlp_fused_bwd_ln(const void* ln2, const void* dy, const void* w1, const float* b1, const void* w2, const float* rowscale,
                     int rows_per_scale, void* h, void* dz, const void* x1, const float* mean, const float* rstd, const float* gamma,
                     void* dx1, float* part, int part_rows, int64_t M, int C, int ff, hipStream_t st) {
  if (!ln2 || !dy || !w1 || !w2 || !h || !dz || !x1 || !mean || !rstd || !gamma || !dx1 || !part) return VTX_ERR_NULL;
  MlpArgs a = {};
  a.a = (const bf16*)ln2; a.dy = (const bf16*)dy; a.w1 = (const bf16*)w1; a.w2 = (const bf16*)w2; a.b1 = b1;
  a.h = (bf16*)h; a.dz = (bf16*)dz; a.dx = (bf16*)dx1; a.rowscale = rowscale; a.rows_per_scale = rows_per_scale > 0 ? rows_per_scale : 1;
  a.M = (int)M; a.ff = ff;
  a.x1 = (const bf16*)x1; a.mean = mean; a.rstd = rstd; a.gamma = gamma; a.part = part; a.part_rows = part_rows;
  if (C == 96) return mlp_bwd_launch<3>(a, st);
  if (C == 64) return mlp_bwd_launch<2>(a, st);
  return VTX_ERR_SHAPE;
}

extern "C" {

int vtx_mlp_fused_ok(int dtype, int64_t M, int C, int ff) { return mlp_fused_ok(dtype, M, C, ff) ? 1 : 0; }

int vtx_mlp_fwd(int dtype, const void* ln2, const void* w1, const float* b1, const void* w2, const float* b2, const void* resid,
                const float* rowscale, int rows_per_scale, void* y, void* z, void* h, int64_t M, int C, int ff, void* stream) {
  if (dtype != VTX_BF16) return VTX_ERR_DTYPE;
  if (M <= 0 || !mlp_fused_ok(dtype, M > 32768 ? M : 32768, C, ff)) return VTX_ERR_SHAPE;      // (any row count through the C ABI)
  return mlp_fused_fwd(ln2, w1, b1, w2, b2, resid, rowscale, rows_per_scale, y, z, h, M, C, ff, (hipStream_t)stream);
}

int vtx_mlp_bwd(int dtype, const void* ln2, const void* dy, const void* w1, const float* b1, const void* w2, const float* rowscale,
                int rows_per_scale, void* h, void* dz, void* dln2, int64_t M, int C, int ff, void* stream) {
  if (dtype != VTX_BF16) return VTX_ERR_DTYPE;
  if (M <= 0 || !mlp_fused_ok(dtype, M > 32768 ? M : 32768, C, ff)) return VTX_ERR_SHAPE;
  return mlp_fused_bwd(ln2, dy, w1, b1, w2, rowscale, rows_per_scale, h, dz, dln2, M, C, ff, (hipStream_t)stream);
}

/* vtx_mlp_bwd with the LayerNorm backward of norm_ff folded into its epilogue (round 6, option LN_FOLD): dx1 = dy + LN'(dln2) -- the
 * bits of vtx_mlp_bwd followed by vtx_layernorm_bwd(dln2, x1, mean, rstd, gamma, dres = dy); part: [part_rows][2 C] fp32 dgamma | dbeta
 * partial rows for vtx_colreduce / vtx_colreduce_multi (part_rows >= compute units; rows this launch does not own are zeroed). */
int vtx_mlp_bwd_ln(int dtype, const void* ln2, const void* dy, const void* w1, const float* b1, const void* w2, const float* rowscale,
                   int rows_per_scale, void* h, void* dz, const void* x1, const float* mean, const float* rstd, const float* gamma,
                   void* dx1, float* part, int part_rows, int64_t M, int C, int ff, void* stream) {
  if (dtype != VTX_BF16) return VTX_ERR_DTYPE;
  if (M <= 0 || !mlp_fused_ok(dtype, M > 32768 ? M : 32768, C, ff)) return VTX_ERR_SHAPE;
  return mlp_fused_bwd_ln(ln2, dy, w1, b1, w2, rowscale, rows_per_scale, h, dz, x1, mean, rstd, gamma, dx1, part, part_rows, M, C, ff,
                          (hipStream_t)stream);
}

/* vtx_layernorm_fwd(x1 -> ln2, mean, rstd) + vtx_mlp_fwd(ln2, .., resid = x1) in one launch (round 6, option LN_FOLD bit 2): the same bits. */
int vtx_mlp_fwd_ln(int dtype, const void* x1, const float* gamma, const float* beta, float eps, void* ln2, float* mean, float* rstd,
                   const void* w1, const float* b1, const void* w2, const float* b2, const float* rowscale, int rows_per_scale, void* y,
                   int64_t M, int C, int ff, void* stream) {
  if (dtype != VTX_BF16) return VTX_ERR_DTYPE;
  if (M <= 0 || !mlp_fused_ok(dtype, M > 32768 ? M : 32768, C, ff)) return VTX_ERR_SHAPE;
  return mlp_fused_fwd_ln(x1, gamma, beta, eps, ln2, mean, rstd, w1, b1, w2, b2, rowscale, rows_per_scale, y, M, C, ff, (hipStream_t)stream);
}

}  // extern "C"
