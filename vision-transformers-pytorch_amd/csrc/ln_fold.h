// LayerNorm backward folded into the epilogue of a row-streaming kernel (round 6; option LN_FOLD).
//
// The row-streaming kernels (mlp_fused.hip, gemm_skinny.hip) end a 32-row block with the block's output in accumulator registers in the
// layout of their row operands: lane (c, g) of a wave holds, for rows (mt, c) = 32 rb + 16 mt + c, the columns 32 tp + 8 g .. + 7 of every
// 32-column block tp -- whole 8-element vectors v = 4 tp + g of the row.  That is the lane-group layout <16 lanes, 1 vector> of the
// stand-alone ln_bwd_kernel (csrc/layernorm.hip) spread over the four lanes c, c + 16, c + 32, c + 48, so the LayerNorm backward
//     dx = dres + rstd (g - mean(g) - xhat mean(g xhat)),   g = dln gamma,   xhat = (x - mean) rstd
// of the tensor the launch was about to store (dln) can run on the accumulators: one more row operand (x, the norm's input), two row
// sums over four lanes, and the dln tensor is neither written nor read back; the stand-alone launch disappears.
//
// Bit-identical dx: the element expressions are the stand-alone kernel's (vtx_common.h ln_bwd_elem_*, same fp contraction), dln goes
// through the bf16 rounding the store would have applied, and the row sums are formed in group_sum<16>'s association: per-vector
// partials in element order, (v0 + v1) + (v2 + v3) per quad, then (q0 + q1) + (q2 + q3) with the quads past C / 32 zero.
// dgamma / dbeta: per-lane register sums over the rows a lane visits, DPP sums over the 16 lanes c of a row (same columns), waves summed
// through LDS in wave order, one partial row per workgroup in the stand-alone launch's workspace layout [rows][2 C] (the rows the launch
// does not own are zeroed for the deferred column reduce): another grouping of the rows than the stand-alone launch, i.e. fp32
// summation order, not bits.
#pragma once
#include "vtx_common.h"

template <int NPR> struct LnFold {           // NPR = C / 32
  float gm[NPR][8], dg[NPR][8], db[NPR][8];

  __device__ __forceinline__ void init(const float* __restrict__ gamma, int g) {
#pragma unroll
    for (int tp = 0; tp < NPR; ++tp) {
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + tp * 32 + 8 * g), g1 = *reinterpret_cast<const f32x4*>(gamma + tp * 32 + 8 * g + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { gm[tp][e] = g0[e]; gm[tp][4 + e] = g1[e]; }
#pragma unroll
      for (int e = 0; e < 8; ++e) { dg[tp][e] = 0.f; db[tp][e] = 0.f; }
    }
  }

  // one row: dl = the row's dln vectors (already rounded to bf16), xv = the norm's input, dres = the gradient that bypasses the norm;
  // live: the row exists (rows past M take part in the shuffles with clamped operands and are neither accumulated nor stored)
  __device__ __forceinline__ void row(const Vec8<bf16> (&dl)[NPR], const bf16x8 (&xv)[NPR], const bf16x8 (&dres)[NPR], float mu, float rs, bool live,
                                      bf16* __restrict__ dst /* dx + row * C + 8 g */) {
    constexpr float invC = 1.f / (float)(32 * NPR);
    float xh[NPR][8], gv[NPR][8], q1[NPR], q2[NPR];
#pragma unroll
    for (int tp = 0; tp < NPR; ++tp) {
      Vec8<bf16> x8;
      x8.v = xv[tp];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = dl[tp].get(e);
        ln_bwd_elem_accum(d, x8.get(e), mu, rs, gm[tp][e], s1, s2, xh[tp][e], gv[tp][e]);
        if (live) { dg[tp][e] += d * xh[tp][e]; db[tp][e] += d; }
      }
      // vector v = 4 tp + g of the row: pairs (g, g ^ 1), then the quad -- the first two steps of group_sum<16>
      s1 += shfl_xor_f(s1, 16);
      s2 += shfl_xor_f(s2, 16);
      q1[tp] = s1 + shfl_xor_f(s1, 32);
      q2[tp] = s2 + shfl_xor_f(s2, 32);
    }
    // 8-lane halves (q0 + q1), (q2 + q3), then their sum; the quads past C / 32 hold zeros in the stand-alone kernel
    float c1, c2;
    static_assert(NPR >= 2 && NPR <= 4, "C = 64, 96 or 128");
    if constexpr (NPR == 2) { c1 = ((q1[0] + q1[1]) + 0.f) * invC; c2 = ((q2[0] + q2[1]) + 0.f) * invC; }
    else if constexpr (NPR == 3) { c1 = ((q1[0] + q1[1]) + (q1[2] + 0.f)) * invC; c2 = ((q2[0] + q2[1]) + (q2[2] + 0.f)) * invC; }
    else { c1 = ((q1[0] + q1[1]) + (q1[2] + q1[NPR - 1])) * invC; c2 = ((q2[0] + q2[1]) + (q2[2] + q2[NPR - 1])) * invC; }
    if (!live) return;
#pragma unroll
    for (int tp = 0; tp < NPR; ++tp) {
      Vec8<bf16> dv, o;
      dv.v = dres[tp];
#pragma unroll
      for (int e = 0; e < 8; ++e) o.set(e, ln_bwd_elem_out(dv.get(e), rs, gv[tp][e], c1, xh[tp][e], c2));
      store8<bf16>(dst + tp * 32, o);
    }
  }

  // end of the kernel, every thread of the workgroup: `red` = WAVES x 2 C floats of LDS nobody else uses any more
  template <int WAVES> __device__ __forceinline__ void finish(float* red, float* __restrict__ part, int part_rows, int wave, int c, int g) {
    constexpr int C = 32 * NPR;
    __syncthreads();
#pragma unroll
    for (int tp = 0; tp < NPR; ++tp)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float a = group_sum<16>(dg[tp][e]), b = group_sum<16>(db[tp][e]);
        if (c == 0) { red[wave * 2 * C + tp * 32 + 8 * g + e] = a; red[wave * 2 * C + C + tp * 32 + 8 * g + e] = b; }
      }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 64 * WAVES) {
      float a = 0.f;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) a += red[w * 2 * C + i];
      part[(int64_t)blockIdx.x * 2 * C + i] = a;
      for (int r = blockIdx.x + gridDim.x; r < part_rows; r += gridDim.x) part[(int64_t)r * 2 * C + i] = 0.f;
    }
  }
};

// ---- LayerNorm FORWARD folded into the row-streaming kernel that consumes the normalised rows (option LN_FOLD bits 2, 3).
// The kernel requests the RAW rows x in its operand layout (lane (c, g): vectors v = 4 tp + g of rows (mt, c)), normalises them in
// registers -- these ARE the MFMA operands of its first product -- and stores the normalised rows and the statistics on the side
// (the backward and the weight gradient read them): the stand-alone launch and its read of x disappear.
// Bit-identical y, mean, rstd: ln_fwd_kernel<bf16, 16, 1>'s expressions (vtx_common.h ln_fwd_elem_*) and group_sum<16>'s association
// (per-vector partials in element order, quads, (q0 + q1) + (q2 + q3) with zero quads past C / 32), two passes over the registers.
template <int NPR> struct LnFwdFold {
  // gamma / beta: [C] fp32, global or LDS; xv: the row's raw vectors; -> yv (bf16), mu, rs.  Every lane of the four-lane group gets mu / rs.
  static __device__ __forceinline__ void row(const bf16x8 (&xv)[NPR], const float* gamma, const float* beta, float eps, int g,
                                             bf16x8 (&yv)[NPR], float& mu, float& rs) {
    static_assert(NPR >= 2 && NPR <= 4, "C = 64, 96 or 128");
    constexpr float invC = 1.f / (float)(32 * NPR);
    float x[NPR][8], qs[NPR];
#pragma unroll
    for (int tp = 0; tp < NPR; ++tp) {
      Vec8<bf16> t;
      t.v = xv[tp];
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) { x[tp][e] = t.get(e); s += x[tp][e]; }
      s += shfl_xor_f(s, 16);
      qs[tp] = s + shfl_xor_f(s, 32);
    }
    if constexpr (NPR == 2) mu = ((qs[0] + qs[1]) + 0.f) * invC;
    else if constexpr (NPR == 3) mu = ((qs[0] + qs[1]) + (qs[2] + 0.f)) * invC;
    else mu = ((qs[0] + qs[1]) + (qs[2] + qs[NPR - 1])) * invC;
#pragma unroll
    for (int tp = 0; tp < NPR; ++tp) {
      float q = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) ln_fwd_elem_sq(x[tp][e], mu, q);
      q += shfl_xor_f(q, 16);
      qs[tp] = q + shfl_xor_f(q, 32);
    }
    float var;
    if constexpr (NPR == 2) var = ((qs[0] + qs[1]) + 0.f);
    else if constexpr (NPR == 3) var = ((qs[0] + qs[1]) + (qs[2] + 0.f));
    else var = ((qs[0] + qs[1]) + (qs[2] + qs[NPR - 1]));
    rs = rsqrtf(var * invC + eps);
#pragma unroll
    for (int tp = 0; tp < NPR; ++tp) {
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + tp * 32 + 8 * g), g1 = *reinterpret_cast<const f32x4*>(gamma + tp * 32 + 8 * g + 4);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(beta + tp * 32 + 8 * g), b1 = *reinterpret_cast<const f32x4*>(beta + tp * 32 + 8 * g + 4);
      Vec8<bf16> o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o.set(e, ln_fwd_elem_out(x[tp][e], mu, rs, g0[e], b0[e]));
        o.set(4 + e, ln_fwd_elem_out(x[tp][4 + e], mu, rs, g1[e], b1[e]));
      }
      yv[tp] = o.v;
    }
  }
};
