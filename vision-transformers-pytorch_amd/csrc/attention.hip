// Fused multi-head attention cores for gfx950: global (ViT) and (shifted-)window (Swin).
//
// Replaces, per (batch, window, head) problem, the reference's
//   vit.MultiHeadedAttention.forward  models/vit.py:30-42  (reshape/permute, q@k^T/sqrt(d), softmax, @v)
//   swin.MultiHeadedLocalAttention.forward models/swin_transformer.py:109-154 (roll, window
//   partition, q@k^T/sqrt(d) + rel_pos(pos) bias, masked_fill(local_mask,-inf), softmax, @v, inverse
//   partition, roll back)
// reading Q/K/V straight out of the QKV-projection output [rows, 3*h*D] (channel order
// [q|k|v][head][d], vit.py:30-34 / swin:128) and writing O into [rows, h*D]: roll, window
// partition and head split are address arithmetic (SURVEY "A9 semantic specification"), the
// L x L scores never leave registers.
//
// Tiling: 16x16 MFMA tiles via mma16 (vtx_common.h).  Forward and the dQ phase use the
// "swapped" product S^T = K Q^T so that a lane holds, for ONE query, 4 keys of every key tile:
// softmax is lane-local + two __shfl_xor (16, 32), and the accumulators are directly the A
// operand (k-slots = keys) of the P.V / dS.K products.  The dK/dV phase uses S = Q K^T so the
// accumulators are the A operand of the products contracting over queries.  Operands that are
// contracted over their row index (V, K for dQ; Q, dO for dK/dV) are staged TRANSPOSED in LDS.
#include "vtx_common.h"

struct AttnGeom {
  int L;        // tokens per problem (49 / 197 / 37)
  int nH;       // heads
  int hd;       // nH * D
  int nW;       // problems (windows) per image; 1 for global attention
  int swin;     // 0: token i of image b is row b*L + i;  1: window addressing below
  int H, W, win, shift, nWx;
  float scale;  // 1/sqrt(D), applied to the product (vit.py:37, swin:134)
};

__device__ __forceinline__ int64_t attn_token_row(const AttnGeom& g, int b, int n, int i) {
  if (!g.swin) return (int64_t)b * g.L + i;
  const int wi = n / g.nWx, wj = n - wi * g.nWx;
  const int ay = i / g.win, ax = i - ay * g.win;
  int y = wi * g.win + ay + g.shift; if (y >= g.H) y -= g.H;   // rolled position p holds original (p + w/2) mod H
  int x = wj * g.win + ax + g.shift; if (x >= g.W) x -= g.W;
  return ((int64_t)b * g.H + y) * g.W + x;
}

// (every caller passes a clamped, readable address: see load8_clamped)
template <typename T> __device__ __forceinline__ Vec8<T> load8_or_zero(const T* p, bool valid) { return load8_clamped<T>(p, valid); }

// A-operand fragment from two accumulator tiles: k-slot (g, j) <-> rows 4g+j of tile `lo` (j<4) / tile `hi` (j>=4)
template <typename T> __device__ __forceinline__ Vec8<T> frag_from_acc(const f32x4& lo, const f32x4& hi) {
  Vec8<T> f;
#pragma unroll
  for (int j = 0; j < 4; ++j) { f.set(j, lo[j]); f.set(4 + j, hi[j]); }
  return f;
}

// B-operand fragment from a transposed LDS image Xt[d][token]: the same k-slot <-> token map
// (tokens 32*ks + 4g + j and 32*ks + 16 + 4g + j).  `p` points at Xt[d][32*ks].
template <typename T> __device__ __forceinline__ Vec8<T> frag_from_transposed(const T* p, int g) {
  Vec8<T> f;
  if constexpr (sizeof(T) == 2) {
    bf16x4 a = *reinterpret_cast<const bf16x4*>(p + 4 * g);
    bf16x4 b = *reinterpret_cast<const bf16x4*>(p + 16 + 4 * g);
#pragma unroll
    for (int j = 0; j < 4; ++j) { f.v[j] = a[j]; f.v[4 + j] = b[j]; }
  } else {
    f32x4 a = *reinterpret_cast<const f32x4*>(p + 4 * g);
    f32x4 b = *reinterpret_cast<const f32x4*>(p + 16 + 4 * g);
#pragma unroll
    for (int j = 0; j < 4; ++j) { f.v[j] = a[j]; f.v[4 + j] = b[j]; }
  }
  return f;
}

// stage X[token][0..D) (token rows of one problem) transposed into LDS Xt[d][LP + 8]; pad tokens -> 0
template <typename T, int D, int NKT>
__device__ __forceinline__ void stage_transposed(T* __restrict__ xt, const T* __restrict__ src, int64_t ld,
                                                 const AttnGeom& g, int b, int n) {
  constexpr int LP = NKT * 16, STR = LP + 8, DV = D / 8;
  for (int idx = threadIdx.x; idx < LP * DV; idx += blockDim.x) {
    const int tok = idx / DV, dv = idx - tok * DV;
    Vec8<T> v = vec8_zero<T>();
    if (tok < g.L) v = load8<T>(src + attn_token_row(g, b, n, tok) * ld + dv * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      xt[(dv * 8 + e) * STR + tok] = v.v[e];
    }
  }
}

// bias + mask term of score (q, key); returns -inf for masked / padded keys (swin:135-141)
__device__ __forceinline__ float attn_bias_mask(const float* __restrict__ bias, const uint8_t* __restrict__ mask,
                                                int L, int q, int key) {
  if (key >= L) return -INFINITY;
  float v = 0.f;
  if (bias) v = bias[q * L + key];
  if (mask && mask[q * L + key]) v = -INFINITY;
  return v;
}

// ------------------------------------------------------------------------------------- forward
// grid.x = B * nW * nH problems; 256 threads; wave w handles query tiles w, w+4, ...
template <typename T, int D, int NKT, bool DROP = false>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ o,
                                                      float* __restrict__ lse, const float* __restrict__ bias,
                                                      const uint8_t* __restrict__ mask, AttnGeom g, DropArgs da) {
  constexpr int LP = NKT * 16, STR = LP + 8, DS = D / 32, DT = D / 16, KSN = NKT / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char attn_smem[];
  T* vt = reinterpret_cast<T*>(attn_smem);                       // Vt[D][STR]

  const int prob = blockIdx.x;
  const int h = prob % g.nH;
  const int bn = prob / g.nH;
  const int n = bn % g.nW, b = bn / g.nW;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c_ = lane & 15, g_ = lane >> 4;
  const int64_t ld = 3 * (int64_t)g.hd;
  const T* qb = qkv + h * D;
  const T* kb = qkv + g.hd + h * D;
  const T* vb = qkv + 2 * g.hd + h * D;
  const float* bias_h = bias ? bias + (int64_t)h * g.L * g.L : nullptr;
  const uint8_t* mask_n = mask ? mask + (int64_t)n * g.L * g.L : nullptr;

  stage_transposed<T, D, NKT>(vt, vb, ld, g, b, n);
  __syncthreads();

  for (int qt = wave; qt * 16 < g.L; qt += 4) {
    const int q = qt * 16 + c_;
    const bool qv = q < g.L;
    Vec8<T> qf[DS];
    {
      const T* qp = qb + (qv ? attn_token_row(g, b, n, q) : 0) * ld + g_ * 8;
#pragma unroll
      for (int ds = 0; ds < DS; ++ds) qf[ds] = load8_or_zero<T>(qp + ds * 32, qv);
    }
    // S^T tiles: st[kt][r] = S[q = c_][key = 16 kt + 4 g_ + r]
    f32x4 st[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      st[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int key = kt * 16 + c_;
      const bool kv = key < g.L;
      const T* kp = kb + (kv ? attn_token_row(g, b, n, key) : 0) * ld + g_ * 8;
#pragma unroll
      for (int ds = 0; ds < DS; ++ds) {
        Vec8<T> kf = load8_or_zero<T>(kp + ds * 32, kv);
        mma16(kf, qf[ds], st[kt]);
      }
      if constexpr (sizeof(T) == 4 || DROP) acc_settle(st[kt]);
    }
    // softmax over keys (fp32), row = this lane's query
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kt * 16 + g_ * 4 + r;
        const float s = st[kt][r] * g.scale + (qv ? attn_bias_mask(bias_h, mask_n, g.L, q, key)
                                                  : (key < g.L ? 0.f : -INFINITY));
        st[kt][r] = s;
        m = fmaxf(m, s);
      }
    m = fmaxf(m, shfl_xor_f(m, 16));
    m = fmaxf(m, shfl_xor_f(m, 32));
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __expf(st[kt][r] - m);
        st[kt][r] = p;
        l += p;
      }
    l += shfl_xor_f(l, 16);
    l += shfl_xor_f(l, 32);
    const float inv = 1.f / l;
    if (qv && g_ == 0) lse[(int64_t)prob * g.L + q] = m + __logf(l);
    if constexpr (DROP) {
      // dropout acts on the NORMALISED probabilities (vit.py:38-39): the row sum above is the undropped one
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) st[kt][r] *= drop_factor(da, (unsigned)prob, q, kt * 16 + g_ * 4 + r);
    }
    // O[q][d] = sum_key P[q][key] V[key][d]
    f32x4 oacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KSN; ++ks) {
      f32x4 lo = st[2 * ks] * inv, hi = st[2 * ks + 1] * inv;
      Vec8<T> pf = frag_from_acc<T>(lo, hi);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        Vec8<T> vf = frag_from_transposed<T>(vt + (dt * 16 + c_) * STR + ks * 32, g_);
        mma16(pf, vf, oacc[dt]);
      }
      if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) acc_settle(oacc[dt]);
      }
    }
    // oacc[dt][r] = O[q = 16 qt + 4 g_ + r][d = 16 dt + c_]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qo = qt * 16 + g_ * 4 + r;
      if (qo < g.L) {
        T* op = o + attn_token_row(g, b, n, qo) * (int64_t)g.hd + h * D + c_;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) op[dt * 16] = from_f32<T>(oacc[dt][r]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------- backward
// grid = (nblk, nH).  Block (x, h) walks the (b, n) pairs x, x + nblk, ... of head h (so the
// rel-pos-bias gradient accumulates in registers and is written once per block: deterministic).
//   phase A (wave <-> query tile, swapped layout): Dq = rowsum(dO o O), then per key-tile pair P, dP, dS,
//           dQ = scale * dS K
//   phase B (wave <-> key tile,   plain   layout): P, dP, dS, dV = P^T dO, dK = scale * dS^T Q, dbias += dS
template <typename T, int D, int NKT, bool HAS_BIAS, bool DROP = false>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const T* __restrict__ qkv, const T* __restrict__ oin,
                                                      const T* __restrict__ dout, const float* __restrict__ lse, const float* __restrict__ bias,
                                                      const uint8_t* __restrict__ mask, T* __restrict__ dqkv,
                                                      float* __restrict__ dbias_part, int nbn, AttnGeom g, DropArgs da) {
  constexpr int LP = NKT * 16, STR = LP + 8, DS = D / 32, DT = D / 16, KSN = NKT / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char attn_smem[];
  T* r0 = reinterpret_cast<T*>(attn_smem);                 // Kt (phase A) then Qt (phase B): [D][STR]
  T* r1 = r0 + D * STR;                                    // dOt: [D][STR]
  float* dq_s = reinterpret_cast<float*>(r1 + D * STR);    // Dq[LP]
  float* lse_s = dq_s + LP;                                // lse[LP]

  const int h = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c_ = lane & 15, g_ = lane >> 4;
  const int64_t ld = 3 * (int64_t)g.hd;
  const T* qb = qkv + h * D;
  const T* kb = qkv + g.hd + h * D;
  const T* vb = qkv + 2 * g.hd + h * D;
  const T* dob = dout + h * D;
  const T* ob = oin + h * D;
  const float* bias_h = bias ? bias + (int64_t)h * g.L * g.L : nullptr;

  // rel-pos-bias gradient accumulators: wave owns key tiles kt = wave, wave + 4, ... (KPW of them); [key tile of the wave][query tile].
  // One key tile per wave for windows of <= 64 tokens; 12 x 12 windows (NKT = 10, Swin at 384 x 384) hold 3 x 10 tiles: both phase-B
  // loops are then fully unrolled so that every index is a compile-time one.
  constexpr int KPW = (NKT + 3) / 4;
  constexpr bool BIAS_WIDE = HAS_BIAS && NKT > 4;
  f32x4 dsacc[HAS_BIAS ? KPW * NKT : 1];
  if (HAS_BIAS) {
#pragma unroll
    for (int i = 0; i < KPW * NKT; ++i) dsacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  for (int bn = blockIdx.x; bn < nbn; bn += gridDim.x) {
    const int n = bn % g.nW, b = bn / g.nW;
    const int prob = bn * g.nH + h;
    const uint8_t* mask_n = mask ? mask + (int64_t)n * g.L * g.L : nullptr;

    stage_transposed<T, D, NKT>(r0, kb, ld, g, b, n);                     // Kt
    stage_transposed<T, D, NKT>(r1, dob, (int64_t)g.hd, g, b, n);          // dOt
    for (int i = threadIdx.x; i < LP; i += blockDim.x) lse_s[i] = i < g.L ? lse[(int64_t)prob * g.L + i] : 0.f;
    __syncthreads();

    // ---------------- phase A
    for (int qt = wave; qt * 16 < g.L; qt += 4) {
      const int q = qt * 16 + c_;
      const bool qv = q < g.L;
      const int64_t qrow = qv ? attn_token_row(g, b, n, q) : 0;
      Vec8<T> qf[DS], dof[DS];
#pragma unroll
      for (int ds = 0; ds < DS; ++ds) {
        qf[ds] = load8_or_zero<T>(qb + qrow * ld + ds * 32 + g_ * 8, qv);
        dof[ds] = load8_or_zero<T>(dob + qrow * g.hd + ds * 32 + g_ * 8, qv);
      }
      const float lq = lse_s[qv ? q : 0];
      // Dq = rowsum(P o dP) = rowsum(dO o O): lane-local partial over this lane's d-slots + 2 shuffles
      float dsum = 0.f;
#pragma unroll
      for (int ds = 0; ds < DS; ++ds) {
        Vec8<T> of = load8_or_zero<T>(ob + qrow * g.hd + ds * 32 + g_ * 8, qv);
#pragma unroll
        for (int e2 = 0; e2 < 8; ++e2) dsum += of.get(e2) * dof[ds].get(e2);
      }
      dsum += shfl_xor_f(dsum, 16);
      dsum += shfl_xor_f(dsum, 32);
      if (g_ == 0) dq_s[q] = qv ? dsum : 0.f;
      f32x4 dqacc[DT];
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) dqacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll (NKT <= 4 ? KSN : 1)
      for (int ks = 0; ks < KSN; ++ks) {
        f32x4 dsv[2];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int kt = 2 * ks + half;
          f32x4 pt = f32x4{0.f, 0.f, 0.f, 0.f}, dpt = f32x4{0.f, 0.f, 0.f, 0.f};
          const int key = kt * 16 + c_;
          const bool kv = key < g.L;
          const int64_t krow = kv ? attn_token_row(g, b, n, key) : 0;
#pragma unroll
          for (int ds = 0; ds < DS; ++ds) {
            Vec8<T> kf = load8_or_zero<T>(kb + krow * ld + ds * 32 + g_ * 8, kv);
            Vec8<T> vf = load8_or_zero<T>(vb + krow * ld + ds * 32 + g_ * 8, kv);
            mma16(kf, qf[ds], pt);      // pt[r]  = S [q = c_][key = 16 kt + 4 g_ + r]
            mma16(vf, dof[ds], dpt);    // dpt[r] = dP[same]
          }
          if constexpr (sizeof(T) == 4 || DROP) { acc_settle(pt); acc_settle(dpt); }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int kk = kt * 16 + g_ * 4 + r;
            float p = 0.f;
            if (qv) p = __expf(pt[r] * g.scale + attn_bias_mask(bias_h, mask_n, g.L, q, kk) - lq);
            // with dropout O = (P o F) V, F = keep / (1 - p): dP = F o (dO V^T), and rowsum(dO o O) is still rowsum(P o dP)
            float dpv = dpt[r];
            if constexpr (DROP) dpv *= drop_factor(da, (unsigned)prob, q, kk);
            dsv[half][r] = p * (dpv - dsum);
          }
        }
        Vec8<T> dsf = frag_from_acc<T>(dsv[0], dsv[1]);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          Vec8<T> kf = frag_from_transposed<T>(r0 + (dt * 16 + c_) * STR + ks * 32, g_);
          mma16(dsf, kf, dqacc[dt]);
        }
        if constexpr (sizeof(T) == 4) {
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) acc_settle(dqacc[dt]);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qo = qt * 16 + g_ * 4 + r;
        if (qo < g.L) {
          T* p = dqkv + attn_token_row(g, b, n, qo) * ld + h * D + c_;
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) p[dt * 16] = from_f32<T>(dqacc[dt][r] * g.scale);
        }
      }
    }
    __syncthreads();
    stage_transposed<T, D, NKT>(r0, qb, ld, g, b, n);                     // Qt (Kt is dead)
    __syncthreads();

    // ---------------- phase B
#pragma unroll (BIAS_WIDE ? KPW : 1)
    for (int kti = 0; kti < KPW; ++kti) {
      const int kt = wave + 4 * kti;
      if (kt * 16 >= g.L) break;
      const int key = kt * 16 + c_;
      const bool kv = key < g.L;
      const int64_t krow = kv ? attn_token_row(g, b, n, key) : 0;
      Vec8<T> kf[DS], vf[DS];
#pragma unroll
      for (int ds = 0; ds < DS; ++ds) {
        kf[ds] = load8_or_zero<T>(kb + krow * ld + ds * 32 + g_ * 8, kv);
        vf[ds] = load8_or_zero<T>(vb + krow * ld + ds * 32 + g_ * 8, kv);
      }
      f32x4 dkacc[DT], dvacc[DT];
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) { dkacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dvacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll ((NKT <= 4 || BIAS_WIDE) ? KSN : 1)
      for (int qs = 0; qs < KSN; ++qs) {
        f32x4 pp[2], dss[2];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int qt = 2 * qs + half;
          f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
          const int q = qt * 16 + c_;
          const bool qv = q < g.L;
          const int64_t qrow = qv ? attn_token_row(g, b, n, q) : 0;
#pragma unroll
          for (int ds = 0; ds < DS; ++ds) {
            Vec8<T> qf = load8_or_zero<T>(qb + qrow * ld + ds * 32 + g_ * 8, qv);
            Vec8<T> dof = load8_or_zero<T>(dob + qrow * g.hd + ds * 32 + g_ * 8, qv);
            mma16(qf, kf[ds], s);       // s[r]  = S [q = 16 qt + 4 g_ + r][key = 16 kt + c_]
            mma16(dof, vf[ds], dp);     // dp[r] = dP[same]
          }
          if constexpr (sizeof(T) == 4 || DROP) { acc_settle(s); acc_settle(dp); }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int qq = qt * 16 + g_ * 4 + r;
            float p = 0.f;
            if (qq < g.L) p = __expf(s[r] * g.scale + attn_bias_mask(bias_h, mask_n, g.L, qq, key) - lse_s[qq]);
            float f = 1.f;
            if constexpr (DROP) f = drop_factor(da, (unsigned)prob, qq, key);
            pp[half][r] = p * f;                       // dV = (P o F)^T dO
            dss[half][r] = p * (dp[r] * f - dq_s[qq < g.L ? qq : 0]);
          }
          if constexpr (HAS_BIAS) dsacc[kti * NKT + qt] += dss[half];
        }
        Vec8<T> pf = frag_from_acc<T>(pp[0], pp[1]);
        Vec8<T> dsf = frag_from_acc<T>(dss[0], dss[1]);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          Vec8<T> dotf = frag_from_transposed<T>(r1 + (dt * 16 + c_) * STR + qs * 32, g_);
          Vec8<T> qtf = frag_from_transposed<T>(r0 + (dt * 16 + c_) * STR + qs * 32, g_);
          mma16(pf, dotf, dvacc[dt]);
          mma16(dsf, qtf, dkacc[dt]);
        }
        if constexpr (sizeof(T) == 4) {
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) { acc_settle(dvacc[dt]); acc_settle(dkacc[dt]); }
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ko = kt * 16 + g_ * 4 + r;
        if (ko < g.L) {
          T* p = dqkv + attn_token_row(g, b, n, ko) * ld + h * D + c_;
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            p[g.hd + dt * 16] = from_f32<T>(dkacc[dt][r] * g.scale);
            p[2 * g.hd + dt * 16] = from_f32<T>(dvacc[dt][r]);
          }
        }
      }
    }
    __syncthreads();
  }

  if (HAS_BIAS) {
    // dsacc[qt][r] = sum over this block's problems of dS[q = 16 qt + 4 g_ + r][key = 16 wave + c_]
    // per-block slab stride padded to 4 floats (vectorised slab reduce)
    const int64_t slab = ((int64_t)g.nH * g.L * g.L + 3) & ~(int64_t)3;
    float* out = dbias_part + (int64_t)blockIdx.x * slab + (int64_t)h * g.L * g.L;
#pragma unroll
    for (int kti = 0; kti < KPW; ++kti) {
      const int key = (wave + 4 * kti) * 16 + c_;
      if (key < g.L) {
#pragma unroll
        for (int qt = 0; qt < NKT; ++qt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int q = qt * 16 + g_ * 4 + r;
            if (q < g.L) out[q * g.L + key] = dsacc[kti * NKT + qt][r];
          }
      }
    }
  }
}

// bias[h][a][b] = rel_pos[pos[a][b]][h]   (swin:135-136: rel_pos(pos).permute(2,0,1))
__global__ void relpos_bias_kernel(const float* __restrict__ rel_pos, const int64_t* __restrict__ pos,
                                   float* __restrict__ bias, int LL, int nH) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= LL * nH) return;
  const int h = i / LL, ab = i - h * LL;
  bias[i] = rel_pos[pos[ab] * nH + h];
}

// d rel_pos[idx][h] = sum_{(a,b): pos[a][b] == idx} full[h][a][b]  (nn.Embedding dense grad, swin:46,135).
// The (a,b) lists per table index come as a CSR (order / offsets) built once on the host from `pos`;
// fixed summation order -> deterministic (no atomics).
__global__ void relpos_bias_bwd_kernel(const float* __restrict__ full, const int* __restrict__ order,
                                       const int* __restrict__ offsets, float* __restrict__ drel, int LL, int nH,
                                       int ntab) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ntab * nH) return;
  const int idx = i / nH, h = i - idx * nH;
  float s = 0.f;
  for (int k = offsets[idx]; k < offsets[idx + 1]; ++k) s += full[(int64_t)h * LL + order[k]];
  drel[i] = s;
}

static int attn_geom(AttnGeom& g, int L, int nH, int D, int swin, int H, int W, int win, int shift) {
  g.L = L; g.nH = nH; g.hd = nH * D; g.swin = swin; g.H = H; g.W = W; g.win = win;
  g.scale = 1.0f / sqrtf((float)D);
  if (swin) {
    if (win <= 0 || H % win || W % win || L != win * win) return VTX_ERR_SHAPE;
    g.nWx = W / win; g.nW = (H / win) * (W / win); g.shift = shift ? win / 2 : 0;
  } else {
    g.nWx = 1; g.nW = 1; g.shift = 0;
  }
  return VTX_OK;
}

template <typename T, int D, int NKT>
static int attn_fwd_launch(const void* qkv, void* o, float* lse, const float* bias, const uint8_t* mask, int B,
                           const AttnGeom& g, hipStream_t st, const DropArgs* da = nullptr) {
  constexpr size_t smem = (size_t)D * (NKT * 16 + 8) * sizeof(T);
  auto kern = da ? attn_fwd_kernel<T, D, NKT, true> : attn_fwd_kernel<T, D, NKT, false>;
  if (smem > 64 * 1024) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return VTX_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(kern, dim3(B * g.nW * g.nH), dim3(256), smem, st, (const T*)qkv, (T*)o, lse, bias, mask, g,
                     da ? *da : DropArgs{});
  return vtx_check_launch();
}

static int attn_bwd_blocks(int nbn, int nH) {
  int nblk = 2048 / nH;
  if (nblk < 1) nblk = 1;
  if (nblk > nbn) nblk = nbn;
  return nblk;
}

template <typename T, int D, int NKT>
static int attn_bwd_launch(const void* qkv, const void* oin, const void* dout, const float* lse, const float* bias,
                           const uint8_t* mask, void* dqkv, float* part, int B, int nblk, const AttnGeom& g,
                           hipStream_t st, const DropArgs* da = nullptr) {
  constexpr size_t smem = (size_t)2 * D * (NKT * 16 + 8) * sizeof(T) + (size_t)2 * NKT * 16 * sizeof(float);
  const int nbn = B * g.nW;
  const DropArgs dv = da ? *da : DropArgs{};
  if (bias) {
    if (NKT > 10) return VTX_ERR_SHAPE;   // register-resident bias gradient: up to 3 key tiles x 10 query tiles per wave
    auto kern = da ? attn_bwd_kernel<T, D, NKT, true, true> : attn_bwd_kernel<T, D, NKT, true, false>;
    if (smem > 64 * 1024 &&
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return VTX_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(nblk, g.nH), dim3(256), smem, st, (const T*)qkv, (const T*)oin, (const T*)dout, lse, bias, mask,
                       (T*)dqkv, part, nbn, g, dv);
  } else {
    auto kern = da ? attn_bwd_kernel<T, D, NKT, false, true> : attn_bwd_kernel<T, D, NKT, false, false>;
    if (smem > 64 * 1024 &&
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return VTX_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(nbn, g.nH), dim3(256), smem, st, (const T*)qkv, (const T*)oin, (const T*)dout, lse, bias, mask,
                       (T*)dqkv, part, nbn, g, dv);
  }
  return vtx_check_launch();
}

// keep[prob][q][key] = 1 where the hash keeps the cell (what the kernels regenerate): for tests and for feeding an oracle
__global__ void attn_keep_mask_kernel(uint8_t* __restrict__ out, int64_t total, DropArgs da) {
  const int64_t cells = (int64_t)da.Lq * da.Lk;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t prob = i / cells;
    const int cell = (int)(i - prob * cells);
    out[i] = drop_hash(da.s0, da.s1, (unsigned)prob, (unsigned)cell) >= da.thresh ? 1 : 0;
  }
}

#define ATTN_DISPATCH(FN, ...)                                                          \
  do {                                                                                  \
    if (dtype == VTX_BF16) {                                                            \
      if (D == 32 && L <= 64) return FN<bf16, 32, 4>(__VA_ARGS__);                      \
      if (D == 32 && L <= 160) return FN<bf16, 32, 10>(__VA_ARGS__);                    \
      if (D == 64 && L <= 64) return FN<bf16, 64, 4>(__VA_ARGS__);                      \
      if (D == 64 && L <= 224) return FN<bf16, 64, 14>(__VA_ARGS__);                    \
    } else if (dtype == VTX_F32) {                                                      \
      if (D == 32 && L <= 64) return FN<float, 32, 4>(__VA_ARGS__);                     \
      if (D == 32 && L <= 160) return FN<float, 32, 10>(__VA_ARGS__);                   \
      if (D == 64 && L <= 64) return FN<float, 64, 4>(__VA_ARGS__);                     \
      if (D == 64 && L <= 224) return FN<float, 64, 14>(__VA_ARGS__);                   \
    } else {                                                                            \
      return VTX_ERR_DTYPE;                                                             \
    }                                                                                   \
    return VTX_ERR_SHAPE;                                                               \
  } while (0)

// bf16 / head-dim-64 / 64 < L <= 224 global attention: LDS-DMA + transpose-read kernels (attention_seq.hip)
bool sattn_ok(int dtype, int L, int D, int swin, const void* bias);
int sattn_fwd_launch(const void* qkv, void* o, float* lse, int B, int L, int nH, hipStream_t st, const int* perm = nullptr);
int sattn_bwd_launch(const void* qkv, const void* o, const void* dout, const float* lse, void* dqkv, int B, int L, int nH,
                     hipStream_t st, const int* perm = nullptr);

// global attention of any length (attention_long.hip): blocks of 64 keys, online softmax
bool lattn_ok(int dtype, int D);
size_t lattn_bwd_workspace(int B, int L, int nH);
int lattn_fwd_launch(const void* qkv, void* o, float* lse, int B, int L, int nH, int D, int dtype, hipStream_t st,
                     const DropArgs* da = nullptr);
int lattn_bwd_launch(const void* qkv, const void* o, const void* dout, const float* lse, void* dqkv, float* ws, int B,
                     int L, int nH, int D, int dtype, hipStream_t st, const DropArgs* da = nullptr);
static bool attn_is_long(int L, int swin, const void* bias, const void* mask) {
  return !swin && bias == nullptr && mask == nullptr && L > 224;
}

extern "C" {

int vtx_relpos_bias(const float* rel_pos, const int64_t* pos, float* bias, int L, int nH, void* stream) {
  if (!rel_pos || !pos || !bias) return VTX_ERR_NULL;
  const int n = L * L * nH;
  hipLaunchKernelGGL(relpos_bias_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, rel_pos, pos, bias,
                     L * L, nH);
  return vtx_check_launch();
}

static int attention_fwd_impl(const void* qkv, void* o, float* lse, const float* bias, const uint8_t* mask, int B, int L,
                              int nH, int D, int swin, int H, int W, int win, int shift, int dtype, hipStream_t st,
                              const DropArgs* da) {
  if (!qkv || !o || !lse) return VTX_ERR_NULL;
  AttnGeom g;
  int rc = attn_geom(g, L, nH, D, swin, H, W, win, shift);
  if (rc) return rc;
  if (B <= 0) return VTX_OK;
  if (!da && sattn_ok(dtype, L, D, swin, bias)) return sattn_fwd_launch(qkv, o, lse, B, L, nH, st);
  if (attn_is_long(L, swin, bias, mask) && lattn_ok(dtype, D)) return lattn_fwd_launch(qkv, o, lse, B, L, nH, D, dtype, st, da);
  ATTN_DISPATCH(attn_fwd_launch, qkv, o, lse, bias, mask, B, g, st, da);
}

/* bias[h][cell] = table[pos[cell]][h] over any number of cells (halo attention: pos is [W^2][(W + 2A)^2], reference
 * models/halo_transformer.py:95-98) and the dense table gradient from a full [nH][cells] gradient through the CSR of pos. */
int vtx_table_bias(const float* table, const int64_t* pos, float* bias, int64_t cells, int nH, void* stream) {
  if (!table || !pos || !bias) return VTX_ERR_NULL;
  if (cells <= 0 || nH <= 0 || cells * nH > 0x7fffffff) return VTX_ERR_SHAPE;
  const int n = (int)(cells * nH);
  hipLaunchKernelGGL(relpos_bias_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, table, pos, bias, (int)cells, nH);
  return vtx_check_launch();
}
int vtx_table_bias_bwd(const float* full, const int* csr_order, const int* csr_offsets, float* dtable, int64_t cells, int nH, int ntab,
                       void* stream) {
  if (!full || !csr_order || !csr_offsets || !dtable) return VTX_ERR_NULL;
  if (cells <= 0 || nH <= 0 || ntab <= 0 || cells > 0x7fffffff) return VTX_ERR_SHAPE;
  const int n = ntab * nH;
  hipLaunchKernelGGL(relpos_bias_bwd_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, full, csr_order, csr_offsets, dtable,
                     (int)cells, nH, ntab);
  return vtx_check_launch();
}

int vtx_attention_fwd(const void* qkv, void* o, float* lse, const float* bias, const uint8_t* mask, int B, int L,
                      int nH, int D, int swin, int H, int W, int win, int shift, int dtype, void* stream) {
  return attention_fwd_impl(qkv, o, lse, bias, mask, B, L, nH, D, swin, H, W, win, shift, dtype, (hipStream_t)stream, nullptr);
}

/* The same with dropout of the attention probabilities (training mode of the reference's F.dropout(attn, p)): register-resident
 * kernels only (L <= 224 with head dim 64, L <= 64 with head dim 32). */
int vtx_attention_fwd_drop(const void* qkv, void* o, float* lse, const float* bias, const uint8_t* mask, int B, int L,
                           int nH, int D, int swin, int H, int W, int win, int shift, int dtype, float drop_p, uint64_t seed,
                           const uint8_t* keep, void* stream) {
  DropArgs da;
  int rc = drop_args(da, drop_p, seed, keep, L, L);
  if (rc) return rc;
  return attention_fwd_impl(qkv, o, lse, bias, mask, B, L, nH, D, swin, H, W, win, shift, dtype, (hipStream_t)stream, &da);
}

int vtx_attn_keep_mask(uint8_t* out, int64_t nprob, int Lq, int Lk, float drop_p, uint64_t seed, void* stream) {
  if (!out) return VTX_ERR_NULL;
  if (nprob <= 0 || Lq <= 0 || Lk <= 0) return VTX_ERR_SHAPE;
  DropArgs da;
  int rc = drop_args(da, drop_p, seed, nullptr, Lq, Lk);
  if (rc) return rc;
  const int64_t total = nprob * Lq * Lk;
  int64_t nb = (total + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(attn_keep_mask_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, out, total, da);
  return vtx_check_launch();
}

size_t vtx_attention_bwd_workspace(int B, int L, int nH, int swin, int H, int W, int win) {
  if (!swin && L > 224) return lattn_bwd_workspace(B, L, nH);        // Dq = rowsum(dO o O) per query (no bias there)
  const int nW = swin ? (H / win) * (W / win) : 1;
  const int nblk = attn_bwd_blocks(B * nW, nH);
  const size_t slab = ((size_t)nH * L * L + 3) & ~(size_t)3;
  return (size_t)(nblk + 1) * slab * sizeof(float);
}

/* dqkv [rows, 3*h*D] (every element written); drel_pos [(2w-1)^2, nH] fp32 when bias is given. */
static int attention_bwd_impl(const void* qkv, const void* o, const void* dout, const float* lse, const float* bias, const uint8_t* mask,
                              const int* csr_order, const int* csr_offsets, void* dqkv, float* drel_pos, int ntab,
                              void* workspace, size_t ws_bytes,
                              int B, int L, int nH, int D, int swin, int H, int W, int win, int shift, int dtype,
                              hipStream_t st, const DropArgs* da) {
  if (!qkv || !o || !dout || !lse || !dqkv) return VTX_ERR_NULL;
  AttnGeom g;
  int rc = attn_geom(g, L, nH, D, swin, H, W, win, shift);
  if (rc) return rc;
  if (B <= 0) return VTX_OK;
  if (!da && sattn_ok(dtype, L, D, swin, bias)) return sattn_bwd_launch(qkv, o, dout, lse, dqkv, B, L, nH, st);
  if (attn_is_long(L, swin, bias, mask) && lattn_ok(dtype, D)) {
    if (!workspace) return VTX_ERR_NULL;
    if (ws_bytes < lattn_bwd_workspace(B, L, nH)) return VTX_ERR_WORKSPACE;
    return lattn_bwd_launch(qkv, o, dout, lse, dqkv, (float*)workspace, B, L, nH, D, dtype, st, da);
  }
  const int nblk = attn_bwd_blocks(B * g.nW, nH);
  float* part = (float*)workspace;
  if (bias) {
    if (!csr_order || !csr_offsets || !drel_pos || !workspace) return VTX_ERR_NULL;
    if (ws_bytes < vtx_attention_bwd_workspace(B, L, nH, swin, H, W, win)) return VTX_ERR_WORKSPACE;
  }
  auto run = [&]() -> int { ATTN_DISPATCH(attn_bwd_launch, qkv, o, dout, lse, bias, mask, dqkv, part, B, nblk, g, st, da); };
  rc = run();
  if (rc) return rc;
  if (bias) {
    const int64_t nfull = ((int64_t)nH * L * L + 3) & ~(int64_t)3;
    float* full = part + (int64_t)nblk * nfull;
    hipLaunchKernelGGL(slab_reduce_kernel, slab_reduce_grid(nfull), dim3(256), 0, st, (const float*)part, full, nfull,
                       nblk);
    rc = vtx_check_launch();
    if (rc) return rc;
    const int n = ntab * nH;
    hipLaunchKernelGGL(relpos_bias_bwd_kernel, dim3((n + 63) / 64), dim3(64), 0, st, (const float*)full, csr_order,
                       csr_offsets, drel_pos, L * L, nH, ntab);
    rc = vtx_check_launch();
  }
  return rc;
}

int vtx_attention_bwd(const void* qkv, const void* o, const void* dout, const float* lse, const float* bias, const uint8_t* mask,
                      const int* csr_order, const int* csr_offsets, void* dqkv, float* drel_pos, int ntab,
                      void* workspace, size_t ws_bytes,
                      int B, int L, int nH, int D, int swin, int H, int W, int win, int shift, int dtype,
                      void* stream) {
  return attention_bwd_impl(qkv, o, dout, lse, bias, mask, csr_order, csr_offsets, dqkv, drel_pos, ntab, workspace, ws_bytes, B, L, nH,
                            D, swin, H, W, win, shift, dtype, (hipStream_t)stream, nullptr);
}

/* Backward of vtx_attention_fwd_drop: same (drop_p, seed, keep) as the forward; workspace as vtx_attention_bwd with a bias. */
int vtx_attention_bwd_drop(const void* qkv, const void* o, const void* dout, const float* lse, const float* bias, const uint8_t* mask,
                           const int* csr_order, const int* csr_offsets, void* dqkv, float* drel_pos, int ntab,
                           void* workspace, size_t ws_bytes,
                           int B, int L, int nH, int D, int swin, int H, int W, int win, int shift, int dtype,
                           float drop_p, uint64_t seed, const uint8_t* keep, void* stream) {
  DropArgs da;
  int rc = drop_args(da, drop_p, seed, keep, L, L);
  if (rc) return rc;
  return attention_bwd_impl(qkv, o, dout, lse, bias, mask, csr_order, csr_offsets, dqkv, drel_pos, ntab, workspace, ws_bytes, B, L, nH,
                            D, swin, H, W, win, shift, dtype, (hipStream_t)stream, &da);
}

/* Global attention over Bk <= B images only (stochastic-depth compaction, csrc/layer.hip): the b-th image worked on is image
 * perm[b]; lse is indexed by b.  Only where the bf16 fast path applies (head dim 64, L <= 224): VTX_ERR_SHAPE otherwise. */
int vtx_attention_fwd_mapped(const void* qkv, void* o, float* lse, const int* perm, int Bk, int L, int nH, int D, int dtype,
                             void* stream) {
  if (!qkv || !o || !lse || !perm) return VTX_ERR_NULL;
  if (Bk <= 0 || !sattn_ok(dtype, L, D, 0, nullptr)) return VTX_ERR_SHAPE;
  return sattn_fwd_launch(qkv, o, lse, Bk, L, nH, (hipStream_t)stream, perm);
}
int vtx_attention_bwd_mapped(const void* qkv, const void* o, const void* dout, const float* lse, void* dqkv, const int* perm,
                             int Bk, int L, int nH, int D, int dtype, void* stream) {
  if (!qkv || !o || !dout || !lse || !dqkv || !perm) return VTX_ERR_NULL;
  if (Bk <= 0 || !sattn_ok(dtype, L, D, 0, nullptr)) return VTX_ERR_SHAPE;
  return sattn_bwd_launch(qkv, o, dout, lse, dqkv, Bk, L, nH, (hipStream_t)stream, perm);
}

}  // extern "C"
