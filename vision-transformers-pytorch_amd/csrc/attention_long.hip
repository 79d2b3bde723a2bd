// Global attention for ANY sequence length (head dim 64 or 32): the ViT path beyond the 224 tokens the register-resident
// kernels of attention.hip / attention_seq.hip hold -- e.g. ViT-S/16 at 384 x 384 = 577 tokens, the fine-tuning resolution
// whose position-embedding interpolation the reference implements at models/vit.py:153-175.
//
// Same math as vit.MultiHeadedAttention.forward (models/vit.py:30-42: (q k^T) / sqrt(d), softmax over keys, @ v) on the
// QKV-projection output [B*L, 3*h*D] (channel order [q|k|v][head][d]), with the keys walked in blocks of 64 and an
// online softmax (running row maximum m and denominator l; the output accumulators are rescaled by exp(m_old - m_new)
// when the maximum moves), so nothing of size L x L exists anywhere.  The backward recomputes P from the saved
// log-sum-exp in two launches: dQ per query tile over all key blocks, dK / dV per key tile over all query blocks
// (every output element is owned by exactly one wave: deterministic, no atomics).
//
// Tiling as in attention.hip (16 x 16 MFMA tiles via mma16): forward and dQ use the swapped product S^T = K Q^T so that
// a lane holds, for ONE query, 4 keys of every key tile (softmax statistics are lane-local + two shuffles) and the
// score tiles are directly the A operand of P V / dS K; operands contracted over tokens (V, K for dQ; Q, dO for dK / dV)
// are staged transposed in LDS one 64-token block at a time.
#include "vtx_common.h"

#define LA_KB 64                 // tokens per block (4 tiles of 16)
#define LA_STR (LA_KB + 8)       // transposed LDS row stride

// Queries and keys may come from different tensors (round 5): Lq queries of row stride ldq against Lk keys / values of row stride
// ldkv -- the packed QKV projection (Lq = Lk, ldq = ldkv = 3 hd) or the q | kv pair of the sub-sampled attention of PVT / Twins-SVT
// with MORE than 64 reduced keys (models/pvt.py:38-66 at 384 x 384: 144 keys, 145 in stage 4; csrc/attention_sr.hip holds <= 64).
struct LongGeom {
  int Lq, Lk, nH, hd;
  int64_t ldq, ldkv;
  float scale;
  const float* bias;      // [nH][Lq][Lk] added to the scaled scores (halo attention's relative-position term), or NULL
  DropArgs drop;          // dropout of the attention probabilities (vtx_common.h); drop.scale == 0: off.  problem = image * nH + head
};
__device__ __forceinline__ float la_drop(const LongGeom& g, int bh, int q, int key) {
  return g.drop.scale == 0.f ? 1.f : drop_factor(g.drop, (unsigned)bh, q, key);
}

// (every caller passes a clamped, readable address: see load8_clamped)
template <typename T> __device__ __forceinline__ Vec8<T> la_load(const T* p, bool valid) { return load8_clamped<T>(p, valid); }
template <typename T> __device__ __forceinline__ Vec8<T> la_frag_acc(const f32x4& lo, const f32x4& hi) {
  Vec8<T> f;
#pragma unroll
  for (int j = 0; j < 4; ++j) { f.set(j, lo[j]); f.set(4 + j, hi[j]); }
  return f;
}
template <typename T> __device__ __forceinline__ Vec8<T> la_frag_t(const T* p, int g) {   // p -> Xt[d][32 * ks]
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
// X[tok0 .. tok0 + 64)[0..D) of image b (row stride ld) -> Xt[d][LA_STR], tokens >= L as zeros
template <typename T, int D>
__device__ __forceinline__ void la_stage_t(T* __restrict__ xt, const T* __restrict__ src, int64_t ld, int64_t row0,
                                           int tok0, int L) {
  constexpr int DV = D / 8;
  for (int idx = threadIdx.x; idx < LA_KB * DV; idx += blockDim.x) {
    const int t = idx / DV, dv = idx - t * DV;
    Vec8<T> v = vec8_zero<T>();
    if (tok0 + t < L) v = load8<T>(src + (row0 + tok0 + t) * ld + dv * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) xt[(dv * 8 + e) * LA_STR + t] = v.v[e];
  }
}

// ------------------------------------------------------------------------------------------------- forward
// grid = (ceil(L / 64), B * nH); wave w of a block owns query tile 4 blockIdx.x + w
template <typename T, int D>
__global__ __launch_bounds__(256) void lattn_fwd_kernel(const T* __restrict__ qp, const T* __restrict__ kp_, const T* __restrict__ vp,
                                                       T* __restrict__ o, float* __restrict__ lse, LongGeom g) {
  constexpr int DS = D / 32, DT = D / 16;
  __shared__ __attribute__((aligned(16))) T vt[D * LA_STR];
  const int bh = blockIdx.y, b = bh / g.nH, h = bh - b * g.nH;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c_ = lane & 15, g_ = lane >> 4;
  const int64_t ldq = g.ldq, ld = g.ldkv, qrow0 = (int64_t)b * g.Lq, row0 = (int64_t)b * g.Lk;
  const T* qb = qp + h * D;
  const T* kb = kp_ + h * D;
  const T* vb = vp + h * D;
  const int qt = blockIdx.x * 4 + wave;
  const int q = qt * 16 + c_;
  const bool qv = q < g.Lq;
  Vec8<T> qf[DS];
#pragma unroll
  for (int ds = 0; ds < DS; ++ds) qf[ds] = la_load<T>(qb + (qrow0 + (qv ? q : 0)) * ldq + ds * 32 + g_ * 8, qv);
  float m_run = -INFINITY, l_run = 0.f;
  f32x4 oacc[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int k0 = 0; k0 < g.Lk; k0 += LA_KB) {
    __syncthreads();                                           // the previous block's Vt readers are done
    la_stage_t<T, D>(vt, vb, ld, row0, k0, g.Lk);
    __syncthreads();
    f32x4 st[4];
    float mb = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      st[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int key = k0 + kt * 16 + c_;
      const bool kv = key < g.Lk;
      const T* kp = kb + (row0 + (kv ? key : 0)) * ld + g_ * 8;
#pragma unroll
      for (int ds = 0; ds < DS; ++ds) mma16(la_load<T>(kp + ds * 32, kv), qf[ds], st[kt]);   // S^T[key][q = c_]
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kk = k0 + kt * 16 + g_ * 4 + r;
        float sv = kk < g.Lk ? st[kt][r] * g.scale : -INFINITY;
        if (g.bias != nullptr && kk < g.Lk && qv) sv += g.bias[((int64_t)h * g.Lq + q) * g.Lk + kk];
        st[kt][r] = sv;
        mb = fmaxf(mb, sv);
      }
    }
    mb = fmaxf(mb, shfl_xor_f(mb, 16));
    mb = fmaxf(mb, shfl_xor_f(mb, 32));
    const float m_new = fmaxf(m_run, mb);                      // finite: every block holds at least one real key
    const float alpha = __expf(m_run - m_new);                 // exp(-inf) = 0 on the first block
    float lb = 0.f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __expf(st[kt][r] - m_new);
        st[kt][r] = p;
        lb += p;
      }
    lb += shfl_xor_f(lb, 16);
    lb += shfl_xor_f(lb, 32);
    l_run = l_run * alpha + lb;
    m_run = m_new;
    // the accumulators hold queries 4 g_ + r (not c_): fetch their rescale factors from the lanes that own them
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float a = shfl_f(alpha, 4 * g_ + r);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) oacc[dt][r] *= a;
    }
    if (g.drop.scale != 0.f) {                              // (wave-uniform) F.dropout on the probabilities: scaled keep factors
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) st[kt][r] *= drop_factor(g.drop, (unsigned)bh, q, k0 + kt * 16 + g_ * 4 + r);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      Vec8<T> pf = la_frag_acc<T>(st[2 * ks], st[2 * ks + 1]);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) mma16(pf, la_frag_t<T>(vt + (dt * 16 + c_) * LA_STR + ks * 32, g_), oacc[dt]);
    }
    if constexpr (sizeof(T) == 4) {                  // (fp32 parity mode: see acc_settle)
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) acc_settle(oacc[dt]);
    }
  }
  if (qv && g_ == 0) lse[(int64_t)bh * g.Lq + q] = m_run + __logf(l_run);
  const float inv = 1.f / l_run;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float iv = shfl_f(inv, 4 * g_ + r);
    const int qo = qt * 16 + g_ * 4 + r;
    if (qo < g.Lq) {
      T* op = o + (qrow0 + qo) * (int64_t)g.hd + h * D + c_;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) op[dt * 16] = from_f32<T>(oacc[dt][r] * iv);
    }
  }
}

// ------------------------------------------------------------------------------------------------- backward: dQ
// grid = (ceil(L / 64), B * nH); wave <-> query tile; also writes Dq[q] = rowsum(dO o O) for the dK / dV launch
template <typename T, int D>
__global__ __launch_bounds__(256) void lattn_bwd_dq_kernel(const T* __restrict__ qp, const T* __restrict__ kp_, const T* __restrict__ vp,
                                                          const T* __restrict__ oin,
                                                          const T* __restrict__ dout, const float* __restrict__ lse,
                                                          T* __restrict__ dq, float* __restrict__ dsum_out, LongGeom g) {
  constexpr int DS = D / 32, DT = D / 16;
  __shared__ __attribute__((aligned(16))) T kt_s[D * LA_STR];
  const int bh = blockIdx.y, b = bh / g.nH, h = bh - b * g.nH;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c_ = lane & 15, g_ = lane >> 4;
  const int64_t ldq = g.ldq, ld = g.ldkv, qrow0 = (int64_t)b * g.Lq, row0 = (int64_t)b * g.Lk;
  const T* qb = qp + h * D;
  const T* kb = kp_ + h * D;
  const T* vb = vp + h * D;
  const int qt = blockIdx.x * 4 + wave;
  const int q = qt * 16 + c_;
  const bool qv = q < g.Lq;
  const int64_t qrow = qrow0 + (qv ? q : 0);
  Vec8<T> qf[DS], dof[DS];
  float dsum = 0.f;
#pragma unroll
  for (int ds = 0; ds < DS; ++ds) {
    qf[ds] = la_load<T>(qb + qrow * ldq + ds * 32 + g_ * 8, qv);
    dof[ds] = la_load<T>(dout + qrow * g.hd + h * D + ds * 32 + g_ * 8, qv);
    Vec8<T> of = la_load<T>(oin + qrow * g.hd + h * D + ds * 32 + g_ * 8, qv);
#pragma unroll
    for (int e = 0; e < 8; ++e) dsum += of.get(e) * dof[ds].get(e);
  }
  dsum += shfl_xor_f(dsum, 16);
  dsum += shfl_xor_f(dsum, 32);
  const float lq = qv ? lse[(int64_t)bh * g.Lq + q] : INFINITY;          // padded query rows: exp(. - inf) = 0
  if (qv && g_ == 0) dsum_out[(int64_t)bh * g.Lq + q] = dsum;
  f32x4 dqacc[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) dqacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int k0 = 0; k0 < g.Lk; k0 += LA_KB) {
    __syncthreads();
    la_stage_t<T, D>(kt_s, kb, ld, row0, k0, g.Lk);
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      f32x4 dsv[2];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int kt = 2 * ks + half;
        f32x4 pt = f32x4{0.f, 0.f, 0.f, 0.f}, dpt = f32x4{0.f, 0.f, 0.f, 0.f};
        const int key = k0 + kt * 16 + c_;
        const bool kv = key < g.Lk;
        const int64_t krow = row0 + (kv ? key : 0);
#pragma unroll
        for (int ds = 0; ds < DS; ++ds) {
          mma16(la_load<T>(kb + krow * ld + ds * 32 + g_ * 8, kv), qf[ds], pt);      // S^T [key][q = c_]
          mma16(la_load<T>(vb + krow * ld + ds * 32 + g_ * 8, kv), dof[ds], dpt);    // dP^T
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int kk = k0 + kt * 16 + g_ * 4 + r;
          float bb = 0.f;
          if (g.bias != nullptr && kk < g.Lk && qv) bb = g.bias[((int64_t)h * g.Lq + q) * g.Lk + kk];
          const float p = kk < g.Lk ? __expf(pt[r] * g.scale + bb - lq) : 0.f;
          dsv[half][r] = p * (dpt[r] * la_drop(g, bh, q, kk) - dsum);
        }
      }
      Vec8<T> dsf = la_frag_acc<T>(dsv[0], dsv[1]);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) mma16(dsf, la_frag_t<T>(kt_s + (dt * 16 + c_) * LA_STR + ks * 32, g_), dqacc[dt]);
    }
    if constexpr (sizeof(T) == 4) {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) acc_settle(dqacc[dt]);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int qo = qt * 16 + g_ * 4 + r;
    if (qo < g.Lq) {
      T* p = dq + (qrow0 + qo) * ldq + h * D + c_;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) p[dt * 16] = from_f32<T>(dqacc[dt][r] * g.scale);
    }
  }
}

// ------------------------------------------------------------------------------------------------- backward: dK, dV
// grid = (ceil(L / 64), B * nH); wave <-> key tile; walks the queries in blocks of 64 (Qt, dOt staged transposed)
template <typename T, int D>
__global__ __launch_bounds__(256) void lattn_bwd_dkv_kernel(const T* __restrict__ qp, const T* __restrict__ kp_, const T* __restrict__ vp,
                                                           const T* __restrict__ dout,
                                                           const float* __restrict__ lse, const float* __restrict__ dsum_in,
                                                           T* __restrict__ dk, T* __restrict__ dv, LongGeom g) {
  constexpr int DS = D / 32, DT = D / 16;
  __shared__ __attribute__((aligned(16))) T qt_s[D * LA_STR];
  __shared__ __attribute__((aligned(16))) T dot_s[D * LA_STR];
  __shared__ float lse_s[LA_KB], dsum_s[LA_KB];
  const int bh = blockIdx.y, b = bh / g.nH, h = bh - b * g.nH;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c_ = lane & 15, g_ = lane >> 4;
  const int64_t ldq = g.ldq, ld = g.ldkv, qrow0 = (int64_t)b * g.Lq, row0 = (int64_t)b * g.Lk;
  const T* qb = qp + h * D;
  const T* kb = kp_ + h * D;
  const T* vb = vp + h * D;
  const T* dob = dout + h * D;
  const int kt = blockIdx.x * 4 + wave;
  const int key = kt * 16 + c_;
  const bool kv = key < g.Lk;
  const int64_t krow = row0 + (kv ? key : 0);
  Vec8<T> kf[DS], vf[DS];
#pragma unroll
  for (int ds = 0; ds < DS; ++ds) {
    kf[ds] = la_load<T>(kb + krow * ld + ds * 32 + g_ * 8, kv);
    vf[ds] = la_load<T>(vb + krow * ld + ds * 32 + g_ * 8, kv);
  }
  f32x4 dkacc[DT], dvacc[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) { dkacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dvacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  for (int q0 = 0; q0 < g.Lq; q0 += LA_KB) {
    __syncthreads();
    la_stage_t<T, D>(qt_s, qb, ldq, qrow0, q0, g.Lq);
    la_stage_t<T, D>(dot_s, dob, (int64_t)g.hd, qrow0, q0, g.Lq);
    if (threadIdx.x < LA_KB) {
      const int qq = q0 + threadIdx.x;
      lse_s[threadIdx.x] = qq < g.Lq ? lse[(int64_t)bh * g.Lq + qq] : INFINITY;      // padded queries: p = 0
      dsum_s[threadIdx.x] = qq < g.Lq ? dsum_in[(int64_t)bh * g.Lq + qq] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int qs = 0; qs < 2; ++qs) {
      f32x4 pp[2], dss[2];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int qtl = 2 * qs + half;
        f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
        const int q = q0 + qtl * 16 + c_;
        const bool qv = q < g.Lq;
        const int64_t qrow = qrow0 + (qv ? q : 0);
#pragma unroll
        for (int ds = 0; ds < DS; ++ds) {
          mma16(la_load<T>(qb + qrow * ldq + ds * 32 + g_ * 8, qv), kf[ds], s);           // S [q][key = c_]
          mma16(la_load<T>(dob + qrow * g.hd + ds * 32 + g_ * 8, qv), vf[ds], dp);       // dP
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ql = qtl * 16 + g_ * 4 + r;
          float bb = 0.f;
          if (g.bias != nullptr && kv && q0 + ql < g.Lq) bb = g.bias[((int64_t)h * g.Lq + q0 + ql) * g.Lk + key];
          const float p = kv ? __expf(s[r] * g.scale + bb - lse_s[ql]) : 0.f;
          const float f = la_drop(g, bh, q0 + ql, key);
          pp[half][r] = p * f;
          dss[half][r] = p * (dp[r] * f - dsum_s[ql]);
        }
      }
      Vec8<T> pf = la_frag_acc<T>(pp[0], pp[1]);
      Vec8<T> dsf = la_frag_acc<T>(dss[0], dss[1]);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        mma16(pf, la_frag_t<T>(dot_s + (dt * 16 + c_) * LA_STR + qs * 32, g_), dvacc[dt]);
        mma16(dsf, la_frag_t<T>(qt_s + (dt * 16 + c_) * LA_STR + qs * 32, g_), dkacc[dt]);
      }
    }
    if constexpr (sizeof(T) == 4) {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) { acc_settle(dvacc[dt]); acc_settle(dkacc[dt]); }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int ko = kt * 16 + g_ * 4 + r;
    if (ko < g.Lk) {
      const int64_t off = (row0 + ko) * ld + h * D + c_;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        dk[off + dt * 16] = from_f32<T>(dkacc[dt][r] * g.scale);
        dv[off + dt * 16] = from_f32<T>(dvacc[dt][r]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------- backward: d bias
// dbias[h][q][key] = sum over the B problems of dS[b, h, q, key] (the relative-position term of halo attention is shared by every
// window of every image).  grid = (key tiles of 16, query blocks of 64, nH); wave <-> query tile of the block; every workgroup walks
// ALL B problems in order and owns its output tile: deterministic, no atomics, no B x Lq x Lk intermediate.
template <typename T, int D>
__global__ __launch_bounds__(256) void lattn_bwd_dbias_kernel(const T* __restrict__ qp, const T* __restrict__ kp_, const T* __restrict__ vp,
                                                             const T* __restrict__ dout, const float* __restrict__ lse,
                                                             const float* __restrict__ dsum_in, float* __restrict__ dbias, int B,
                                                             LongGeom g) {
  constexpr int DS = D / 32;
  const int h = blockIdx.z;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c_ = lane & 15, g_ = lane >> 4;
  const int key = blockIdx.x * 16 + c_;
  const bool kv = key < g.Lk;
  const int qt = blockIdx.y * 4 + wave;
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  if (qt * 16 < g.Lq) {
    const int q = qt * 16 + c_;
    const bool qv = q < g.Lq;
    for (int b = 0; b < B; ++b) {
      const int bh = b * g.nH + h;
      const int64_t qrow = (int64_t)b * g.Lq + (qv ? q : 0), krow = (int64_t)b * g.Lk + (kv ? key : 0);
      f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ds = 0; ds < DS; ++ds) {
        const Vec8<T> kf = la_load<T>(kp_ + krow * g.ldkv + h * D + ds * 32 + g_ * 8, kv);
        const Vec8<T> vf = la_load<T>(vp + krow * g.ldkv + h * D + ds * 32 + g_ * 8, kv);
        mma16(la_load<T>(qp + qrow * g.ldq + h * D + ds * 32 + g_ * 8, qv), kf, s);          // S [q = 16 qt + 4 g + r][key = c]
        mma16(la_load<T>(dout + qrow * g.hd + h * D + ds * 32 + g_ * 8, qv), vf, dp);       // dP
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qq = qt * 16 + g_ * 4 + r;
        if (qq < g.Lq && kv) {
          const float bb = g.bias[((int64_t)h * g.Lq + qq) * g.Lk + key];
          const float p = __expf(s[r] * g.scale + bb - lse[(int64_t)bh * g.Lq + qq]);
          acc[r] += p * (dp[r] * la_drop(g, bh, qq, key) - dsum_in[(int64_t)bh * g.Lq + qq]);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qq = qt * 16 + g_ * 4 + r;
      if (qq < g.Lq && kv) dbias[((int64_t)h * g.Lq + qq) * g.Lk + key] = acc[r];
    }
  }
}

// ------------------------------------------------------------------------------------------------- host
bool lattn_ok(int dtype, int D) { return (dtype == VTX_BF16 || dtype == VTX_F32) && (D == 64 || D == 32); }
size_t lattn_bwd_workspace(int B, int L, int nH) { return (size_t)B * nH * L * sizeof(float); }

template <typename T, int D>
static int lattn_fwd_t(const void* q, const void* k, const void* v, void* o, float* lse, int B, const LongGeom& g, hipStream_t st) {
  hipLaunchKernelGGL((lattn_fwd_kernel<T, D>), dim3((g.Lq + 63) / 64, B * g.nH), dim3(256), 0, st, (const T*)q, (const T*)k, (const T*)v,
                     (T*)o, lse, g);
  return vtx_check_launch();
}
template <typename T, int D>
static int lattn_bwd_t(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse, void* dq, void* dk,
                       void* dv, float* ws, int B, const LongGeom& g, hipStream_t st) {
  hipLaunchKernelGGL((lattn_bwd_dq_kernel<T, D>), dim3((g.Lq + 63) / 64, B * g.nH), dim3(256), 0, st, (const T*)q, (const T*)k,
                     (const T*)v, (const T*)o, (const T*)dout, lse, (T*)dq, ws, g);
  int rc = vtx_check_launch();
  if (rc) return rc;
  hipLaunchKernelGGL((lattn_bwd_dkv_kernel<T, D>), dim3((g.Lk + 63) / 64, B * g.nH), dim3(256), 0, st, (const T*)q, (const T*)k,
                     (const T*)v, (const T*)dout, lse, (const float*)ws, (T*)dk, (T*)dv, g);
  return vtx_check_launch();
}
template <typename T> static const void* la_off(const void* p, int64_t n) { return (const T*)p + n; }
template <typename T> static void* la_offw(void* p, int64_t n) { return (T*)p + n; }

#define LA_DISPATCH(FN, ...)                                                                                          \
  (dtype == VTX_BF16 ? (D == 64 ? FN<bf16, 64>(__VA_ARGS__) : FN<bf16, 32>(__VA_ARGS__))                              \
                     : (D == 64 ? FN<float, 64>(__VA_ARGS__) : FN<float, 32>(__VA_ARGS__)))

// packed QKV projection [B*L, 3 hd]
int lattn_fwd_launch(const void* qkv, void* o, float* lse, int B, int L, int nH, int D, int dtype, hipStream_t st, const DropArgs* da) {
  const int hd = nH * D;
  LongGeom g{L, L, nH, hd, 3 * (int64_t)hd, 3 * (int64_t)hd, 1.0f / sqrtf((float)D), nullptr, da ? *da : DropArgs{}};
  const size_t es = dtype == VTX_BF16 ? 2 : 4;
  const char* base = (const char*)qkv;
  return LA_DISPATCH(lattn_fwd_t, base, base + es * hd, base + 2 * es * hd, o, lse, B, g, st);
}
int lattn_bwd_launch(const void* qkv, const void* o, const void* dout, const float* lse, void* dqkv, float* ws, int B,
                     int L, int nH, int D, int dtype, hipStream_t st, const DropArgs* da) {
  const int hd = nH * D;
  LongGeom g{L, L, nH, hd, 3 * (int64_t)hd, 3 * (int64_t)hd, 1.0f / sqrtf((float)D), nullptr, da ? *da : DropArgs{}};
  const size_t es = dtype == VTX_BF16 ? 2 : 4;
  const char* base = (const char*)qkv;
  char* out = (char*)dqkv;
  return LA_DISPATCH(lattn_bwd_t, base, base + es * hd, base + 2 * es * hd, o, dout, lse, out, out + es * hd, out + 2 * es * hd, ws, B, g, st);
}
// q [B*Lq, hd] against kv [B*Lk, 2 hd] (k | v halves): the sub-sampled attention of PVT / Twins-SVT with any number of keys
int lattn_cross_fwd_launch(const void* q, const void* kv, void* o, float* lse, int B, int Lq, int Lk, int nH, int D, int dtype,
                           hipStream_t st, const float* bias, const DropArgs* da) {
  const int hd = nH * D;
  LongGeom g{Lq, Lk, nH, hd, (int64_t)hd, 2 * (int64_t)hd, 1.0f / sqrtf((float)D), bias, da ? *da : DropArgs{}};
  const size_t es = dtype == VTX_BF16 ? 2 : 4;
  const char* kb = (const char*)kv;
  return LA_DISPATCH(lattn_fwd_t, q, kb, kb + es * hd, o, lse, B, g, st);
}
template <typename T, int D>
static int lattn_dbias_t(const void* q, const void* k, const void* v, const void* dout, const float* lse, const float* ws, float* dbias,
                         int B, const LongGeom& g, hipStream_t st) {
  hipLaunchKernelGGL((lattn_bwd_dbias_kernel<T, D>), dim3((g.Lk + 15) / 16, (g.Lq + 63) / 64, g.nH), dim3(256), 0, st, (const T*)q,
                     (const T*)k, (const T*)v, (const T*)dout, lse, ws, dbias, B, g);
  return vtx_check_launch();
}
// dbias [nH][Lq][Lk] fp32 (NULL without a bias): summed over the B problems in fixed order
int lattn_cross_bwd_launch(const void* q, const void* kv, const void* o, const void* dout, const float* lse, void* dq, void* dkv,
                           float* ws, int B, int Lq, int Lk, int nH, int D, int dtype, hipStream_t st, const float* bias,
                           float* dbias, const DropArgs* da) {
  const int hd = nH * D;
  LongGeom g{Lq, Lk, nH, hd, (int64_t)hd, 2 * (int64_t)hd, 1.0f / sqrtf((float)D), bias, da ? *da : DropArgs{}};
  const size_t es = dtype == VTX_BF16 ? 2 : 4;
  const char* kb = (const char*)kv;
  char* out = (char*)dkv;
  int rc = LA_DISPATCH(lattn_bwd_t, q, kb, kb + es * hd, o, dout, lse, dq, out, out + es * hd, ws, B, g, st);
  if (rc || bias == nullptr || dbias == nullptr) return rc;
  return LA_DISPATCH(lattn_dbias_t, q, kb, kb + es * hd, dout, lse, ws, dbias, B, g, st);
}
