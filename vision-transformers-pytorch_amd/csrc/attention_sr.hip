// Spatial-reduction (cross) attention of PVT (and of Twins-SVT's global sub-sampled attention) for gfx950: head dim D = 64
// or 32 (template parameter), Lq queries against Lk <= 64 reduced keys.
//
// Replaces reference models/pvt.py:38-66 per (image, head): q = linear_q(x) [B, Lq, h*64],
// k | v = linear_kv(reduced).chunk(2) [B, Lk, 2*h*64]; S = q k^T / sqrt(64), softmax over the Lk keys, O = P v --
// with the head split / merge as address arithmetic and the scores kept in registers.  PVT-Small at 224^2 has
// Lk = 49 (stages 1-3, after the r x r reduction conv) or 50 (stage 4, cls + 7x7) and Lq = 3136 / 784 / 196 / 50.
//
// Same fragment algebra as attention_win.hip (64-lane waves, mma16, swapped product S^T = K Q^T so that softmax is
// lane-local + two shuffles, outputs from transposed products so every lane stores 16 contiguous bytes), organised for
// many queries against few keys:
//   * a workgroup = 4 waves serves one (image, head) and a run of 64-query sub-chunks; wave w owns query tile w of
//     every sub-chunk (forward, dQ) and KEY tile w for the key-side gradients, which it accumulates in registers over
//     all sub-chunks of the workgroup;
//   * K / V fragments live in registers for the whole workgroup; the token-contracted operands (V^T forward; K^T, Q^T,
//     dO^T backward) are transposed through LDS images shared by the 4 waves; in the backward every wave fetches only
//     its own query tile from HBM, one sub-chunk ahead of the one being computed, and the other waves read its Q / dO
//     row fragments from row-major LDS images (4x fewer global loads, no register spills: PVT-Small +2.9 %);
//   * dK / dV partials of the workgroups of one (image, head) go to fp32 slabs and are summed in fixed order
//     (deterministic, no atomics).
// Templated on T in {bf16, float} (float = parity mode on the exact fp32 MFMA).
#include <stdlib.h>

#include "options.h"
#include "vtx_common.h"

#define SR_LK 64         // padded key count
#define SR_STR 72        // transposed LDS row stride (elements)
#define SR_QB 64         // queries per sub-chunk (4 waves x 16)

struct SrGeom { int Lq, Lk, nH, hd, nsub, qc; float scale; };   // nsub = ceil(Lq / 64), qc = sub-chunks per workgroup

template <typename T> __device__ __forceinline__ Vec8<T> sr_load(const T* p, bool valid) {
  return valid ? load8<T>(p) : vec8_zero<T>();
}
template <typename T> __device__ __forceinline__ Vec8<T> sr_frag_acc(const f32x4& lo, const f32x4& hi) {
  Vec8<T> f;
#pragma unroll
  for (int j = 0; j < 4; ++j) { f.set(j, lo[j]); f.set(4 + j, hi[j]); }
  return f;
}
// fragment of a transposed image Xt[row][token]: k-slots j < 4 <-> tokens t0 + 4g + j, j >= 4 <-> t0 + 16 + 4g + (j-4)
template <typename T> __device__ __forceinline__ Vec8<T> sr_frag_t(const T* p, int g) {
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
// One 16-token tile (lane (c, g): token t0 + c, channels 32 ds + 8 g ..+7) -> transposed image Xt[pi(d)][token]
// with pi(32 ds + 8 g + e) = 16 (2 ds + (e >> 2)) + 4 g + (e & 3): image rows 16 j .. 16 j + 15 (j = 2 dp + dtl) used as
// an MFMA A operand leave lane (c, g) with the outputs of token c for d = 32 dp + 8 g + 4 dtl + r.
template <typename T, int DS> __device__ __forceinline__ void sr_store_t(T* xt, const Vec8<T> (&f)[DS], int t0, int c, int g) {
#pragma unroll
  for (int ds = 0; ds < DS; ++ds)
#pragma unroll
    for (int e = 0; e < 8; ++e) xt[(16 * (2 * ds + (e >> 2)) + 4 * g + (e & 3)) * SR_STR + t0 + c] = f[ds].v[e];
}
template <typename T> __device__ __forceinline__ Vec8<T> sr_out8(const f32x4& a0, const f32x4& a1, float scale) {
  Vec8<T> f;
#pragma unroll
  for (int j = 0; j < 4; ++j) { f.set(j, a0[j] * scale); f.set(4 + j, a1[j] * scale); }
  return f;
}

// --------------------------------------------------------------------------------------------- forward
// grid = (workgroups per (image, head), B * nH)
template <typename T, int D, bool DROP = false>
__global__ __launch_bounds__(256) void srattn_fwd_kernel(const T* __restrict__ q, const T* __restrict__ kv,
                                                        T* __restrict__ o, float* __restrict__ lse, SrGeom g, DropArgs da) {
  constexpr int DS = D / 32, DJ = D / 16;                               // 32-channel k-steps, 16-channel output tiles
  __shared__ __attribute__((aligned(16))) T vt[D * SR_STR];             // Vt[pi(d)][key], shared by the 4 waves
  const int bh = blockIdx.y, h = bh % g.nH, b = bh / g.nH;
  const int lane = threadIdx.x & 63, c_ = lane & 15, g_ = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t ldkv = 2 * (int64_t)g.hd;
  const T* kb = kv + (int64_t)b * g.Lk * ldkv + h * D;

  Vec8<T> kf[4][DS];
  f32x4 kmask[4];                                 // 0 on real keys, -inf on padded ones (lane: keys 16 kt + 4 g + r)
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
    const int key = 16 * kt + c_;
#pragma unroll
    for (int ds = 0; ds < DS; ++ds) kf[kt][ds] = sr_load<T>(kb + (int64_t)key * ldkv + 32 * ds + 8 * g_, key < g.Lk);
#pragma unroll
    for (int r = 0; r < 4; ++r) kmask[kt][r] = (16 * kt + 4 * g_ + r) < g.Lk ? 0.f : -INFINITY;
  }
  {
    const int key = 16 * wave + c_;               // wave w transposes V tile w for everybody
    Vec8<T> vf[DS];
#pragma unroll
    for (int ds = 0; ds < DS; ++ds) vf[ds] = sr_load<T>(kb + g.hd + (int64_t)key * ldkv + 32 * ds + 8 * g_, key < g.Lk);
    sr_store_t<T>(vt, vf, 16 * wave, c_, g_);
  }
  __syncthreads();

  for (int sc = 0; sc < g.qc; ++sc) {
    const int sub = blockIdx.x * g.qc + sc;
    if (sub >= g.nsub) break;
    const int qi = sub * SR_QB + 16 * wave + c_;
    const bool qv = qi < g.Lq;
    const int64_t qrow = (int64_t)b * g.Lq + (qv ? qi : 0);
    Vec8<T> qf[DS];
#pragma unroll
    for (int ds = 0; ds < DS; ++ds) qf[ds] = sr_load<T>(q + qrow * g.hd + h * D + 32 * ds + 8 * g_, qv);
    f32x4 st[4];
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      st[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ds = 0; ds < DS; ++ds) mma16(kf[kt][ds], qf[ds], st[kt]);   // S[q = c][key = 16 kt + 4 g + r]
#pragma unroll
      for (int r = 0; r < 4; ++r) { st[kt][r] = st[kt][r] * g.scale + kmask[kt][r]; m = fmaxf(m, st[kt][r]); }
    }
    m = fmaxf(m, shfl_xor_f(m, 16));
    m = fmaxf(m, shfl_xor_f(m, 32));
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) { st[kt][r] = __expf(st[kt][r] - m); l += st[kt][r]; }
    l += shfl_xor_f(l, 16);
    l += shfl_xor_f(l, 32);
    const float inv = 1.f / l;
    if (qv && g_ == 0) lse[(int64_t)bh * g.Lq + qi] = m + __logf(l);
    if constexpr (DROP) {                           // F.dropout on the normalised probabilities (pvt.py:60, twins.py:88)
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) st[kt][r] *= drop_factor(da, (unsigned)bh, qi, 16 * kt + 4 * g_ + r);
    }
    f32x4 oacc[DJ];
#pragma unroll
    for (int j = 0; j < DJ; ++j) oacc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      Vec8<T> pf = sr_frag_acc<T>(st[2 * ks] * inv, st[2 * ks + 1] * inv);
#pragma unroll
      for (int j = 0; j < DJ; ++j) mma16(sr_frag_t<T>(vt + (16 * j + c_) * SR_STR + 32 * ks, g_), pf, oacc[j]);
    }
    // oacc[2 dp + dtl][r] = O[q = c][d = 32 dp + 8 g + 4 dtl + r]
    if (qv) {
      T* op = o + qrow * g.hd + h * D + 8 * g_;
#pragma unroll
      for (int dp = 0; dp < DS; ++dp) store8<T>(op + 32 * dp, sr_out8<T>(oacc[2 * dp], oacc[2 * dp + 1], 1.f));
    }
  }
}

// --------------------------------------------------------------------------------------------- backward
// same grid; part: fp32 [B * nH][workgroups][64 keys][2 D] (dK channels 0..D-1, dV channels D..2D-1 of the head)
template <typename T, int D, bool DROP = false>
__global__ __launch_bounds__(256, 2) void srattn_bwd_kernel(const T* __restrict__ q, const T* __restrict__ kv,
                                                           const T* __restrict__ oin, const T* __restrict__ dout,
                                                           const float* __restrict__ lse, T* __restrict__ dq,
                                                           float* __restrict__ part, SrGeom g, DropArgs da) {
  constexpr int DS = D / 32, DJ = D / 16;
  __shared__ __attribute__((aligned(16))) T kt_s[D * SR_STR];           // Kt[pi(d)][key]
  __shared__ __attribute__((aligned(16))) T qt_s[D * SR_STR];           // Qt[pi(d)][q of the sub-chunk]
  __shared__ __attribute__((aligned(16))) T dot_s[D * SR_STR];          // dOt[pi(d)][q]
  __shared__ __attribute__((aligned(16))) T qr_s[SR_QB * SR_STR];       // Q [q][d] row-major (phase B row fragments)
  __shared__ __attribute__((aligned(16))) T dor_s[SR_QB * SR_STR];      // dO[q][d]
  __shared__ __attribute__((aligned(16))) float dq_s[SR_QB];            // D[q] = rowsum(dO o O)
  __shared__ __attribute__((aligned(16))) float lse_s[SR_QB];
  const int bh = blockIdx.y, h = bh % g.nH, b = bh / g.nH;
  const int lane = threadIdx.x & 63, c_ = lane & 15, g_ = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t ldkv = 2 * (int64_t)g.hd;
  const T* kb = kv + (int64_t)b * g.Lk * ldkv + h * D;

  Vec8<T> kf[4][DS], vf[4][DS];
  f32x4 kmask[4];
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
    const int key = 16 * kt + c_;
#pragma unroll
    for (int ds = 0; ds < DS; ++ds) {
      kf[kt][ds] = sr_load<T>(kb + (int64_t)key * ldkv + 32 * ds + 8 * g_, key < g.Lk);
      vf[kt][ds] = sr_load<T>(kb + g.hd + (int64_t)key * ldkv + 32 * ds + 8 * g_, key < g.Lk);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) kmask[kt][r] = (16 * kt + 4 * g_ + r) < g.Lk ? 0.f : -INFINITY;
  }
  // wave w: its own key tile as the B operands of phase B, and the transposed K tile for everybody
  Vec8<T> kw[DS], vw[DS];
#pragma unroll
  for (int ds = 0; ds < DS; ++ds) {
    kw[ds] = kf[0][ds]; vw[ds] = vf[0][ds];
#pragma unroll
    for (int kt = 1; kt < 4; ++kt)
      if (wave == kt) { kw[ds] = kf[kt][ds]; vw[ds] = vf[kt][ds]; }
  }
  sr_store_t<T>(kt_s, kw, 16 * wave, c_, g_);
  const float kbias = (16 * wave + c_) < g.Lk ? 0.f : -INFINITY;         // phase B: this lane's key
  f32x4 dkacc[DJ], dvacc[DJ];                      // [2 dp + dtl][r]: key = 16 w + c, d = 32 dp + 8 g + 4 dtl + r
#pragma unroll
  for (int j = 0; j < DJ; ++j) { dkacc[j] = f32x4{0.f, 0.f, 0.f, 0.f}; dvacc[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  // Every wave fetches only ITS query tile of a sub-chunk (Q, dO, O rows) -- one sub-chunk ahead of the one being
  // computed, so the loads fly under the MFMAs -- and publishes it to the other waves through LDS: transposed images
  // (token-contracted operands) and row-major images (the Q / dO row fragments of phase B).
  auto fetch = [&](int sub, Vec8<T> (&qm)[DS], Vec8<T> (&dom)[DS], Vec8<T> (&om)[DS], float& lq) {
    const int qi = sub * SR_QB + 16 * wave + c_;
    const bool qv = sub < g.nsub && qi < g.Lq;
    const int64_t qrow = (int64_t)b * g.Lq + (qv ? qi : 0);
#pragma unroll
    for (int ds = 0; ds < DS; ++ds) {
      qm[ds] = sr_load<T>(q + qrow * g.hd + h * D + 32 * ds + 8 * g_, qv);
      dom[ds] = sr_load<T>(dout + qrow * g.hd + h * D + 32 * ds + 8 * g_, qv);
      om[ds] = sr_load<T>(oin + qrow * g.hd + h * D + 32 * ds + 8 * g_, qv);
    }
    lq = qv ? lse[(int64_t)bh * g.Lq + qi] : INFINITY;             // padded queries: exp(. - inf) = 0
  };
  Vec8<T> qn[DS], don[DS], on[DS];
  float lqn;
  fetch(blockIdx.x * g.qc, qn, don, on, lqn);

  for (int sc = 0; sc < g.qc; ++sc) {
    const int sub = blockIdx.x * g.qc + sc;
    if (sub >= g.nsub) break;
    const int q0 = sub * SR_QB;
    Vec8<T> qm[DS], dom[DS];
#pragma unroll
    for (int ds = 0; ds < DS; ++ds) { qm[ds] = qn[ds]; dom[ds] = don[ds]; }
    const float lq = lqn;
    float dsum = 0.f;
#pragma unroll
    for (int ds = 0; ds < DS; ++ds)
#pragma unroll
      for (int e = 0; e < 8; ++e) dsum += on[ds].get(e) * dom[ds].get(e);
    dsum += shfl_xor_f(dsum, 16);
    dsum += shfl_xor_f(dsum, 32);
    const int qi = q0 + 16 * wave + c_;
    const bool qv = qi < g.Lq;
    const int64_t qrow = (int64_t)b * g.Lq + (qv ? qi : 0);
    if (sc + 1 < g.qc) fetch(sub + 1, qn, don, on, lqn);              // next sub-chunk's rows: in flight during this one
    __syncthreads();                               // the previous sub-chunk's readers of the shared images are done
    sr_store_t<T>(qt_s, qm, 16 * wave, c_, g_);
    sr_store_t<T>(dot_s, dom, 16 * wave, c_, g_);
#pragma unroll
    for (int ds = 0; ds < DS; ++ds) {
      store8<T>(qr_s + (16 * wave + c_) * SR_STR + 32 * ds + 8 * g_, qm[ds]);
      store8<T>(dor_s + (16 * wave + c_) * SR_STR + 32 * ds + 8 * g_, dom[ds]);
    }
    if (g_ == 0) { dq_s[16 * wave + c_] = dsum; lse_s[16 * wave + c_] = lq; }
    __syncthreads();

    // ---------------- phase A (wave <-> query tile `wave`): dQ = scale * dS K
    {
      f32x4 dqacc[DJ];
#pragma unroll
      for (int j = 0; j < DJ; ++j) dqacc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        f32x4 dsv[2];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int kt = 2 * ks + half;
          f32x4 pt = f32x4{0.f, 0.f, 0.f, 0.f}, dpt = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ds = 0; ds < DS; ++ds) { mma16(kf[kt][ds], qm[ds], pt); mma16(vf[kt][ds], dom[ds], dpt); }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = __expf(pt[r] * g.scale + kmask[kt][r] - lq);
            float dpv = dpt[r];
            if constexpr (DROP) dpv *= drop_factor(da, (unsigned)bh, qi, 16 * kt + 4 * g_ + r);
            dsv[half][r] = p * (dpv - dsum);
          }
        }
        Vec8<T> dsf = sr_frag_acc<T>(dsv[0], dsv[1]);
#pragma unroll
        for (int j = 0; j < DJ; ++j) mma16(sr_frag_t<T>(kt_s + (16 * j + c_) * SR_STR + 32 * ks, g_), dsf, dqacc[j]);
      }
      if (qv) {
        T* p = dq + qrow * g.hd + h * D + 8 * g_;
#pragma unroll
        for (int dp = 0; dp < DS; ++dp) store8<T>(p + 32 * dp, sr_out8<T>(dqacc[2 * dp], dqacc[2 * dp + 1], g.scale));
      }
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---------------- phase B (wave <-> key tile `wave`): dV += P^T dO, dK += scale * dS^T Q over this sub-chunk
#pragma unroll
    for (int qs = 0; qs < 2; ++qs) {
      f32x4 pp[2], dss[2];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int t = 2 * qs + half;
        f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ds = 0; ds < DS; ++ds) {
          const Vec8<T> qf = load8<T>(qr_s + (16 * t + c_) * SR_STR + 32 * ds + 8 * g_);
          const Vec8<T> dof = load8<T>(dor_s + (16 * t + c_) * SR_STR + 32 * ds + 8 * g_);
          mma16(qf, kw[ds], s);
          mma16(dof, vw[ds], dp);
        }
        const f32x4 ls = *reinterpret_cast<const f32x4*>(lse_s + 16 * t + 4 * g_);   // rows q = 16 t + 4 g + r
        const f32x4 dd = *reinterpret_cast<const f32x4*>(dq_s + 16 * t + 4 * g_);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __expf(s[r] * g.scale + kbias - ls[r]);
          float f = 1.f;
          if constexpr (DROP) f = drop_factor(da, (unsigned)bh, q0 + 16 * t + 4 * g_ + r, 16 * wave + c_);
          pp[half][r] = p * f;
          dss[half][r] = p * (dp[r] * f - dd[r]);
        }
      }
      Vec8<T> pf = sr_frag_acc<T>(pp[0], pp[1]);
      Vec8<T> dsf = sr_frag_acc<T>(dss[0], dss[1]);
#pragma unroll
      for (int j = 0; j < DJ; ++j) {
        mma16(sr_frag_t<T>(dot_s + (16 * j + c_) * SR_STR + 32 * qs, g_), pf, dvacc[j]);
        mma16(sr_frag_t<T>(qt_s + (16 * j + c_) * SR_STR + 32 * qs, g_), dsf, dkacc[j]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  // partial dK | dV of this workgroup: row key = 16 w + c, 8 contiguous channels per accumulator pair
  float* pr = part + (((int64_t)bh * gridDim.x + blockIdx.x) * SR_LK + 16 * wave + c_) * (2 * D) + 8 * g_;
#pragma unroll
  for (int dp = 0; dp < DS; ++dp) {
    store8<float>(pr + 32 * dp, sr_out8<float>(dkacc[2 * dp], dkacc[2 * dp + 1], g.scale));
    store8<float>(pr + D + 32 * dp, sr_out8<float>(dvacc[2 * dp], dvacc[2 * dp + 1], 1.f));
  }
}

// dkv[b * Lk + key][(k | v) * hd + h * D + d] = sum over workgroups (fixed order) of the partial slabs
template <typename T, int D>
__global__ void srattn_reduce_kernel(const float* __restrict__ part, T* __restrict__ dkv, int nwg, int Lk, int nH,
                                     int64_t total) {
  constexpr int C4 = D / 2;                        // float4 groups per key row of a slab (2 D channels)
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // over (bh, key, 2 D / 4)
  if (idx >= total) return;
  const int c4 = (int)(idx % C4), key = (int)((idx / C4) % Lk);
  const int64_t bh = idx / (C4 * (int64_t)Lk);
  const int h = (int)(bh % nH);
  const int64_t b = bh / nH;
  const float* p = part + ((bh * nwg) * SR_LK + key) * (2 * D) + 4 * c4;
  f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int w = 0; w < nwg; ++w) s += *reinterpret_cast<const f32x4*>(p + (int64_t)w * SR_LK * 2 * D);
  const int ch = 4 * c4;                           // 0..D-1 dK, D..2D-1 dV
  T* out = dkv + (b * Lk + key) * (2 * (int64_t)nH * D) + (ch >= D ? (int64_t)nH * D + (ch - D) : ch) + h * D;
#pragma unroll
  for (int e = 0; e < 4; ++e) out[e] = from_f32<T>(s[e]);
}

static int sr_geom(SrGeom& g, int Lq, int Lk, int nH, int B, int D) {
  if (Lq <= 0 || Lk <= 0 || Lk > SR_LK || nH <= 0 || B <= 0 || (D != 64 && D != 32)) return VTX_ERR_SHAPE;
  g.Lq = Lq; g.Lk = Lk; g.nH = nH; g.hd = nH * D;
  g.nsub = (Lq + SR_QB - 1) / SR_QB;
  // sub-chunks per workgroup: enough workgroups to fill the chip (~2048), as few partial slabs as possible
  int target = vtx_opt(VTX_OPT_SRATTN_WGS);
  if (target <= 0) target = 2048;
  int64_t per_bh = target / ((int64_t)B * nH);
  if (per_bh < 1) per_bh = 1;
  g.qc = (int)((g.nsub + per_bh - 1) / per_bh);
  if (g.qc < 1) g.qc = 1;
  g.scale = 1.0f / sqrtf((float)D);
  return VTX_OK;
}
static int sr_wgs(const SrGeom& g) { return (g.nsub + g.qc - 1) / g.qc; }

// score[b][h][i][j] = <q[b, i, h, :], k[b, j, h, :]> / sqrt(D): the pre-softmax scores the reference's
// MultiHeadedAttention.forward RETURNS next to its output (models/pvt.py:53, 69).  Not on the training path (the PVT
// layers discard it), so a plain kernel: one workgroup per (image, head, 64 queries), K in LDS, fp32 accumulation.
template <typename T, int D>
__global__ __launch_bounds__(256) void srattn_score_kernel(const T* __restrict__ q, const T* __restrict__ kv,
                                                           T* __restrict__ score, int Lq, int Lk, int nH, float scale) {
  __shared__ float ks[SR_LK][D + 1];
  const int bh = blockIdx.y, b = bh / nH, h = bh - b * nH;
  const int hd = nH * D;
  const int q0 = blockIdx.x * 64;
  for (int k0 = 0; k0 < Lk; k0 += SR_LK) {            // blocks of 64 keys (any Lk)
    const int nk = min(SR_LK, Lk - k0);
    __syncthreads();
    for (int i = threadIdx.x; i < nk * D; i += 256) {
      const int j = i / D, d = i - j * D;
      ks[j][d] = to_f32<T>(kv[((int64_t)b * Lk + k0 + j) * 2 * hd + h * D + d]);
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 64 * nk; idx += 256) {
      const int qi = q0 + idx / nk, j = idx % nk;
      if (qi >= Lq) continue;
      const T* qp = q + ((int64_t)b * Lq + qi) * hd + h * D;
      float s = 0.f;
#pragma unroll 8
      for (int d = 0; d < D; ++d) s += to_f32<T>(qp[d]) * ks[j][d];
      score[(((int64_t)b * nH + h) * Lq + qi) * Lk + k0 + j] = from_f32<T>(s * scale);
    }
  }
}

template <typename T, int D>
static int sr_launch_scores(const void* q, const void* kv, void* score, int B, int Lq, int Lk, int nH, hipStream_t st) {
  dim3 grid((Lq + 63) / 64, B * nH);
  hipLaunchKernelGGL((srattn_score_kernel<T, D>), grid, dim3(256), 0, st, (const T*)q, (const T*)kv, (T*)score, Lq, Lk, nH,
                     1.0f / sqrtf((float)D));
  return vtx_check_launch();
}
template <typename T, int D>
static int sr_launch_fwd(const void* q, const void* kv, void* o, float* lse, int B, const SrGeom& g, hipStream_t st,
                         const DropArgs* da = nullptr) {
  dim3 grid(sr_wgs(g), B * g.nH);
  if (da)
    hipLaunchKernelGGL((srattn_fwd_kernel<T, D, true>), grid, dim3(256), 0, st, (const T*)q, (const T*)kv, (T*)o, lse, g, *da);
  else
    hipLaunchKernelGGL((srattn_fwd_kernel<T, D, false>), grid, dim3(256), 0, st, (const T*)q, (const T*)kv, (T*)o, lse, g, DropArgs{});
  return vtx_check_launch();
}
template <typename T, int D>
static int sr_launch_bwd(const void* q, const void* kv, const void* o, const void* dout, const float* lse, void* dq, void* dkv,
                         float* part, int B, const SrGeom& g, hipStream_t st, const DropArgs* da = nullptr) {
  const int nwg = sr_wgs(g);
  dim3 grid(nwg, B * g.nH);
  if (da)
    hipLaunchKernelGGL((srattn_bwd_kernel<T, D, true>), grid, dim3(256), 0, st, (const T*)q, (const T*)kv, (const T*)o, (const T*)dout,
                       lse, (T*)dq, part, g, *da);
  else
    hipLaunchKernelGGL((srattn_bwd_kernel<T, D, false>), grid, dim3(256), 0, st, (const T*)q, (const T*)kv, (const T*)o, (const T*)dout,
                       lse, (T*)dq, part, g, DropArgs{});
  int rc = vtx_check_launch();
  if (rc) return rc;
  const int64_t total = (int64_t)B * g.nH * g.Lk * (D / 2);
  const int rb = (int)((total + 255) / 256);
  hipLaunchKernelGGL((srattn_reduce_kernel<T, D>), dim3(rb), dim3(256), 0, st, (const float*)part, (T*)dkv, nwg, g.Lk, g.nH,
                     total);
  return vtx_check_launch();
}
// (dtype, head dim) -> instantiation
#define SR_DISPATCH(fn, ...)                                                              \
  (dtype == VTX_BF16 ? (D == 64 ? fn<bf16, 64>(__VA_ARGS__) : fn<bf16, 32>(__VA_ARGS__))   \
   : dtype == VTX_F32 ? (D == 64 ? fn<float, 64>(__VA_ARGS__) : fn<float, 32>(__VA_ARGS__)) \
                      : VTX_ERR_DTYPE)

// more than 64 reduced keys (PVT / Twins at 384 x 384 and beyond): the key-block / online-softmax kernels of attention_long.hip
bool lattn_ok(int dtype, int D);
int lattn_cross_fwd_launch(const void* q, const void* kv, void* o, float* lse, int B, int Lq, int Lk, int nH, int D, int dtype,
                           hipStream_t st, const float* bias = nullptr, const DropArgs* da = nullptr);
int lattn_cross_bwd_launch(const void* q, const void* kv, const void* o, const void* dout, const float* lse, void* dq, void* dkv,
                           float* ws, int B, int Lq, int Lk, int nH, int D, int dtype, hipStream_t st, const float* bias = nullptr,
                           float* dbias = nullptr, const DropArgs* da = nullptr);
static bool sr_long(int B, int Lq, int Lk, int nH, int D, int dtype) {
  return Lk > SR_LK && B > 0 && Lq > 0 && nH > 0 && lattn_ok(dtype, D) && (int64_t)B * Lq < 0x7fffffff;
}

extern "C" {

/* Pre-softmax scores q k^T / sqrt(D) [B, nH, Lq, Lk] of the same operands as vtx_srattn_fwd (reference models/pvt.py:53):
 * the second return value of pvt.MultiHeadedAttention.forward (models/pvt.py:69); no gradient flows through it here. */
int vtx_srattn_scores(const void* q, const void* kv, void* score, int B, int Lq, int Lk, int nH, int D, int dtype,
                      void* stream) {
  if (!q || !kv || !score) return VTX_ERR_NULL;
  if (B <= 0 || Lq <= 0 || Lk <= 0 || nH <= 0 || (D != 64 && D != 32)) return VTX_ERR_SHAPE;
  return SR_DISPATCH(sr_launch_scores, q, kv, score, B, Lq, Lk, nH, (hipStream_t)stream);
}

/* Spatial-reduction attention of PVT (reference models/pvt.py:38-66) and the global sub-sampled attention of Twins-SVT
 * (models/twins.py:56-93), head dim D = 64 | 32, Lk <= 64 keys:
 * q [B*Lq, nH*D], kv [B*Lk, 2*nH*D] (k | v halves, head-major inside each), o [B*Lq, nH*D], lse [B*nH*Lq] fp32. */
int vtx_srattn_fwd(const void* q, const void* kv, void* o, float* lse, int B, int Lq, int Lk, int nH, int D, int dtype,
                   void* stream) {
  if (!q || !kv || !o || !lse) return VTX_ERR_NULL;
  if (sr_long(B, Lq, Lk, nH, D, dtype)) return lattn_cross_fwd_launch(q, kv, o, lse, B, Lq, Lk, nH, D, dtype, (hipStream_t)stream);
  SrGeom g;
  int rc = sr_geom(g, Lq, Lk, nH, B, D);
  if (rc) return rc;
  return SR_DISPATCH(sr_launch_fwd, q, kv, o, lse, B, g, (hipStream_t)stream);
}

/* The same with dropout of the attention probabilities (reference models/pvt.py:60, models/twins.py:88): keep mask by hash of
 * (seed, image * nH + head, query * Lk + key), or `keep` [B*nH][Lq][Lk] bytes when given (see vtx_attention_fwd_drop). */
int vtx_srattn_fwd_drop(const void* q, const void* kv, void* o, float* lse, int B, int Lq, int Lk, int nH, int D, int dtype,
                        float drop_p, uint64_t seed, const uint8_t* keep, void* stream) {
  if (!q || !kv || !o || !lse) return VTX_ERR_NULL;
  DropArgs da;
  int rc = drop_args(da, drop_p, seed, keep, Lq, Lk);
  if (rc) return rc;
  if (sr_long(B, Lq, Lk, nH, D, dtype)) return lattn_cross_fwd_launch(q, kv, o, lse, B, Lq, Lk, nH, D, dtype, (hipStream_t)stream, nullptr, &da);
  SrGeom g;
  rc = sr_geom(g, Lq, Lk, nH, B, D);
  if (rc) return rc;
  return SR_DISPATCH(sr_launch_fwd, q, kv, o, lse, B, g, (hipStream_t)stream, &da);
}

size_t vtx_srattn_bwd_workspace(int B, int Lq, int Lk, int nH, int D) {
  if (Lk > SR_LK) return (size_t)B * nH * Lq * sizeof(float);      // rowsum(dO o O) per query (attention_long.hip)
  SrGeom g;
  if (sr_geom(g, Lq, Lk, nH, B, D)) return 0;
  return (size_t)B * nH * sr_wgs(g) * SR_LK * 2 * D * sizeof(float);
}

/* dq [B*Lq, nH*D], dkv [B*Lk, 2*nH*D]; deterministic (fixed-order slab reduction). */
int vtx_srattn_bwd(const void* q, const void* kv, const void* o, const void* dout, const float* lse, void* dq, void* dkv,
                   void* workspace, size_t ws_bytes, int B, int Lq, int Lk, int nH, int D, int dtype, void* stream) {
  if (!q || !kv || !o || !dout || !lse || !dq || !dkv || !workspace) return VTX_ERR_NULL;
  if (sr_long(B, Lq, Lk, nH, D, dtype)) {
    if (ws_bytes < vtx_srattn_bwd_workspace(B, Lq, Lk, nH, D)) return VTX_ERR_WORKSPACE;
    return lattn_cross_bwd_launch(q, kv, o, dout, lse, dq, dkv, (float*)workspace, B, Lq, Lk, nH, D, dtype, (hipStream_t)stream);
  }
  SrGeom g;
  int rc = sr_geom(g, Lq, Lk, nH, B, D);
  if (rc) return rc;
  if (ws_bytes < vtx_srattn_bwd_workspace(B, Lq, Lk, nH, D)) return VTX_ERR_WORKSPACE;
  return SR_DISPATCH(sr_launch_bwd, q, kv, o, dout, lse, dq, dkv, (float*)workspace, B, g, (hipStream_t)stream);
}

int vtx_srattn_bwd_drop(const void* q, const void* kv, const void* o, const void* dout, const float* lse, void* dq, void* dkv,
                        void* workspace, size_t ws_bytes, int B, int Lq, int Lk, int nH, int D, int dtype, float drop_p,
                        uint64_t seed, const uint8_t* keep, void* stream) {
  if (!q || !kv || !o || !dout || !lse || !dq || !dkv || !workspace) return VTX_ERR_NULL;
  DropArgs da;
  int rc = drop_args(da, drop_p, seed, keep, Lq, Lk);
  if (rc) return rc;
  if (ws_bytes < vtx_srattn_bwd_workspace(B, Lq, Lk, nH, D)) return VTX_ERR_WORKSPACE;
  if (sr_long(B, Lq, Lk, nH, D, dtype))
    return lattn_cross_bwd_launch(q, kv, o, dout, lse, dq, dkv, (float*)workspace, B, Lq, Lk, nH, D, dtype, (hipStream_t)stream, nullptr,
                                  nullptr, &da);
  SrGeom g;
  rc = sr_geom(g, Lq, Lk, nH, B, D);
  if (rc) return rc;
  return SR_DISPATCH(sr_launch_bwd, q, kv, o, dout, lse, dq, dkv, (float*)workspace, B, g, (hipStream_t)stream, &da);
}

}  // extern "C"
