// Window attention (Swin) for gfx950, specialised for head dim 32 and windows of <= 64 tokens:
// ONE WAVEFRONT PER (image, window, head) PROBLEM, no workgroup barriers on the data path.
//
// Same math and addressing as attention.hip (reference models/swin_transformer.py:109-154: roll, window
// partition, q k^T / sqrt(d) + rel_pos(pos) bias, masked_fill(local_mask, -inf), softmax, @ v, inverse
// partition, roll back) but organised for the regime these problems live in -- 49x49x32 is far too small
// to be MFMA-bound, the kernel is HBM / latency bound:
//   * every global load of a problem (Q, K, V [, dO, O] fragments: 12 / 20 x 16 B per lane) is issued up
//     front, independent of each other, so a problem pays the HBM latency once;
//   * operands that are contracted over tokens (V; K, Q, dO in the backward) are transposed through a
//     wave-private LDS region straight from those registers (no second global read);
//   * bias and -inf mask come pre-combined from a padded table [nWm][h][64][64] fp32 (and its transpose)
//     built once per layer call, so the per-score lookup is one aligned 16-byte load per 4 scores;
//   * the rel_pos gradient is binned per wave into 172 wave-private LDS bins (ds_add_f32; single wave,
//     program order => deterministic) accumulated over all problems a persistent wave processes, then
//     reduced over waves in fixed order.
#include <stdlib.h>

#include "vtx_common.h"

#define WA_D 32
#define WA_LP 64
#define WA_STR 72        // transposed LDS row stride (elements)
#define WA_NBIN 172      // (2*7-1)^2 = 169 padded to a multiple of 4

struct WinGeom {
  int L, nH, hd, nW, H, W, win, shift, nWx, nWm;   // nWm = windows with distinct masks (nW if shifted else 1)
  float scale;
};

__device__ __forceinline__ int64_t win_token_row(const WinGeom& g, int b, int n, int i) {
  const int wi = n / g.nWx, wj = n - wi * g.nWx;
  const int ay = i / g.win, ax = i - ay * g.win;
  int y = wi * g.win + ay + g.shift; if (y >= g.H) y -= g.H;
  int x = wj * g.win + ax + g.shift; if (x >= g.W) x -= g.W;
  return ((int64_t)b * g.H + y) * g.W + x;
}

template <typename T> __device__ __forceinline__ Vec8<T> wa_load(const T* p, bool valid) {
  return valid ? load8<T>(p) : vec8_zero<T>();
}
template <typename T> __device__ __forceinline__ Vec8<T> wa_frag_acc(const f32x4& lo, const f32x4& hi) {
  Vec8<T> f;
#pragma unroll
  for (int j = 0; j < 4; ++j) { f.set(j, lo[j]); f.set(4 + j, hi[j]); }
  return f;
}
template <typename T> __device__ __forceinline__ Vec8<T> wa_frag_t(const T* p, int g) {   // p -> Xt[d][32*ks]
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
// registers (token tile t4, lane (c, g): token 16 t4 + c, d-slots 8g..8g+7) -> transposed LDS image Xt[pi(d)][token]
// with the row permutation pi(8g + e) = 16 (e >> 2) + 4 g + (e & 3).  A product X^T-tile x F^T taken with the
// image rows 16 dt + (0..15) as the MFMA A operand then leaves lane (c, g) holding, in accumulator dt register r,
// the output of token (column) c for d = 8 g + 4 dt + r: eight CONTIGUOUS head channels per lane, i.e. exactly the
// fragment layout of the loads -> one 16-byte global store per lane and tile instead of eight 2-byte ones.
template <typename T> __device__ __forceinline__ void wa_store_t(T* xt, const Vec8<T> (&f)[4], int c, int g) {
#pragma unroll
  for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
    for (int e = 0; e < 8; ++e) xt[(16 * (e >> 2) + 4 * g + (e & 3)) * WA_STR + 16 * t4 + c] = f[t4].v[e];
}
// accumulators of the two d-tiles of such a transposed product -> the lane's 8 contiguous channels
template <typename T> __device__ __forceinline__ Vec8<T> wa_out8(const f32x4& a0, const f32x4& a1, float scale) {
  Vec8<T> f;
#pragma unroll
  for (int j = 0; j < 4; ++j) { f.set(j, a0[j] * scale); f.set(4 + j, a1[j] * scale); }
  return f;
}

// ---- bias + mask tables: bm[m][h][q][k] and bmT[m][h][k][q], 64x64 padded, fp32; -inf on masked / padded keys
__global__ void win_bias_mask_kernel(const float* __restrict__ rel_pos, const int64_t* __restrict__ pos,
                                     const uint8_t* __restrict__ mask, float* __restrict__ bm, float* __restrict__ bmT,
                                     int L, int nH, int nWm) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int total = nWm * nH * 64 * 64;
  if (idx >= total) return;
  const int k = idx & 63, q = (idx >> 6) & 63, mh = idx >> 12;
  const int h = mh % nH, m = mh / nH;
  float v;
  if (k >= L) v = -INFINITY;
  else if (q >= L) v = 0.f;
  else {
    v = rel_pos[pos[q * L + k] * nH + h];
    if (mask && mask[((int64_t)m * L + q) * L + k]) v = -INFINITY;
  }
  bm[idx] = v;
  bmT[(((int64_t)mh * 64 + k) << 6) + q] = v;
}

// --------------------------------------------------------------------------------------------- forward
template <typename T>
__global__ __launch_bounds__(64) void wattn_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ o,
                                                      float* __restrict__ lse, const float* __restrict__ bm,
                                                      int nprob, WinGeom g) {
  __shared__ __attribute__((aligned(16))) T vt[WA_D * WA_STR];
  // XCD-aware remap: consecutive problems (heads of one window share 128-B lines) stay on one XCD
  const int did = blockIdx.x;
  const int xq = nprob >> 3, xr = nprob & 7, xcd = did & 7;
  const int prob = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (did >> 3);
  const int h = prob % g.nH;
  const int bn = prob / g.nH;
  const int n = bn % g.nW, b = bn / g.nW;
  const int lane = threadIdx.x, c_ = lane & 15, g_ = lane >> 4;
  const int64_t ld = 3 * (int64_t)g.hd;

  int64_t row[4];
  bool val[4];
  Vec8<T> qf[4], kf[4], vf[4];
#pragma unroll
  for (int t4 = 0; t4 < 4; ++t4) {
    const int tok = 16 * t4 + c_;
    val[t4] = tok < g.L;
    row[t4] = val[t4] ? win_token_row(g, b, n, tok) : 0;
    const T* p = qkv + row[t4] * ld + h * WA_D + g_ * 8;
    qf[t4] = wa_load<T>(p, val[t4]);
    kf[t4] = wa_load<T>(p + g.hd, val[t4]);
    vf[t4] = wa_load<T>(p + 2 * g.hd, val[t4]);
  }
  wa_store_t<T>(vt, vf, c_, g_);
  __syncthreads();
  const float* bmh = bm + (((int64_t)(g.nWm > 1 ? n : 0) * g.nH + h) << 12);

#pragma unroll
  for (int qt = 0; qt < 4; ++qt) {
    if (qt * 16 >= g.L) break;
    const int q = qt * 16 + c_;
    f32x4 st[4];
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      st[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
      mma16(kf[kt], qf[qt], st[kt]);            // st[kt][r] = S[q = 16 qt + c][key = 16 kt + 4 g + r]
      const f32x4 bb = *reinterpret_cast<const f32x4*>(bmh + q * 64 + kt * 16 + g_ * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) { st[kt][r] = st[kt][r] * g.scale + bb[r]; m = fmaxf(m, st[kt][r]); }
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
    if (val[qt] && g_ == 0) lse[(int64_t)prob * g.L + q] = m + __logf(l);
    f32x4 oacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      Vec8<T> pf = wa_frag_acc<T>(st[2 * ks] * inv, st[2 * ks + 1] * inv);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) mma16(wa_frag_t<T>(vt + (dt * 16 + c_) * WA_STR + ks * 32, g_), pf, oacc[dt]);
    }
    // oacc[dt][r] = O[q = 16 qt + c][d = 8 g + 4 dt + r]
    if (val[qt]) store8<T>(o + row[qt] * (int64_t)g.hd + h * WA_D + g_ * 8, wa_out8<T>(oacc[0], oacc[1], 1.f));
  }
}

// --------------------------------------------------------------------------------------------- backward
// grid = (nblk, nH) persistent waves; wave (x, h) walks the (image, window) pairs x, x + nblk, ... of head h.
template <typename T>
__global__ __launch_bounds__(64, 2) void wattn_bwd_kernel(const T* __restrict__ qkv, const T* __restrict__ oin,
                                                      const T* __restrict__ dout, const float* __restrict__ lse,
                                                      const float* __restrict__ bm, const float* __restrict__ bmT,
                                                      const int* __restrict__ posT, T* __restrict__ dqkv,
                                                      float* __restrict__ bins_part, int nbn, WinGeom g) {
  __shared__ __attribute__((aligned(16))) T kt_s[WA_D * WA_STR];     // Kt[d][key]
  __shared__ __attribute__((aligned(16))) T qt_s[WA_D * WA_STR];     // Qt[d][q]
  __shared__ __attribute__((aligned(16))) T dot_s[WA_D * WA_STR];    // dOt[d][q]
  __shared__ __attribute__((aligned(16))) float dq_s[WA_LP];         // D[q] = rowsum(dO o O)
  __shared__ __attribute__((aligned(16))) float lse_s[WA_LP];
  __shared__ float bins[WA_NBIN];
  const int h = blockIdx.y;
  const int lane = threadIdx.x, c_ = lane & 15, g_ = lane >> 4;
  const int64_t ld = 3 * (int64_t)g.hd;
  for (int i = lane; i < WA_NBIN; i += 64) bins[i] = 0.f;
  // running sum over this wave's problems of dS[q = 16 qt + 4 g + r][key = 16 kt + c] (all of head h): LDS float
  // atomics cost ~100 LDS cycles per instruction, so the rel_pos binning is done ONCE per wave, not per problem
  f32x4 dsacc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) dsacc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int bn = blockIdx.x; bn < nbn; bn += gridDim.x) {
    const int n = bn % g.nW, b = bn / g.nW;
    const int prob = bn * g.nH + h;
    const int64_t mh = (int64_t)(g.nWm > 1 ? n : 0) * g.nH + h;
    const float* bmh = bm + (mh << 12);
    const float* bmTh = bmT + (mh << 12);

    int64_t row[4];
    bool val[4];
    Vec8<T> qf[4], kf[4], vf[4], dof[4];
    float dsum[4], lq[4];
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) {
      const int tok = 16 * t4 + c_;
      val[t4] = tok < g.L;
      row[t4] = val[t4] ? win_token_row(g, b, n, tok) : 0;
      const T* p = qkv + row[t4] * ld + h * WA_D + g_ * 8;
      qf[t4] = wa_load<T>(p, val[t4]);
      kf[t4] = wa_load<T>(p + g.hd, val[t4]);
      vf[t4] = wa_load<T>(p + 2 * g.hd, val[t4]);
      dof[t4] = wa_load<T>(dout + row[t4] * g.hd + h * WA_D + g_ * 8, val[t4]);
      Vec8<T> of = wa_load<T>(oin + row[t4] * g.hd + h * WA_D + g_ * 8, val[t4]);
      lq[t4] = val[t4] ? lse[(int64_t)prob * g.L + tok] : 0.f;
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) s += of.get(e) * dof[t4].get(e);
      s += shfl_xor_f(s, 16);
      s += shfl_xor_f(s, 32);
      dsum[t4] = s;                                   // D[q = 16 t4 + c]
    }
    __syncthreads();                                  // previous problem's LDS readers are done (single wave: cheap)
    wa_store_t<T>(kt_s, kf, c_, g_);
    wa_store_t<T>(qt_s, qf, c_, g_);
    wa_store_t<T>(dot_s, dof, c_, g_);
    if (g_ == 0) {
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4) { dq_s[16 * t4 + c_] = dsum[t4]; lse_s[16 * t4 + c_] = lq[t4]; }
    }
    __syncthreads();

    // ---------------- phase A (swapped layout, per query tile): dQ = scale * dS K
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
      if (qt * 16 >= g.L) break;
      const int q = qt * 16 + c_;
      f32x4 dqacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        f32x4 dsv[2];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int kt = 2 * ks + half;
          f32x4 pt = f32x4{0.f, 0.f, 0.f, 0.f}, dpt = f32x4{0.f, 0.f, 0.f, 0.f};
          mma16(kf[kt], qf[qt], pt);          // S [q = 16 qt + c][key = 16 kt + 4 g + r]
          mma16(vf[kt], dof[qt], dpt);        // dP[same]
          const f32x4 bb = *reinterpret_cast<const f32x4*>(bmh + q * 64 + kt * 16 + g_ * 4);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = val[qt] ? __expf(pt[r] * g.scale + bb[r] - lq[qt]) : 0.f;
            dsv[half][r] = p * (dpt[r] - dsum[qt]);
          }
        }
        Vec8<T> dsf = wa_frag_acc<T>(dsv[0], dsv[1]);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) mma16(wa_frag_t<T>(kt_s + (dt * 16 + c_) * WA_STR + ks * 32, g_), dsf, dqacc[dt]);
      }
      // dqacc[dt][r] = dQ[q = 16 qt + c][d = 8 g + 4 dt + r] / scale
      if (val[qt]) store8<T>(dqkv + row[qt] * ld + h * WA_D + g_ * 8, wa_out8<T>(dqacc[0], dqacc[1], g.scale));
      __builtin_amdgcn_sched_barrier(0);   // keep iterations apart: no cross-iteration hoisting (register pressure)
    }

    // ---------------- phase B (plain layout, per key tile): dV = P^T dO, dK = scale * dS^T Q, bins += dS
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      if (kt * 16 >= g.L) break;
      const int key = kt * 16 + c_;
      f32x4 dkacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      f32x4 dvacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int qs = 0; qs < 2; ++qs) {
        f32x4 pp[2], dss[2];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int qt = 2 * qs + half;
          f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
          mma16(qf[qt], kf[kt], s);           // S [q = 16 qt + 4 g + r][key = 16 kt + c]
          mma16(dof[qt], vf[kt], dp);         // dP[same]
          const int q0 = qt * 16 + g_ * 4;
          const f32x4 bb = *reinterpret_cast<const f32x4*>(bmTh + key * 64 + q0);
          const f32x4 ls = *reinterpret_cast<const f32x4*>(lse_s + q0);
          const f32x4 dd = *reinterpret_cast<const f32x4*>(dq_s + q0);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool ok = (q0 + r) < g.L && val[kt];
            const float p = ok ? __expf(s[r] * g.scale + bb[r] - ls[r]) : 0.f;
            pp[half][r] = p;
            dss[half][r] = p * (dp[r] - dd[r]);
          }
          dsacc[kt][qt] += dss[half];
        }
        Vec8<T> pf = wa_frag_acc<T>(pp[0], pp[1]);
        Vec8<T> dsf = wa_frag_acc<T>(dss[0], dss[1]);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          mma16(wa_frag_t<T>(dot_s + (dt * 16 + c_) * WA_STR + qs * 32, g_), pf, dvacc[dt]);
          mma16(wa_frag_t<T>(qt_s + (dt * 16 + c_) * WA_STR + qs * 32, g_), dsf, dkacc[dt]);
        }
      }
      // d{k,v}acc[dt][r] = d{K,V}[key = 16 kt + c][d = 8 g + 4 dt + r]
      if (val[kt]) {
        T* p = dqkv + row[kt] * ld + h * WA_D + g_ * 8;
        store8<T>(p + g.hd, wa_out8<T>(dkacc[0], dkacc[1], g.scale));
        store8<T>(p + 2 * g.hd, wa_out8<T>(dvacc[0], dvacc[1], 1.f));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // bin the accumulated dS by relative-position index (ds_add_f32; single wave, program order => deterministic)
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
    const int key = kt * 16 + c_;
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
      const int q0 = qt * 16 + g_ * 4;
      const int4 pb = *reinterpret_cast<const int4*>(posT + key * 64 + q0);
      const int pbin[4] = {pb.x, pb.y, pb.z, pb.w};
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (key < g.L && q0 + r < g.L) atomicAdd(&bins[pbin[r]], dsacc[kt][qt][r]);
    }
  }
  __syncthreads();
  // partial layout [wave x][bin][head]: the fixed-order column reduce then yields drel_pos[(bin, head)] directly
  float* out = bins_part + (int64_t)blockIdx.x * WA_NBIN * g.nH + h;
  for (int i = lane; i < WA_NBIN; i += 64) out[(int64_t)i * g.nH] = bins[i];
}

// transposed, padded pos table: posT[key][q] (int32, 64 x 64), 0 where out of range (those never accumulate)
__global__ void win_pos_t_kernel(const int64_t* __restrict__ pos, int* __restrict__ posT, int L) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 64 * 64) return;
  const int q = idx & 63, k = idx >> 6;
  posT[idx] = (q < L && k < L) ? (int)pos[q * L + k] : 0;
}

static int win_geom(WinGeom& g, int L, int nH, int H, int W, int win, int shift) {
  if (win <= 0 || H % win || W % win || L != win * win || L > WA_LP) return VTX_ERR_SHAPE;
  if ((2 * win - 1) * (2 * win - 1) > WA_NBIN) return VTX_ERR_SHAPE;
  g.L = L; g.nH = nH; g.hd = nH * WA_D; g.H = H; g.W = W; g.win = win;
  g.nWx = W / win; g.nW = (H / win) * (W / win); g.shift = shift ? win / 2 : 0;
  g.nWm = 1;
  g.scale = 1.0f / sqrtf((float)WA_D);
  return VTX_OK;
}

// Persistent waves: 2 per SIMD are resident (VGPR-bound) => 2048 on the chip; every wave of head h gets the same
// number of (image, window) pairs (+-1) and the grid never exceeds what is resident at once.
static int wattn_bwd_blocks(int nbn, int nH) {
  static int cap = -1;
  if (cap < 0) { const char* e = getenv("VTX_WATTN_WAVES"); cap = e ? atoi(e) : 2048; }
  int per_head = cap / nH;
  if (per_head < 1) per_head = 1;
  const int ppw = (nbn + per_head - 1) / per_head;   // problems per wave
  return (nbn + ppw - 1) / ppw;
}

extern "C" {

/* Window attention (head dim 32, window <= 64 tokens) -- the Swin path.
 * tables: workspace of vtx_wattn_tables_bytes(): [bm | bmT | posT], built by vtx_wattn_tables from the rel_pos
 * parameter and the pos / local_mask buffers (mask == NULL for un-shifted layers). */
size_t vtx_wattn_tables_bytes(int nH, int nWm) {
  return ((size_t)2 * nWm * nH * 64 * 64) * sizeof(float) + 64 * 64 * sizeof(int);
}

int vtx_wattn_tables(const float* rel_pos, const int64_t* pos, const uint8_t* mask, void* tables, int L, int nH,
                     int nW, void* stream) {
  if (!rel_pos || !pos || !tables) return VTX_ERR_NULL;
  if (L > WA_LP) return VTX_ERR_SHAPE;
  const int nWm = mask ? nW : 1;
  float* bm = (float*)tables;
  float* bmT = bm + (size_t)nWm * nH * 4096;
  int* posT = (int*)(bmT + (size_t)nWm * nH * 4096);
  hipStream_t st = (hipStream_t)stream;
  const int total = nWm * nH * 4096;
  hipLaunchKernelGGL(win_bias_mask_kernel, dim3((total + 255) / 256), dim3(256), 0, st, rel_pos, pos, mask, bm, bmT, L,
                     nH, nWm);
  int rc = vtx_check_launch();
  if (rc) return rc;
  hipLaunchKernelGGL(win_pos_t_kernel, dim3(16), dim3(256), 0, st, pos, posT, L);
  return vtx_check_launch();
}

int vtx_wattn_fwd(const void* qkv, void* o, float* lse, const void* tables, int masked, int B, int L, int nH, int H,
                  int W, int win, int shift, int dtype, void* stream) {
  if (!qkv || !o || !lse || !tables) return VTX_ERR_NULL;
  WinGeom g;
  int rc = win_geom(g, L, nH, H, W, win, shift);
  if (rc) return rc;
  g.nWm = masked ? g.nW : 1;
  const int nprob = B * g.nW * nH;
  if (nprob <= 0) return VTX_OK;
  hipStream_t st = (hipStream_t)stream;
  const float* bm = (const float*)tables;
  if (dtype == VTX_BF16)
    hipLaunchKernelGGL((wattn_fwd_kernel<bf16>), dim3(nprob), dim3(64), 0, st, (const bf16*)qkv, (bf16*)o, lse, bm, nprob, g);
  else if (dtype == VTX_F32)
    hipLaunchKernelGGL((wattn_fwd_kernel<float>), dim3(nprob), dim3(64), 0, st, (const float*)qkv, (float*)o, lse, bm, nprob, g);
  else return VTX_ERR_DTYPE;
  return vtx_check_launch();
}

size_t vtx_wattn_bwd_workspace(int B, int nH, int H, int W, int win) {
  const int nbn = B * (H / win) * (W / win);
  return (size_t)wattn_bwd_blocks(nbn, nH) * nH * WA_NBIN * sizeof(float);
}

int vtx_wattn_bwd(const void* qkv, const void* o, const void* dout, const float* lse, const void* tables, int masked,
                  void* dqkv, float* drel_pos, void* workspace, size_t ws_bytes, int B, int L, int nH, int H, int W,
                  int win, int shift, int dtype, void* stream) {
  if (!qkv || !o || !dout || !lse || !tables || !dqkv || !drel_pos || !workspace) return VTX_ERR_NULL;
  WinGeom g;
  int rc = win_geom(g, L, nH, H, W, win, shift);
  if (rc) return rc;
  g.nWm = masked ? g.nW : 1;
  const int nbn = B * g.nW;
  if (nbn <= 0) return VTX_OK;
  if (ws_bytes < vtx_wattn_bwd_workspace(B, nH, H, W, win)) return VTX_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int nblk = wattn_bwd_blocks(nbn, nH);
  const float* bm = (const float*)tables;
  const float* bmT = bm + (size_t)g.nWm * nH * 4096;
  const int* posT = (const int*)(bmT + (size_t)g.nWm * nH * 4096);
  float* part = (float*)workspace;
  if (dtype == VTX_BF16)
    hipLaunchKernelGGL((wattn_bwd_kernel<bf16>), dim3(nblk, nH), dim3(64), 0, st, (const bf16*)qkv, (const bf16*)o,
                       (const bf16*)dout, lse, bm, bmT, posT, (bf16*)dqkv, part, nbn, g);
  else if (dtype == VTX_F32)
    hipLaunchKernelGGL((wattn_bwd_kernel<float>), dim3(nblk, nH), dim3(64), 0, st, (const float*)qkv, (const float*)o,
                       (const float*)dout, lse, bm, bmT, posT, (float*)dqkv, part, nbn, g);
  else return VTX_ERR_DTYPE;
  rc = vtx_check_launch();
  if (rc) return rc;
  const int ntab = (2 * win - 1) * (2 * win - 1);
  hipLaunchKernelGGL(colreduce_kernel, colreduce_grid(ntab * nH), dim3(256), 0, st, (const float*)part, drel_pos,
                     (float*)nullptr, nblk, ntab * nH, WA_NBIN * nH);
  return vtx_check_launch();
}

}  // extern "C"
