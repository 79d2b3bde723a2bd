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
//   * workgroups are PERSISTENT, 4 waves that all serve one head: the head's rel-pos bias table
//     bias[q][k] = rel_pos[pos[q][k]][h] (-inf on padded keys) is gathered ONCE per workgroup into LDS
//     ([64][68] fp32, conflict-free for both the row-wise 16-byte and the column-wise 4-byte reads), so no score
//     lookup ever leaves the CU -- global table loads inside the tile loops were ~25 % of the backward's time;
//   * the shifted-window -inf mask is carried as one REGION id per token (local_mask[n][a][b] == region[n][a] !=
//     region[n][b]; the host verifies that the module's local_mask buffer has this structure, else the generic
//     kernels of attention.hip run): 64 bytes per problem in LDS instead of a 16 KB mask table per window;
//   * the rel_pos gradient: dS is summed over all problems of the persistent wave in registers, spilled ONCE per wave into
//     the LDS region of its operand images and GATHERED per table bin over a host-built inverse map of `pos`
//     (tables.pos_inverse: lane b adds up bin b in a fixed order => deterministic), then reduced over waves in fixed order.
//     (Round 1 binned with ds_add_f32 -- ~100 LDS cycles per instruction, 26-30 us of every backward launch: replaced in
//     round 2; the scatter survives only behind inv_cells == NULL for the tests.)
#include <stdlib.h>

#include <type_traits>

#include "options.h"
#include "vtx_common.h"

#define WA_D 32
#define WA_LP 64
#define WA_STR 72        // transposed LDS row stride (elements)
#define WA_NBIN 172      // (2*7-1)^2 = 169 padded to a multiple of 4
#define WA_BSTR 68       // bias table row stride (floats)
#define WA_WAVES 4       // waves per workgroup
// phase ablation of the backward for timing (separate builds, tools/probe/build_ablate.sh-style; 0 in the library):
// 1 no final dS binning | 2 no partial-bin write-out | 4 no phase A (dQ) | 8 no phase B (dK, dV, dS sum) |
// 16 no bias-table build (forward and backward) | 32 forward: no score / softmax / PV compute (loads + stores only)
#ifndef WA_ABLATE
#define WA_ABLATE 0
#endif

struct WinGeom {
  int L, nH, hd, nW, H, W, win, shift, nWx;
  float scale;
  const int* perm;     // stochastic-depth compaction: the b-th image a kernel works on is image perm[b] (null: identity)
};

// Round 3: the per-problem address arithmetic was 550 of the backward's ~2 650 instructions per problem (a runtime division
// by the window size per token, 64-bit products and an exec-masked branch around each of the 20 loads -- the kernel is
// instruction-issue bound at 2 waves per SIMD).  A lane's token (16 t4 + c) always sits at the same (ay, ax) of its window:
// computed ONCE per kernel; per problem only the window origin changes.  Tokens past L (the padding of the last 16-token
// tile) read the window's token 0 instead of being predicated off: their keys carry a -inf bias and their queries a +inf
// lse, so every product they enter is multiplied by an exact 0 -- same bits as the zero fill it replaces, no branches.
struct WaTok {
  int ay[4], ax[4];
  bool val[4];
};
__device__ __forceinline__ void wa_tok_init(WaTok& k, const WinGeom& g, int c) {
#pragma unroll
  for (int t4 = 0; t4 < 4; ++t4) {
    const int tok = 16 * t4 + c;
    k.val[t4] = tok < g.L;
    const int t = k.val[t4] ? tok : 0;
    k.ay[t4] = t / g.win;
    k.ax[t4] = t - k.ay[t4] * g.win;
  }
}
// token row of window (wi, wj) of image b (rows < 2^31: checked by the entry points)
__device__ __forceinline__ int wa_row(const WinGeom& g, const WaTok& k, int t4, int b, int wi, int wj) {
  int y = wi * g.win + k.ay[t4] + g.shift;
  y -= y >= g.H ? g.H : 0;
  int x = wj * g.win + k.ax[t4] + g.shift;
  x -= x >= g.W ? g.W : 0;
  return (b * g.H + y) * g.W + x;
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


// wave-level ordering point for wave-private LDS traffic (DS operations of one wave execute in issue order; this
// only stops the compiler from moving memory operations across it -- no workgroup barrier, no counter drain)
__device__ __forceinline__ void wa_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// true if every live lane of the wave holds the same region id
__device__ __forceinline__ bool wa_uniform(uint8_t reg, bool live) {
  const int first = __builtin_amdgcn_readfirstlane((int)reg);
  return __ballot(live && (int)reg != first) == 0ull;
}
__device__ __forceinline__ float group_max16(float v) {      // max over the 16 lanes of a DPP row, in every lane
  v = fmaxf(v, dpp_f<0xB1>(v));
  v = fmaxf(v, dpp_f<0x4E>(v));
  v = fmaxf(v, dpp_f<0x141>(v));
  v = fmaxf(v, dpp_f<0x140>(v));
  return v;
}

// the head's bias table, built once per workgroup: bias_s[q][k] (row stride WA_BSTR), -inf on padded keys.
// Column h of rel_pos goes to LDS first (relh, WA_NBIN floats) while the 16 pos loads of every thread are in flight,
// so the build pays ONE global round trip.
__device__ __forceinline__ void wa_build_bias(float* bias_s, float* relh, const float* __restrict__ rel_pos,
                                              const int64_t* __restrict__ pos, int L, int nH, int h, int ntab,
                                              int qrows = 64) {
  constexpr int PER = 64 * 64 / (64 * WA_WAVES);
  if (WA_ABLATE & 16) return;
  int pidx[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int idx = threadIdx.x + i * 64 * WA_WAVES;
    const int k = idx & 63, q = idx >> 6;
    pidx[i] = (k < L && q < L) ? (int)pos[q * L + k] : -1;
  }
  if ((int)threadIdx.x < ntab) relh[threadIdx.x] = rel_pos[threadIdx.x * nH + h];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int idx = threadIdx.x + i * 64 * WA_WAVES;
    const int k = idx & 63, q = idx >> 6;
    if (q < qrows) bias_s[q * WA_BSTR + k] = k >= L ? -INFINITY : (pidx[i] >= 0 ? relh[pidx[i]] : 0.f);
  }
}

template <typename T> struct WaSmem {
  static constexpr int kImg = WA_D * WA_STR * (int)sizeof(T);          // one transposed operand image
  static constexpr int kBias = 64 * WA_BSTR * 4;
  static constexpr int kFwdWave = kImg + 64 + 64 * (int)sizeof(T);     // Vt + region ids + one row of P
  static constexpr int kBwdWave = 3 * kImg + 2 * WA_LP * 4 + WA_NBIN * 4 + 64;
  static constexpr int kFwd = kBias + WA_WAVES * kFwdWave;
  static constexpr int kBwd = kBias + WA_WAVES * kBwdWave;
};

// Workgroup -> (block of the head's persistent grid, head), 1-D grid of nblk * nH workgroups.  Head dim 32 in bf16 is
// 64 bytes: TWO heads share every 128-byte line of q / k / v / o / dO.  With the head as the slow grid dimension the two
// workgroups that need a line run far apart in time and on unrelated XCDs, and the line is fetched from HBM twice (PMC:
// backward 1.43x its algorithmic bytes).  `xcd_major` (nblk % 8 == 0): workgroup b runs on XCD b % 8 (observed dispatch
// rule, speed only), so give each XCD the sequence slot = b / 8 -> head = slot % nH fastest, block = 8 (slot / nH) + XCD:
// all heads of the same (image, window) pairs are dispatched back to back on ONE XCD and share its L2 lines.
__device__ __forceinline__ void wa_block_map(int nblk, int nH, int xcd_major, int& blk, int& h) {
  const int b = blockIdx.x;
  if (xcd_major) {
    const int xcd = b & 7, slot = b >> 3;
    h = slot % nH;
    blk = (slot / nH) * 8 + xcd;
  } else {
    blk = b % nblk;                 // head-major (round-1 order): all blocks of head 0, then head 1, ...
    h = b / nblk;
  }
}

// --------------------------------------------------------------------------------------------- forward
// grid = nblk * nH workgroups (wa_block_map -> block x, head y); wave w of block x serves the (image, window) pairs
// 4 x + w, 4 x + w + 4 nblk, ... of head y
template <typename T, bool MASKED>
__global__ __launch_bounds__(64 * WA_WAVES) void wattn_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ o,
                                                                  float* __restrict__ lse,
                                                                  const float* __restrict__ rel_pos,
                                                                  const int64_t* __restrict__ pos,
                                                                  const uint8_t* __restrict__ region, int nbn,
                                                                  int nblk, int xcd_major, int fast, WinGeom g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char wa_smem[];
  float* bias_s = reinterpret_cast<float*>(wa_smem);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned char* wbase = wa_smem + WaSmem<T>::kBias + wave * WaSmem<T>::kFwdWave;
  T* vt = reinterpret_cast<T*>(wbase);
  uint8_t* reg_s = wbase + WaSmem<T>::kImg;
  T* prow = reinterpret_cast<T*>(wbase + WaSmem<T>::kImg + 64);       // P[48][key] of a 7 x 7 window's last query
  int blk, h;
  wa_block_map(nblk, g.nH, xcd_major, blk, h);
  const int lane = threadIdx.x & 63, c_ = lane & 15, g_ = lane >> 4;
  const int64_t ld = 3 * (int64_t)g.hd;
  wa_build_bias(bias_s, reinterpret_cast<float*>(wa_smem + WaSmem<T>::kBias), rel_pos, pos, g.L, g.nH, h,
                (2 * g.win - 1) * (2 * g.win - 1));
  __syncthreads();                                    // table complete; the relh scratch (wave 0's region) is free again
  WaTok tk;
  wa_tok_init(tk, g, c_);
  const bool w7 = g.L == 49;
  const bool row48 = w7 && (fast & 1);                // VTX_WATTN_FAST bit 0: the one-row path of the 49th query
  if (row48) tk.ay[3] = tk.ax[3] = 6;                    // every lane of the last tile reads token 48 (its padded lanes read token 0 otherwise)

  for (int bn = blk * WA_WAVES + wave; bn < nbn; bn += nblk * WA_WAVES) {
    const int n = bn % g.nW, b = g.perm ? g.perm[bn / g.nW] : bn / g.nW;
    const int wi = n / g.nWx, wj = n - wi * g.nWx;      // (wave-uniform: scalar arithmetic)
    const int prob = bn * g.nH + h;
    int row[4];
    const bool (&val)[4] = tk.val;
    Vec8<T> qf[4], kf[4], vf[4];
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) {
      row[t4] = wa_row(g, tk, t4, b, wi, wj);
      const T* p = qkv + (int64_t)row[t4] * ld + h * WA_D + g_ * 8;
      qf[t4] = load8<T>(p);
      kf[t4] = load8<T>(p + g.hd);
      vf[t4] = load8<T>(p + 2 * g.hd);
    }
    uint8_t myreg = 0;
    if (MASKED) myreg = region[(int64_t)n * 64 + lane];
    wa_wave_sync();                                   // the previous problem's LDS readers are done
    wa_store_t<T>(vt, vf, c_, g_);
    if (MASKED) reg_s[lane] = myreg;
    wa_wave_sync();

    if (WA_ABLATE & 32) {
#pragma unroll
      for (int qt = 0; qt < 4; ++qt)
        if (val[qt]) store8<T>(o + (int64_t)row[qt] * g.hd + h * WA_D + g_ * 8, qf[qt]);
      continue;
    }
    // P V for one query tile from the P^T fragments of its two key halves; lane (c, g) ends up with O[q = 16 qt + c][d = 8 g .. 8 g + 7]
    auto pv_store = [&](int qt, const Vec8<T>& p0, const Vec8<T>& p1) {
      f32x4 oacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) mma16(wa_frag_t<T>(vt + (dt * 16 + c_) * WA_STR + ks * 32, g_), ks ? p1 : p0, oacc[dt]);
      // oacc[dt][r] = O[q = 16 qt + c][d = 8 g + 4 dt + r]
      if (val[qt]) store8<T>(o + (int64_t)row[qt] * g.hd + h * WA_D + g_ * 8, wa_out8<T>(oacc[0], oacc[1], 1.f));
    };
    auto tiles = [&](auto mk) {
      constexpr bool MK = decltype(mk)::value;
#pragma unroll
      for (int qt = 0; qt < 4; ++qt) {
        if (qt * 16 >= g.L) break;
        if (qt == 3 && row48) {
          // 7 x 7 windows: the last query tile holds ONE live query (token 48).  All 16 rows of the q operand are that query (the
          // token override above), the product is taken the other way round -- s[0] = S[48][key = 16 kt + c] in EVERY lane -- and the
          // softmax of the row costs 4 values per lane instead of 16; P goes through 64 elements of LDS into the fragment layout
          // of the P V product (round 5: ~1/7 of the forward's instructions).
          float sv[4], m = -INFINITY;
          const uint8_t r48 = MK ? reg_s[48] : 0;
#pragma unroll
          for (int kt = 0; kt < 4; ++kt) {
            f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
            mma16(qf[3], kf[kt], s);
            sv[kt] = s[0] * g.scale + bias_s[48 * WA_BSTR + kt * 16 + c_];
            if (MK && reg_s[kt * 16 + c_] != r48) sv[kt] = -INFINITY;
            m = fmaxf(m, sv[kt]);
          }
          m = group_max16(m);
          float l = 0.f;
#pragma unroll
          for (int kt = 0; kt < 4; ++kt) {
            sv[kt] = __expf(sv[kt] - m);
            l += sv[kt];
          }
          l = group_sum<16>(l);
          const float inv = 1.f / l;
          if (lane == 0) lse[(int64_t)prob * g.L + 48] = m + __logf(l);
          Vec8<T> one;
#pragma unroll
          for (int kt = 0; kt < 4; ++kt) {
            one.set(0, sv[kt] * inv);
            prow[kt * 16 + c_] = one.v[0];
          }
          wa_wave_sync();
          pv_store(3, wa_frag_t<T>(prow, g_), wa_frag_t<T>(prow + 32, g_));
          break;
        }
        const int q = qt * 16 + c_;
        f32x4 st[4];
        float m = -INFINITY;
        const unsigned rq = MK ? reg_s[q] * 0x01010101u : 0u;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
          st[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
          mma16(kf[kt], qf[qt], st[kt]);            // st[kt][r] = S[q = 16 qt + c][key = 16 kt + 4 g + r]
          const f32x4 bb = *reinterpret_cast<const f32x4*>(bias_s + q * WA_BSTR + kt * 16 + g_ * 4);
          unsigned rx = 0u;                           // byte r == 0  <=>  key r is in the query's region
          if (MK) rx = *reinterpret_cast<const unsigned*>(reg_s + kt * 16 + g_ * 4) ^ rq;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            // 7 x 7 windows (49 tokens): keys 49 .. 63 are padding in every lane -- of the last key tile only register 0 (key 48 + 4 g)
            // can be live.  Their bias is -inf, i.e. p = 0 exactly: skipping them changes no bit (round 5; ~6 % of the VALU work).
            if (kt == 3 && r > 0 && w7) continue;
            float sv = st[kt][r] * g.scale + bb[r];
            if (MK && (rx & (0xffu << (8 * r))) != 0u) sv = -INFINITY;
            st[kt][r] = sv;
            m = fmaxf(m, sv);
          }
        }
        m = fmaxf(m, shfl_xor_f(m, 16));
        m = fmaxf(m, shfl_xor_f(m, 32));
        float l = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (kt == 3 && r > 0 && w7) { st[kt][r] = 0.f; continue; }
            st[kt][r] = __expf(st[kt][r] - m);
            l += st[kt][r];
          }
        l += shfl_xor_f(l, 16);
        l += shfl_xor_f(l, 32);
        const float inv = 1.f / l;
        if (val[qt] && g_ == 0) lse[(int64_t)prob * g.L + q] = m + __logf(l);
        pv_store(qt, wa_frag_acc<T>(st[0] * inv, st[1] * inv), wa_frag_acc<T>(st[2] * inv, st[3] * inv));
      }
    };
    // A window whose tokens all carry one region id (the interior windows of a shifted layer: 49 of the 64 at 56 x 56) has no masked
    // pair: it takes the unmasked instruction stream (wave-uniform branch; same bits -- the mask arithmetic it skips selects nothing).
    if (MASKED && !((fast & 2) && wa_uniform(myreg, lane < g.L)))
      tiles(std::true_type{});
    else
      tiles(std::false_type{});
  }
}

// --------------------------------------------------------------------------------------------- backward
// same persistent mapping as the forward; every wave keeps the running dS sum of its problems in registers
template <typename T, bool MASKED>
__global__ __launch_bounds__(64 * WA_WAVES, 2) void wattn_bwd_kernel(
    const T* __restrict__ qkv, const T* __restrict__ oin, const T* __restrict__ dout, const float* __restrict__ lse,
    const float* __restrict__ rel_pos, const int64_t* __restrict__ pos, const uint8_t* __restrict__ region,
    T* __restrict__ dqkv, float* __restrict__ bins_part, const int* __restrict__ inv_cells, int inv_count, int nbn,
    int nblk, int xcd_major, WinGeom g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char wa_smem[];
  float* bias_s = reinterpret_cast<float*>(wa_smem);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned char* wbase = wa_smem + WaSmem<T>::kBias + wave * WaSmem<T>::kBwdWave;
  T* kt_s = reinterpret_cast<T*>(wbase);                              // Kt[pi(d)][key]
  T* qt_s = reinterpret_cast<T*>(wbase + WaSmem<T>::kImg);            // Qt[pi(d)][q]
  T* dot_s = reinterpret_cast<T*>(wbase + 2 * WaSmem<T>::kImg);       // dOt[pi(d)][q]
  float* dq_s = reinterpret_cast<float*>(wbase + 3 * WaSmem<T>::kImg); // D[q] = rowsum(dO o O)
  float* lse_s = dq_s + WA_LP;
  float* bins = lse_s + WA_LP;
  uint8_t* reg_s = reinterpret_cast<uint8_t*>(bins + WA_NBIN);
  int blk, h;
  wa_block_map(nblk, g.nH, xcd_major, blk, h);
  const int lane = threadIdx.x & 63, c_ = lane & 15, g_ = lane >> 4;
  const int64_t ld = 3 * (int64_t)g.hd;
  wa_build_bias(bias_s, reinterpret_cast<float*>(wa_smem + WaSmem<T>::kBias), rel_pos, pos, g.L, g.nH, h,
                (2 * g.win - 1) * (2 * g.win - 1));
  __syncthreads();                                    // table complete; the relh scratch (wave 0's region) is free again
  for (int i = lane; i < WA_NBIN; i += 64) bins[i] = 0.f;
  // running sum over this wave's problems of dS[q = 16 qt + 4 g + r][key = 16 kt + c] (all of head h)
  f32x4 dsacc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) dsacc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  WaTok tk;
  wa_tok_init(tk, g, c_);

  for (int bn = blk * WA_WAVES + wave; bn < nbn; bn += nblk * WA_WAVES) {
    const int n = bn % g.nW, b = g.perm ? g.perm[bn / g.nW] : bn / g.nW;
    const int wi = n / g.nWx, wj = n - wi * g.nWx;      // (wave-uniform: scalar arithmetic)
    const int prob = bn * g.nH + h;

    int row[4];
    const bool (&val)[4] = tk.val;
    Vec8<T> qf[4], kf[4], vf[4], dof[4];
    float dsum[4], lq[4];
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) {
      row[t4] = wa_row(g, tk, t4, b, wi, wj);
      const T* p = qkv + (int64_t)row[t4] * ld + h * WA_D + g_ * 8;
      const int64_t ro = (int64_t)row[t4] * g.hd + h * WA_D + g_ * 8;
      qf[t4] = load8<T>(p);
      kf[t4] = load8<T>(p + g.hd);
      vf[t4] = load8<T>(p + 2 * g.hd);
      dof[t4] = load8<T>(dout + ro);
      Vec8<T> of = load8<T>(oin + ro);
      const float lv = lse[(int64_t)prob * g.L + (val[t4] ? 16 * t4 + c_ : 0)];
      lq[t4] = val[t4] ? lv : INFINITY;               // padded query rows: exp(. - inf) = 0
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) s += of.get(e) * dof[t4].get(e);
      s += shfl_xor_f(s, 16);
      s += shfl_xor_f(s, 32);
      dsum[t4] = s;                                   // D[q = 16 t4 + c]
    }
    uint8_t myreg = 0;
    if (MASKED) myreg = region[(int64_t)n * 64 + lane];
    wa_wave_sync();                                   // the previous problem's LDS readers are done
    wa_store_t<T>(kt_s, kf, c_, g_);
    wa_store_t<T>(qt_s, qf, c_, g_);
    wa_store_t<T>(dot_s, dof, c_, g_);
    if (g_ == 0) {
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4) { dq_s[16 * t4 + c_] = dsum[t4]; lse_s[16 * t4 + c_] = lq[t4]; }
    }
    if (MASKED) reg_s[lane] = myreg;
    wa_wave_sync();

    constexpr bool MK = MASKED;
    // ---------------- phase A (swapped layout, per query tile): dQ = scale * dS K
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
      if (qt * 16 >= g.L || (WA_ABLATE & 4)) break;
      const int q = qt * 16 + c_;
      const unsigned rq = MK ? reg_s[q] * 0x01010101u : 0u;
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
          const f32x4 bb = *reinterpret_cast<const f32x4*>(bias_s + q * WA_BSTR + kt * 16 + g_ * 4);
          unsigned rx = 0u;                         // byte r == 0  <=>  key r is in the query's region
          if (MK) rx = *reinterpret_cast<const unsigned*>(reg_s + kt * 16 + g_ * 4) ^ rq;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float p = __expf(pt[r] * g.scale + bb[r] - lq[qt]);   // padded q: lse = +inf, padded key: bias = -inf
            if (MK && (rx & (0xffu << (8 * r))) != 0u) p = 0.f;
            dsv[half][r] = p * (dpt[r] - dsum[qt]);
          }
        }
        Vec8<T> dsf = wa_frag_acc<T>(dsv[0], dsv[1]);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) mma16(wa_frag_t<T>(kt_s + (dt * 16 + c_) * WA_STR + ks * 32, g_), dsf, dqacc[dt]);
      }
      // dqacc[dt][r] = dQ[q = 16 qt + c][d = 8 g + 4 dt + r] / scale
      if (val[qt]) store8<T>(dqkv + (int64_t)row[qt] * ld + h * WA_D + g_ * 8, wa_out8<T>(dqacc[0], dqacc[1], g.scale));
      __builtin_amdgcn_sched_barrier(0);   // keep iterations apart: no cross-iteration hoisting (register pressure)
    }

    // ---------------- phase B (plain layout, per key tile): dV = P^T dO, dK = scale * dS^T Q, dsacc += dS
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      if (kt * 16 >= g.L || (WA_ABLATE & 8)) break;
      const int key = kt * 16 + c_;
      const unsigned rk = MK ? reg_s[key] * 0x01010101u : 0u;
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
          const f32x4 ls = *reinterpret_cast<const f32x4*>(lse_s + q0);
          const f32x4 dd = *reinterpret_cast<const f32x4*>(dq_s + q0);
          unsigned rx = 0u;
          if (MK) rx = *reinterpret_cast<const unsigned*>(reg_s + q0) ^ rk;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float bb = bias_s[(q0 + r) * WA_BSTR + key];
            float p = __expf(s[r] * g.scale + bb - ls[r]);        // padded q: lse = +inf, padded key: bias = -inf
            if (MK && (rx & (0xffu << (8 * r))) != 0u) p = 0.f;
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
        T* p = dqkv + (int64_t)row[kt] * ld + h * WA_D + g_ * 8;
        store8<T>(p + g.hd, wa_out8<T>(dkacc[0], dkacc[1], g.scale));
        store8<T>(p + 2 * g.hd, wa_out8<T>(dvacc[0], dvacc[1], 1.f));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // Bin the accumulated dS by relative-position index.  With the INVERSE map of pos (inv_cells[t][b], t < inv_count: the
  // t-th cell q * 64 + key of bin b in ascending (q, key) order, padded with the cell L * 64; built once per module on the
  // host): the wave spills its dS sum into the LDS region of its operand images (all problems done) and lane b GATHERS
  // bin b -- inv_count coalesced index loads + plain ds_read_b32 per bin, the same trip count for every lane, fixed order.
  // The ds_add_f32 scatter it replaces (kept for callers without the map) cost 26-30 us per launch whatever the stage:
  // 64 LDS atomics per lane with up to 16 lanes on one bin, and all 8 waves of a CU queue on the same LDS atomic unit
  // (phase ablation, tools/probe/build_ablate_wattn.sh: stage 3 85.8 -> 57.1 us, stage 4 60.9 -> 35.2 us without it).
  const int ntab = (2 * g.win - 1) * (2 * g.win - 1);
  if (inv_cells != nullptr && (g.L + 1) * 64 * 4 <= 3 * WaSmem<T>::kImg && !(WA_ABLATE & 1)) {
    float* ds_s = reinterpret_cast<float*>(wbase);           // [L + 1][64] fp32 over the Kt / Qt / dOt images
    wa_wave_sync();
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      const int key = kt * 16 + c_;
#pragma unroll
      for (int qt = 0; qt < 4; ++qt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int q = qt * 16 + g_ * 4 + r;
          if (key < g.L && q < g.L) ds_s[q * 64 + key] = dsacc[kt][qt][r];
        }
      }
    }
    if (lane == 0) ds_s[g.L * 64] = 0.f;                     // the padding cell
    wa_wave_sync();
    for (int b = lane; b < ntab; b += 64) {
      float sum = 0.f;
      const int* __restrict__ ic = inv_cells + b;
#pragma unroll 7
      for (int t = 0; t < inv_count; ++t) sum += ds_s[ic[t * ntab]];
      bins[b] = sum;
    }
  } else {
    // ds_add_f32 scatter; single wave, program order => deterministic
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      const int key = kt * 16 + c_;
#pragma unroll
      for (int qt = 0; qt < 4; ++qt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int q = qt * 16 + g_ * 4 + r;
          if (key < g.L && q < g.L && !(WA_ABLATE & 1)) atomicAdd(&bins[(int)pos[q * g.L + key]], dsacc[kt][qt][r]);
        }
      }
    }
  }
  wa_wave_sync();
  // partial layout [wave][bin][head]: the fixed-order column reduce then yields drel_pos[(bin, head)] directly
  float* out = bins_part + ((int64_t)blk * WA_WAVES + wave) * WA_NBIN * g.nH + h;
  if (!(WA_ABLATE & 2))
    for (int i = lane; i < WA_NBIN; i += 64) out[(int64_t)i * g.nH] = bins[i];
}

// --------------------------------------------------------------------------------------------- backward, 4 waves per problem
// Round 3 (bf16, inverse pos map present).  The one-wave-per-problem backward above holds a problem's Q / K / V / dO fragments
// (64 registers) and the 64-register dS sum: 256 registers, 2 waves per SIMD, and a wave pays its problem's 20 global loads,
// ~2 300 instructions and the stores one after the other -- loads + stores alone take 25 us of the 65-us stage-3 launch, the two
// phases 32 us, the gather 8 us, and they ADD (profiles/round2_wattn_bwd_phase_ablation.txt).  Here the four waves of the
// persistent workgroup share ONE problem: wave w owns token tile w (16 tokens) -- it loads only those rows (5 x 16 B per lane),
// runs phase A for query tile w and phase B for key tile w, and stores dQ / dK / dV of its tokens.  What the phases need of the
// OTHER tiles comes from LDS: plain [token][d] images of K, V, Q, dO, read with ds_read_b128 for the MFMA operands contracted
// over d and with the transposing ds_read_b64_tr_b16 for those contracted over tokens (no transposed copies, no 2-byte stores).  A wave's share of the NEXT problem (20 registers) is requested right
// after the staging barrier, so global latency runs under the phases; the dS sum is 16 registers per wave; ~150 registers ->
// 3 workgroups (12 waves) per CU.  Arithmetic per element is that of the one-wave kernel (same products in the same order).
#define WA4_PSTR 40      // plain image row stride (elements): 80 B
#ifndef WA4_OCC
#define WA4_OCC 3        // workgroups per CU the backward is compiled for (168 registers)
#endif
#define WA4_BROWS 50     // bias table rows kept (queries < L <= 49; padded queries read row 0: their lse is +inf)
#define WA4_MSTR 72      // P / dS matrix row stride (elements)
struct Wa4Smem {
  static constexpr int kBias = WA4_BROWS * WA_BSTR * 4;
  static constexpr int kPlain = WA_LP * WA4_PSTR * 2;          // one plain image (bf16)
  static constexpr int kMat = WA_LP * WA4_MSTR * 2;            // P or dS of the problem, [q][key] bf16
  static constexpr int kSmall = 64;                            // region ids
  static constexpr int kTotal = kBias + 4 * kPlain + 2 * kMat + kSmall;
};
static_assert((WA_LP + 1) * 64 * 4 <= 4 * Wa4Smem::kPlain, "the dS spill must fit the images");
static_assert(3 * Wa4Smem::kTotal <= 160 * 1024, "three workgroups per CU");

typedef __attribute__((ext_vector_type(4))) short wa_s16x4;
typedef __attribute__((ext_vector_type(8))) short wa_s16x8;
typedef __attribute__((address_space(3))) wa_s16x4 wa_lds_s16x4;
// Fragment contracted over TOKENS straight from a plain [token][d] image (ds_read_b64_tr_b16; cf. sa_frag_trp in
// attention_seq.hip): k-slots j < 4 <-> tokens t0 + 4 g + j, j >= 4 <-> t0 + 16 + 4 g + (j - 4); the A row of lane c is the
// head channel d = 8 (c >> 2) + 4 dt + (c & 3) -- the row permutation of wa_store_t, so that accumulator dt register r of
// lane (c, g) is the output of token c for d = 8 g + 4 dt + r (8 contiguous channels per lane: one 16-byte store).
__device__ __forceinline__ Vec8<bf16> wa4_frag_trp(const bf16* img, int t0, int dt, int lane) {
  const int p = lane & 15, g = lane >> 4;
  const int n = ((p & 3) << 3) + 4 * dt;
  wa_s16x4 v[2];
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int r = t0 + half * 16 + g * 4 + (p >> 2);
    v[half] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wa_lds_s16x4*)(img + r * WA4_PSTR + n));
  }
  wa_s16x8 wv = __builtin_shufflevector(v[0], v[1], 0, 1, 2, 3, 4, 5, 6, 7);   // whole-vector cast only (gemm_wgrad_glds.hip)
  Vec8<bf16> f;
  f.v = __builtin_bit_cast(bf16x8, wv);
  return f;
}

__device__ __forceinline__ void wa_wg_sync() {      // workgroup barrier that leaves global loads / stores in flight
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// Fragment contracted over QUERIES from a [q][key] matrix: k-slots as above (rows t0 ...), column key0 + c for lane c
__device__ __forceinline__ Vec8<bf16> wa4_frag_tr(const bf16* mat, int t0, int key0, int lane) {
  const int p = lane & 15, g = lane >> 4;
  const int n = key0 + ((p & 3) << 2);
  wa_s16x4 v[2];
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int r = t0 + half * 16 + g * 4 + (p >> 2);
    v[half] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wa_lds_s16x4*)(mat + r * WA4_MSTR + n));
  }
  wa_s16x8 wv = __builtin_shufflevector(v[0], v[1], 0, 1, 2, 3, 4, 5, 6, 7);
  Vec8<bf16> f;
  f.v = __builtin_bit_cast(bf16x8, wv);
  return f;
}

template <bool MASKED>
__global__ __launch_bounds__(64 * WA_WAVES, WA4_OCC) void wattn_bwd4_kernel(
    const bf16* __restrict__ qkv, const bf16* __restrict__ oin, const bf16* __restrict__ dout, const float* __restrict__ lse,
    const float* __restrict__ rel_pos, const int64_t* __restrict__ pos, const uint8_t* __restrict__ region,
    bf16* __restrict__ dqkv, float* __restrict__ bins_part, const int* __restrict__ inv_cells, int inv_count, int nbn,
    int nblk, int rows_total, int xcd_major, WinGeom g) {
  using T = bf16;
  extern __shared__ __attribute__((aligned(16))) unsigned char wa_smem[];
  float* bias_s = reinterpret_cast<float*>(wa_smem);
  unsigned char* base = wa_smem + Wa4Smem::kBias;
  T* pk = reinterpret_cast<T*>(base);                               // K [token][d]
  T* pv = reinterpret_cast<T*>(base + Wa4Smem::kPlain);
  T* pq = reinterpret_cast<T*>(base + 2 * Wa4Smem::kPlain);
  T* pdo = reinterpret_cast<T*>(base + 3 * Wa4Smem::kPlain);
  T* pm_s = reinterpret_cast<T*>(base + 4 * Wa4Smem::kPlain);        // P [q][key] of the problem (phase A -> phase B)
  T* dm_s = reinterpret_cast<T*>(base + 4 * Wa4Smem::kPlain + Wa4Smem::kMat);   // dS[q][key]
  uint8_t* reg_s = base + 4 * Wa4Smem::kPlain + 2 * Wa4Smem::kMat;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // this wave's token tile
  const int lane = threadIdx.x & 63, c_ = lane & 15, g_ = lane >> 4;
  int blk, h;
  wa_block_map(nblk, g.nH, xcd_major, blk, h);
  const int64_t ld = 3 * (int64_t)g.hd;
  const int ntab = (2 * g.win - 1) * (2 * g.win - 1);
  wa_build_bias(bias_s, reinterpret_cast<float*>(base), rel_pos, pos, g.L, g.nH, h, ntab, WA4_BROWS);
  __syncthreads();                                    // table complete; the relh scratch (the K image) is free again

  // this lane's token 16 w + c: (ay, ax) inside its window, once per kernel (see WaTok)
  const int tok = 16 * w + c_;
  const bool val = tok < g.L;
  const int tcl = val ? tok : 0;
  const int ay = tcl / g.win, ax = tcl - ay * g.win;
  const bool tile_live = 16 * w < g.L;
  const bool w7 = g.L == 49;

  f32x4 dsacc[4];                                     // sum over this workgroup's problems of dS[q = 16 w + c][key = 16 kt + 4 g + r]
#pragma unroll
  for (int j = 0; j < 4; ++j) dsacc[j] = f32x4{0.f, 0.f, 0.f, 0.f};

  Vec8<T> nq, nk, nv, ndo, no;                        // this wave's rows of the NEXT problem
  float nl = 0.f;
  int nrow = 0;
  auto request = [&](int bn) {
    const int n = bn % g.nW, b = g.perm ? g.perm[bn / g.nW] : bn / g.nW;
    const int wi = n / g.nWx, wj = n - wi * g.nWx;
    int y = wi * g.win + ay + g.shift;
    y -= y >= g.H ? g.H : 0;
    int x = wj * g.win + ax + g.shift;
    x -= x >= g.W ? g.W : 0;
    nrow = (b * g.H + y) * g.W + x;
    const T* p = qkv + (int64_t)nrow * ld + h * WA_D + g_ * 8;
    const int64_t ro = (int64_t)nrow * g.hd + h * WA_D + g_ * 8;
    nq = load8<T>(p);
    nk = load8<T>(p + g.hd);
    nv = load8<T>(p + 2 * g.hd);
    ndo = load8<T>(dout + ro);
    no = load8<T>(oin + ro);
    nl = lse[((int64_t)bn * g.nH + h) * g.L + tcl];
  };
  if (blk < nbn) request(blk);

  for (int bn = blk; bn < nbn; bn += nblk) {
    const Vec8<T> qf = nq, kf = nk, vf = nv, dof = ndo;
    const int row = nrow;
    const float lq = val ? nl : INFINITY;             // padded query rows: exp(. - inf) = 0
    float dsum = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) dsum += no.get(e) * dof.get(e);
    dsum += shfl_xor_f(dsum, 16);
    dsum += shfl_xor_f(dsum, 32);                     // D[q = 16 w + c]
    uint8_t myreg = 0;
    if (MASKED && w == 0) myreg = region[(int64_t)(bn % g.nW) * 64 + lane];
    if (bn != blk) wa_wg_sync();                      // every wave is done reading the previous problem's images
    *reinterpret_cast<Vec8<T>*>(pk + tok * WA4_PSTR + g_ * 8) = kf;
    *reinterpret_cast<Vec8<T>*>(pv + tok * WA4_PSTR + g_ * 8) = vf;
    *reinterpret_cast<Vec8<T>*>(pq + tok * WA4_PSTR + g_ * 8) = qf;
    *reinterpret_cast<Vec8<T>*>(pdo + tok * WA4_PSTR + g_ * 8) = dof;
    if (MASKED && w == 0) reg_s[lane] = myreg;
    wa_wg_sync();
    if (bn + nblk < nbn) request(bn + nblk);          // global latency of the next problem runs under the two phases

    constexpr bool MK = MASKED;
    // ---------------- phase A (swapped layout), query tile w: P, dS of the tile (-> LDS for phase B), dQ = scale * dS K
    if (tile_live) {
      const int q = tok;
      const unsigned rq = MK ? reg_s[q] * 0x01010101u : 0u;
      f32x4 dqacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        f32x4 dsv[2], pv2[2];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int kt = 2 * ks + half;
          const Vec8<T> kk = *reinterpret_cast<const Vec8<T>*>(pk + (16 * kt + c_) * WA4_PSTR + g_ * 8);
          const Vec8<T> vv = *reinterpret_cast<const Vec8<T>*>(pv + (16 * kt + c_) * WA4_PSTR + g_ * 8);
          f32x4 pt = f32x4{0.f, 0.f, 0.f, 0.f}, dpt = f32x4{0.f, 0.f, 0.f, 0.f};
          mma16(kk, qf, pt);                  // S [q = 16 w + c][key = 16 kt + 4 g + r]
          mma16(vv, dof, dpt);                // dP[same]
          const f32x4 bb = *reinterpret_cast<const f32x4*>(bias_s + tcl * WA_BSTR + kt * 16 + g_ * 4);
          unsigned rx = 0u;                         // byte r == 0  <=>  key r is in the query's region
          if (MK) rx = *reinterpret_cast<const unsigned*>(reg_s + kt * 16 + g_ * 4) ^ rq;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (kt == 3 && r > 0 && w7) { pv2[half][r] = 0.f; dsv[half][r] = 0.f; continue; }   // always-padded keys of 7 x 7 windows: p = 0
            float p = __expf(pt[r] * g.scale + bb[r] - lq);       // padded q: lse = +inf, padded key: bias = -inf
            if (MK && (rx & (0xffu << (8 * r))) != 0u) p = 0.f;
            pv2[half][r] = p;
            dsv[half][r] = p * (dpt[r] - dsum);
          }
          dsacc[kt] += dsv[half];
        }
        const Vec8<T> pf = wa_frag_acc<T>(pv2[0], pv2[1]);
        const Vec8<T> dsf = wa_frag_acc<T>(dsv[0], dsv[1]);
        // rows q of P / dS: keys 16 (2 ks) + 4 g .. + 3 and 16 (2 ks + 1) + 4 g .. + 3 (8 bytes each)
        bf16x4 plo, phi, dlo, dhi;
#pragma unroll
        for (int j = 0; j < 4; ++j) { plo[j] = pf.v[j]; phi[j] = pf.v[4 + j]; dlo[j] = dsf.v[j]; dhi[j] = dsf.v[4 + j]; }
        *reinterpret_cast<bf16x4*>(pm_s + q * WA4_MSTR + 32 * ks + 4 * g_) = plo;
        *reinterpret_cast<bf16x4*>(pm_s + q * WA4_MSTR + 32 * ks + 16 + 4 * g_) = phi;
        *reinterpret_cast<bf16x4*>(dm_s + q * WA4_MSTR + 32 * ks + 4 * g_) = dlo;
        *reinterpret_cast<bf16x4*>(dm_s + q * WA4_MSTR + 32 * ks + 16 + 4 * g_) = dhi;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) mma16(wa4_frag_trp(pk, ks * 32, dt, lane), dsf, dqacc[dt]);
      }
      // dqacc[dt][r] = dQ[q = 16 w + c][d = 8 g + 4 dt + r] / scale
      if (val) store8<T>(dqkv + (int64_t)row * ld + h * WA_D + g_ * 8, wa_out8<T>(dqacc[0], dqacc[1], g.scale));
    } else {
      // a tile of padding only (16 w >= L): its rows of P / dS are exact zeros
      const bf16x4 z = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        *reinterpret_cast<bf16x4*>(pm_s + tok * WA4_MSTR + 16 * j + 4 * g_) = z;
        *reinterpret_cast<bf16x4*>(dm_s + tok * WA4_MSTR + 16 * j + 4 * g_) = z;
      }
    }
    wa_wg_sync();                                       // P and dS of the whole problem are in LDS
    // ---------------- phase B, key tile w: dV = P^T dO, dK = scale * dS^T Q (P, dS: the bf16 values phase A used)
    if (tile_live) {
      f32x4 dkacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      f32x4 dvacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int qs = 0; qs < 2; ++qs) {
        const Vec8<T> pf = wa4_frag_tr(pm_s, qs * 32, 16 * w, lane);
        const Vec8<T> dsf = wa4_frag_tr(dm_s, qs * 32, 16 * w, lane);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          mma16(wa4_frag_trp(pdo, qs * 32, dt, lane), pf, dvacc[dt]);
          mma16(wa4_frag_trp(pq, qs * 32, dt, lane), dsf, dkacc[dt]);
        }
      }
      // d{k,v}acc[dt][r] = d{K,V}[key = 16 w + c][d = 8 g + 4 dt + r]
      if (val) {
        T* p = dqkv + (int64_t)row * ld + h * WA_D + g_ * 8;
        store8<T>(p + g.hd, wa_out8<T>(dkacc[0], dkacc[1], g.scale));
        store8<T>(p + 2 * g.hd, wa_out8<T>(dvacc[0], dvacc[1], 1.f));
      }
    }
  }

  // rel_pos gradient: the workgroup's dS sum goes to LDS ([q][64] fp32 over the images) and thread b gathers bin b over the
  // inverse map of pos in a fixed order (see the one-wave kernel) -- ONE partial row per workgroup
  wa_wg_sync();
  float* ds_s = reinterpret_cast<float*>(base);
  {
    const int q = tok;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kt * 16 + g_ * 4 + r;
        if (key < g.L && q < g.L) ds_s[q * 64 + key] = dsacc[kt][r];
      }
    if (threadIdx.x == 0) ds_s[g.L * 64] = 0.f;             // the padding cell
  }
  wa_wg_sync();
  for (int b = threadIdx.x; b < WA_NBIN; b += 64 * WA_WAVES) {
    float sum = 0.f;
    if (b < ntab) {
      const int* __restrict__ ic = inv_cells + b;
#pragma unroll 7
      for (int t = 0; t < inv_count; ++t) sum += ds_s[ic[t * ntab]];
    }
    // partial layout [row][bin][head]; rows past this kernel's nblk (the one-wave kernel writes 4 per workgroup) are zero
    bins_part[((int64_t)blk * WA_NBIN + b) * g.nH + h] = sum;
    for (int r = nblk + blk; r < rows_total; r += nblk) bins_part[((int64_t)r * WA_NBIN + b) * g.nH + h] = 0.f;
  }
}

// --------------------------------------------------------------------------------------------- forward, 4 waves per problem
// Round 5 (bf16).  The one-wave forward above pays each problem's 12 global loads, ~1 300 instructions and the stores one after the
// other per wave (116 registers: 4 waves per SIMD, no room for a prefetch -- measured slower with one in round 4): its stage-1 launch
// sits at 78 us, 22 us above what its loads and stores alone take (profiles/round5_wattn_fwd_row_path.txt).  Here, as in the backward
// above, the four waves of the persistent workgroup share ONE problem: a wave owns one 16-token tile -- it loads only those rows
// (3 x 16 B per lane; the NEXT problem's rows are requested right after the staging barrier, 12 registers), puts its K and V rows
// into plain [token][d] LDS images, and runs scores / softmax / P V for its 16 queries against all keys (K fragments: ds_read_b128,
// V contracted over tokens: ds_read_b64_tr_b16).  The tile a wave owns ROTATES from problem to problem ((wave + iteration) % 4): the
// last tile of a 7 x 7 window holds one live query (the one-row path of the one-wave kernel), and the wave index decides the SIMD --
// without the rotation one SIMD of every CU would idle.  Arithmetic per element is that of the one-wave kernel (bit for bit).
#ifndef WF4_OCC
#define WF4_OCC 5        // workgroups per CU the four-wave forward is compiled for
#endif
#ifndef WF4_PF
#define WF4_PF 1         // problems a wave has requested ahead of the one it works on (1 | 2; measured: 2 is 3-10 % slower, the loads-only floor does not move)
#endif
#ifndef WF4_COAL
#define WF4_COAL 0       // 1: global loads / stores with lane -> (token lane / 4, 16-byte piece lane % 4): a quarter wave touches 4 lines of
#endif                   //    64 contiguous bytes instead of 16 lines of 16 bytes; Q and O pass through LDS for the MFMA layout | 0: lane -> (c, g)
                         //    (measured: no gain -- the loads-only floor is the same 63 us at stage 1; profiles/round5_wattn_fwd4.txt)
#ifndef WF4_DB
#define WF4_DB 0         // 1: two sets of K / V images and region ids, ONE workgroup barrier per problem instead of two
#endif
struct Wf4Smem {
  static constexpr int kBias = WA4_BROWS * WA_BSTR * 4;
  static constexpr int kPlain = WA_LP * WA4_PSTR * 2;          // one plain image (bf16)
  static constexpr int kRow = 64 * 2;                          // P[48][key] of the one-row path
  static constexpr int kSmall = 64;                            // region ids
  static constexpr int kSet = (2 + WF4_COAL) * kPlain + kSmall; // K, V [, Q / O], region ids of one problem
  static constexpr int kTotal = kBias + kRow + (1 + WF4_DB) * kSet;
};
static_assert(WF4_OCC * Wf4Smem::kTotal <= 160 * 1024, "WF4_OCC workgroups per CU");
static_assert(WA_NBIN * 4 <= Wf4Smem::kPlain, "the rel_pos column scratch of the table build must fit the K image");

template <bool MASKED>
__global__ __launch_bounds__(64 * WA_WAVES, WF4_OCC) void wattn_fwd4_kernel(
    const bf16* __restrict__ qkv, bf16* __restrict__ o, float* __restrict__ lse, const float* __restrict__ rel_pos,
    const int64_t* __restrict__ pos, const uint8_t* __restrict__ region, int nbn, int nblk, int xcd_major, int fast, WinGeom g) {
  using T = bf16;
  extern __shared__ __attribute__((aligned(16))) unsigned char wa_smem[];
  float* bias_s = reinterpret_cast<float*>(wa_smem);
  unsigned char* base = wa_smem + Wf4Smem::kBias;
  T* prow = reinterpret_cast<T*>(base);
  base += Wf4Smem::kRow;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63, c_ = lane & 15, g_ = lane >> 4;
  const int lt = WF4_COAL ? lane >> 2 : c_, lp = WF4_COAL ? lane & 3 : g_;      // global-memory side: token of the tile, 16-byte piece
  int blk, h;
  wa_block_map(nblk, g.nH, xcd_major, blk, h);
  const int64_t ld = 3 * (int64_t)g.hd;
  wa_build_bias(bias_s, reinterpret_cast<float*>(base), rel_pos, pos, g.L, g.nH, h, (2 * g.win - 1) * (2 * g.win - 1), WA4_BROWS);
  __syncthreads();                                    // table complete; the relh scratch (the K image) is free again
  const bool w7 = g.L == 49;
  const bool row48 = w7 && (fast & 1);
  // this lane's token of each tile, (ay | ax << 8 | live << 16) inside its window, once per kernel (see WaTok); four scalars, not an
  // array: the tile index is a run-time value and an indexed array would live in scratch.  With the one-row path every lane of
  // the last tile reads token 48 (see the one-wave kernel).
  auto tok_pack = [&](int t4) {
    const int tok = 16 * t4 + lt;
    const bool live = tok < g.L;
    const int tc = (t4 == 3 && row48) ? 48 : (live ? tok : 0);
    const int ay = tc / g.win;
    return ay | ((tc - ay * g.win) << 8) | ((int)live << 16);
  };
  const int tp0 = tok_pack(0), tp1 = tok_pack(1), tp2 = tok_pack(2), tp3 = tok_pack(3);

  // this wave's rows of the problems to come (WF4_PF of them in flight), + wave 0: the region ids
  struct Rows { Vec8<T> q, k, v; int row; uint8_t reg; };
  auto request = [&](int bn, int t, Rows& s) __attribute__((always_inline)) {
    const int n = bn % g.nW, b = g.perm ? g.perm[bn / g.nW] : bn / g.nW;
    const int wi = n / g.nWx, wj = n - wi * g.nWx;
    const int tp = t == 0 ? tp0 : t == 1 ? tp1 : t == 2 ? tp2 : tp3;
    const int ay = tp & 0xff, ax = (tp >> 8) & 0xff;
    int y = wi * g.win + ay + g.shift;
    y -= y >= g.H ? g.H : 0;
    int x = wj * g.win + ax + g.shift;
    x -= x >= g.W ? g.W : 0;
    s.row = (b * g.H + y) * g.W + x;
    const T* p = qkv + (int64_t)s.row * ld + h * WA_D + lp * 8;
    s.q = load8<T>(p);
    s.k = load8<T>(p + g.hd);
    s.v = load8<T>(p + 2 * g.hd);
    if (MASKED && w == 0) s.reg = region[(int64_t)n * 64 + lane];
  };
  // (WF4_PF named sets, loop unrolled by as many: a register copy between sets would wait for the loads in flight)
  Rows rs[WF4_PF];
#pragma unroll
  for (int u = 0; u < WF4_PF; ++u) {
    rs[u].reg = 0;
    if (blk + u * nblk < nbn) request(blk + u * nblk, (w + u) & 3, rs[u]);
  }
  for (int bn = blk, it = 0; bn < nbn;) {
#pragma unroll
  for (int u = 0; u < WF4_PF; ++u, bn += nblk, ++it) {
    if (bn >= nbn) break;
    Rows& nx = rs[u];
    const int t = (w + it) & 3;                       // this wave's token tile of this problem
    const Vec8<T> qg = nx.q, kf = nx.k, vf = nx.v;    // rows of token 16 t + lt, piece lp
    const int row = nx.row;
    const uint8_t myreg = nx.reg;
    const int tok = 16 * t + c_, ltok = 16 * t + lt;
    const bool val = tok < g.L;                       // the query of the MFMA layout is live
    const bool gval = ((t == 0 ? tp0 : t == 1 ? tp1 : t == 2 ? tp2 : tp3) >> 16) != 0;   // the token of the global-memory layout is
    const int prob = bn * g.nH + h;
    unsigned char* set = base + (WF4_DB ? (it & 1) * Wf4Smem::kSet : 0);
    T* pk = reinterpret_cast<T*>(set);                // K [token][d]
    T* pv = reinterpret_cast<T*>(set + Wf4Smem::kPlain);
    T* pq = reinterpret_cast<T*>(set + 2 * Wf4Smem::kPlain);        // (WF4_COAL) Q, later O: a wave touches the rows of its own tile only
    uint8_t* reg_s = set + (2 + WF4_COAL) * Wf4Smem::kPlain;
    if (!WF4_DB && bn != blk) wa_wg_sync();           // every wave is done reading the previous problem's images
    *reinterpret_cast<Vec8<T>*>(pk + ltok * WA4_PSTR + lp * 8) = kf;
    *reinterpret_cast<Vec8<T>*>(pv + ltok * WA4_PSTR + lp * 8) = vf;
    if (WF4_COAL) *reinterpret_cast<Vec8<T>*>(pq + ltok * WA4_PSTR + lp * 8) = qg;
    if (MASKED && w == 0) reg_s[lane] = myreg;
    wa_wg_sync();                                     // (two image sets: a wave is at most one problem ahead of the slowest)
    // global latency of the problems to come runs under this one's arithmetic
    if (bn + WF4_PF * nblk < nbn) request(bn + WF4_PF * nblk, (w + it + WF4_PF) & 3, nx);
    if (WA_ABLATE & 32) {
      if (gval) store8<T>(o + (int64_t)row * g.hd + h * WA_D + lp * 8, qg);
      continue;
    }
    if (16 * t >= g.L) continue;                      // a tile of padding only
    Vec8<T> qf = qg;                                  // MFMA layout: token 16 t + c, channels 8 g .. 8 g + 7
    if (WF4_COAL) qf = *reinterpret_cast<const Vec8<T>*>(pq + tok * WA4_PSTR + g_ * 8);

    // P V for this tile from the P^T fragments of the two key halves; lane (c, g) ends up with O[q = 16 t + c][d = 8 g .. 8 g + 7]
    auto pv_store = [&](const Vec8<T>& p0, const Vec8<T>& p1) {
      f32x4 oacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) mma16(wa4_frag_trp(pv, ks * 32, dt, lane), ks ? p1 : p0, oacc[dt]);
      Vec8<T> ov = wa_out8<T>(oacc[0], oacc[1], 1.f);
      if (WF4_COAL) {                                 // back to the global-memory layout through this tile's rows of the Q image
        wa_wave_sync();
        *reinterpret_cast<Vec8<T>*>(pq + tok * WA4_PSTR + g_ * 8) = ov;
        wa_wave_sync();
        ov = *reinterpret_cast<const Vec8<T>*>(pq + ltok * WA4_PSTR + lp * 8);
      }
      if (gval) store8<T>(o + (int64_t)row * g.hd + h * WA_D + lp * 8, ov);
    };
    auto tile = [&](auto mk) {
      constexpr bool MK = decltype(mk)::value;
      if (t == 3 && row48) {                          // the 49th query as one row (see the one-wave kernel)
        float sv[4], m = -INFINITY;
        const uint8_t r48 = MK ? reg_s[48] : 0;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
          const Vec8<T> kk = *reinterpret_cast<const Vec8<T>*>(pk + (16 * kt + c_) * WA4_PSTR + g_ * 8);
          f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
          mma16(qf, kk, s);
          sv[kt] = s[0] * g.scale + bias_s[48 * WA_BSTR + kt * 16 + c_];
          if (MK && reg_s[kt * 16 + c_] != r48) sv[kt] = -INFINITY;
          m = fmaxf(m, sv[kt]);
        }
        m = group_max16(m);
        float l = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
          sv[kt] = __expf(sv[kt] - m);
          l += sv[kt];
        }
        l = group_sum<16>(l);
        const float inv = 1.f / l;
        if (lane == 0) lse[(int64_t)prob * g.L + 48] = m + __logf(l);
        Vec8<T> one;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
          one.set(0, sv[kt] * inv);
          prow[kt * 16 + c_] = one.v[0];
        }
        wa_wave_sync();
        pv_store(wa_frag_t<T>(prow, g_), wa_frag_t<T>(prow + 32, g_));
        return;
      }
      const int q = val ? tok : 0;                    // (padded queries: any table row, nothing of theirs is stored)
      f32x4 st[4];
      float m = -INFINITY;
      const unsigned rq = MK ? reg_s[tok] * 0x01010101u : 0u;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        const Vec8<T> kk = *reinterpret_cast<const Vec8<T>*>(pk + (16 * kt + c_) * WA4_PSTR + g_ * 8);
        st[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
        mma16(kk, qf, st[kt]);                        // st[kt][r] = S[q = 16 t + c][key = 16 kt + 4 g + r]
        const f32x4 bb = *reinterpret_cast<const f32x4*>(bias_s + q * WA_BSTR + kt * 16 + g_ * 4);
        unsigned rx = 0u;                             // byte r == 0  <=>  key r is in the query's region
        if (MK) rx = *reinterpret_cast<const unsigned*>(reg_s + kt * 16 + g_ * 4) ^ rq;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (kt == 3 && r > 0 && w7) continue;       // always-padded keys of 7 x 7 windows: p = 0
          float sv = st[kt][r] * g.scale + bb[r];
          if (MK && (rx & (0xffu << (8 * r))) != 0u) sv = -INFINITY;
          st[kt][r] = sv;
          m = fmaxf(m, sv);
        }
      }
      m = fmaxf(m, shfl_xor_f(m, 16));
      m = fmaxf(m, shfl_xor_f(m, 32));
      float l = 0.f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (kt == 3 && r > 0 && w7) { st[kt][r] = 0.f; continue; }
          st[kt][r] = __expf(st[kt][r] - m);
          l += st[kt][r];
        }
      l += shfl_xor_f(l, 16);
      l += shfl_xor_f(l, 32);
      const float inv = 1.f / l;
      if (val && g_ == 0) lse[(int64_t)prob * g.L + tok] = m + __logf(l);
      pv_store(wa_frag_acc<T>(st[0] * inv, st[1] * inv), wa_frag_acc<T>(st[2] * inv, st[3] * inv));
    };
    bool masked = MASKED;
    if (MASKED && (fast & 2)) masked = !wa_uniform(reg_s[lane], lane < g.L);
    if (masked)
      tile(std::true_type{});
    else
      tile(std::false_type{});
  }
  }
}

static int win_geom(WinGeom& g, int L, int nH, int H, int W, int win, int shift) {
  if (win <= 0 || H % win || W % win || L != win * win || L > WA_LP) return VTX_ERR_SHAPE;
  if ((2 * win - 1) * (2 * win - 1) > WA_NBIN) return VTX_ERR_SHAPE;
  g.L = L; g.nH = nH; g.hd = nH * WA_D; g.H = H; g.W = W; g.win = win;
  g.nWx = W / win; g.nW = (H / win) * (W / win); g.shift = shift ? win / 2 : 0;
  g.scale = 1.0f / sqrtf((float)WA_D);
  g.perm = nullptr;
  return VTX_OK;
}

// Persistent grid: `cap` waves are resident on the chip (forward 4 per SIMD, backward 2 per SIMD -- VGPR-bound);
// every wave of a head gets the same number of (image, window) pairs (+-1), waves come in workgroups of 4, and the
// grid never exceeds what is resident at once.  Returns workgroups per head.
static int wattn_blocks(int nbn, int nH, int cap) {
  int per_head = cap / nH;
  if (per_head < WA_WAVES) per_head = WA_WAVES;
  const int ppw = (nbn + per_head - 1) / per_head;            // problems per wave
  const int waves = (nbn + ppw - 1) / ppw;
  int blocks = (waves + WA_WAVES - 1) / WA_WAVES;
  // a multiple of 8 blocks per head makes the XCD-major head-fastest mapping (wa_block_map) a bijection; the few extra
  // workgroups find no work past nbn
  if (blocks >= 8 && vtx_opt(VTX_OPT_WATTN_XCD_MAJOR)) blocks = (blocks + 7) / 8 * 8;
  return blocks;
}
static int wattn_xcd_major(int nblk) { return (nblk % 8 == 0 && vtx_opt(VTX_OPT_WATTN_XCD_MAJOR)) ? 1 : 0; }
static int wattn_fwd_blocks(int nbn, int nH) {
  const int cap = vtx_opt(VTX_OPT_WATTN_FWD_WAVES);
  return wattn_blocks(nbn, nH, cap > 0 ? cap : 4096);
}
static int wattn_bwd_blocks(int nbn, int nH) {
  const int cap = vtx_opt(VTX_OPT_WATTN_BWD_WAVES);
  return wattn_blocks(nbn, nH, cap > 0 ? cap : 2048);
}
// the four-wave backward: workgroups per head = problems in flight per head (3 workgroups per CU resident)
static int wattn_bwd4_blocks(int nbn, int nH) {
  int per_head = 256 * WA4_OCC / nH;
  if (per_head < 1) per_head = 1;
  const int ppw = (nbn + per_head - 1) / per_head;            // problems per workgroup
  int blocks = (nbn + ppw - 1) / ppw;
  if (blocks >= 8 && vtx_opt(VTX_OPT_WATTN_XCD_MAJOR)) blocks = (blocks + 7) / 8 * 8;
  return blocks;
}
static int wattn_fwd4_blocks(int nbn, int nH) {
  int per_head = 256 * WF4_OCC / nH;
  if (per_head < 1) per_head = 1;
  const int ppw = (nbn + per_head - 1) / per_head;            // problems per workgroup
  int blocks = (nbn + ppw - 1) / ppw;
  if (blocks >= 8 && vtx_opt(VTX_OPT_WATTN_XCD_MAJOR)) blocks = (blocks + 7) / 8 * 8;
  return blocks;
}
// rows of rel_pos-gradient partials a backward launch leaves ([rows][172 * nH]): whichever kernel runs fills them all
static int wattn_bwd_rows(int nbn, int nH) {
  const int a = wattn_bwd_blocks(nbn, nH) * WA_WAVES, b = wattn_bwd4_blocks(nbn, nH);
  return a > b ? a : b;
}

template <typename K> static int wa_smem_attr(K kern, int bytes) {
  if (bytes > 64 * 1024 &&
      hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess)
    return VTX_ERR_LAUNCH;
  return VTX_OK;
}

template <typename T, bool MASKED>
static int wattn_fwd_launch(const void* qkv, void* o, float* lse, const float* rel_pos, const int64_t* pos,
                            const uint8_t* region, int nbn, const WinGeom& g, hipStream_t st) {
  if constexpr (sizeof(T) == 2) {
    // four waves per problem: measured faster only with many problems per head (Swin-S stage 1, 8 192: 70 vs 75-76 us; stage 3, 512: 23.0-23.8
    // vs 22.5-23.0 us; profiles/round5_wattn_fwd4.txt)
    const int f4 = vtx_opt(VTX_OPT_WATTN_FWD4);
    if (f4 >= 2 || (f4 == 1 && nbn >= 4096)) {
      const int nb4 = wattn_fwd4_blocks(nbn, g.nH);
      hipLaunchKernelGGL(wattn_fwd4_kernel<MASKED>, dim3(nb4 * g.nH), dim3(64 * WA_WAVES), Wf4Smem::kTotal, st,
                         (const bf16*)qkv, (bf16*)o, lse, rel_pos, pos, region, nbn, nb4, wattn_xcd_major(nb4),
                         vtx_opt(VTX_OPT_WATTN_FAST), g);
      return vtx_check_launch();
    }
  }
  auto kern = wattn_fwd_kernel<T, MASKED>;
  int rc = wa_smem_attr(kern, WaSmem<T>::kFwd);
  if (rc) return rc;
  const int nblk = wattn_fwd_blocks(nbn, g.nH);
  hipLaunchKernelGGL(kern, dim3(nblk * g.nH), dim3(64 * WA_WAVES), WaSmem<T>::kFwd, st,
                     (const T*)qkv, (T*)o, lse, rel_pos, pos, region, nbn, nblk, wattn_xcd_major(nblk), vtx_opt(VTX_OPT_WATTN_FAST), g);
  return vtx_check_launch();
}

template <typename T, bool MASKED>
static int wattn_bwd_launch(const void* qkv, const void* o, const void* dout, const float* lse, const float* rel_pos,
                            const int64_t* pos, const uint8_t* region, void* dqkv, float* part, const int* inv_cells,
                            int inv_count, int nbn, const WinGeom& g, hipStream_t st) {
  const int rows = wattn_bwd_rows(nbn, g.nH);
  if constexpr (sizeof(T) == 2) {
    if (inv_cells != nullptr && vtx_opt(VTX_OPT_WATTN_BWD4) != 0) {   // four waves per problem
      auto kern4 = wattn_bwd4_kernel<MASKED>;
      int rc4 = wa_smem_attr(kern4, Wa4Smem::kTotal);
      if (rc4) return rc4;
      const int nb4 = wattn_bwd4_blocks(nbn, g.nH);
      hipLaunchKernelGGL(kern4, dim3(nb4 * g.nH), dim3(64 * WA_WAVES), Wa4Smem::kTotal, st,
                         (const bf16*)qkv, (const bf16*)o, (const bf16*)dout, lse, rel_pos, pos, region, (bf16*)dqkv, part,
                         inv_cells, inv_count, nbn, nb4, rows, wattn_xcd_major(nb4), g);
      return vtx_check_launch();
    }
  }
  auto kern = wattn_bwd_kernel<T, MASKED>;
  int rc = wa_smem_attr(kern, WaSmem<T>::kBwd);
  if (rc) return rc;
  const int nblk = wattn_bwd_blocks(nbn, g.nH);
  if (rows > nblk * WA_WAVES &&
      hipMemsetAsync(part + (size_t)nblk * WA_WAVES * WA_NBIN * g.nH, 0,
                     (size_t)(rows - nblk * WA_WAVES) * WA_NBIN * g.nH * sizeof(float), st) != hipSuccess)
    return VTX_ERR_LAUNCH;
  hipLaunchKernelGGL(kern, dim3(nblk * g.nH), dim3(64 * WA_WAVES), WaSmem<T>::kBwd, st,
                     (const T*)qkv, (const T*)o, (const T*)dout, lse, rel_pos, pos, region, (T*)dqkv, part, inv_cells,
                     inv_count, nbn, nblk, wattn_xcd_major(nblk), g);
  return vtx_check_launch();
}

extern "C" {

/* Window attention (head dim 32, window <= 64 tokens) -- the Swin path (reference swin_transformer.py:103-160).
 * rel_pos [(2 win - 1)^2][nH] fp32 parameter, pos [L][L] int64 buffer; region: NULL for un-shifted layers, else
 * [nW][64] uint8 region ids with local_mask[n][a][b] == (region[n][a] != region[n][b]) (vtx.tables.mask_regions
 * derives and verifies them from the module's local_mask buffer). */
int vtx_wattn_fwd(const void* qkv, void* o, float* lse, const float* rel_pos, const int64_t* pos,
                  const uint8_t* region, int B, int L, int nH, int H, int W, int win, int shift, int dtype,
                  void* stream) {
  if (!qkv || !o || !lse || !rel_pos || !pos) return VTX_ERR_NULL;
  WinGeom g;
  int rc = win_geom(g, L, nH, H, W, win, shift);
  if (rc) return rc;
  const int nbn = B * g.nW;
  if (nbn <= 0) return VTX_OK;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VTX_BF16)
    return region ? wattn_fwd_launch<bf16, true>(qkv, o, lse, rel_pos, pos, region, nbn, g, st)
                  : wattn_fwd_launch<bf16, false>(qkv, o, lse, rel_pos, pos, region, nbn, g, st);
  if (dtype == VTX_F32)
    return region ? wattn_fwd_launch<float, true>(qkv, o, lse, rel_pos, pos, region, nbn, g, st)
                  : wattn_fwd_launch<float, false>(qkv, o, lse, rel_pos, pos, region, nbn, g, st);
  return VTX_ERR_DTYPE;
}

/* The same over Bk <= B images only: the b-th one is image perm[b] (perm [B] int32 on the device: the kept samples of a
 * stochastic-depth branch first); lse is indexed by b, i.e. private to the (forward, backward) pair that shares perm. */
int vtx_wattn_fwd_mapped(const void* qkv, void* o, float* lse, const float* rel_pos, const int64_t* pos,
                         const uint8_t* region, const int* perm, int Bk, int L, int nH, int H, int W, int win, int shift,
                         int dtype, void* stream) {
  if (!qkv || !o || !lse || !rel_pos || !pos || !perm) return VTX_ERR_NULL;
  WinGeom g;
  int rc = win_geom(g, L, nH, H, W, win, shift);
  if (rc) return rc;
  g.perm = perm;
  const int nbn = Bk * g.nW;
  if (nbn <= 0) return VTX_OK;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VTX_BF16)
    return region ? wattn_fwd_launch<bf16, true>(qkv, o, lse, rel_pos, pos, region, nbn, g, st)
                  : wattn_fwd_launch<bf16, false>(qkv, o, lse, rel_pos, pos, region, nbn, g, st);
  if (dtype == VTX_F32)
    return region ? wattn_fwd_launch<float, true>(qkv, o, lse, rel_pos, pos, region, nbn, g, st)
                  : wattn_fwd_launch<float, false>(qkv, o, lse, rel_pos, pos, region, nbn, g, st);
  return VTX_ERR_DTYPE;
}

/* partial rows ([rows][172 * nH] fp32, the first (2 win - 1)^2 * nH columns used) vtx_wattn_bwd leaves in its workspace when
 * drel_pos == NULL (the caller reduces them with vtx_colreduce_multi: C = (2 win - 1)^2 * nH, ld = 172 * nH) */
int vtx_wattn_bwd_parts(int B, int nH, int H, int W, int win) {
  if (win <= 0 || H % win || W % win) return 0;
  return wattn_bwd_rows(B * (H / win) * (W / win), nH);
}
int vtx_wattn_bwd_part_ld(int nH) { return WA_NBIN * nH; }

size_t vtx_wattn_bwd_workspace(int B, int nH, int H, int W, int win) {
  const int nbn = B * (H / win) * (W / win);
  return (size_t)wattn_bwd_rows(nbn, nH) * nH * WA_NBIN * sizeof(float);
}

/* inv_cells [inv_count][(2 win - 1)^2] int32 (device) or NULL: the inverse of pos -- inv_cells[t][b] = the t-th cell
 * q * 64 + key (row stride 64) with pos[q][key] == b in ascending (q, key) order, padded with the cell L * 64;
 * inv_count = the largest bin.  With it the rel_pos gradient is gathered per bin from LDS; without it is scattered with
 * LDS atomics (slower, same result up to fp32 summation order). */
int vtx_wattn_bwd(const void* qkv, const void* o, const void* dout, const float* lse, const float* rel_pos,
                  const int64_t* pos, const uint8_t* region, void* dqkv, float* drel_pos, void* workspace,
                  size_t ws_bytes, const int* inv_cells, int inv_count, int B, int L, int nH, int H, int W, int win,
                  int shift, int dtype, void* stream) {
  if (!qkv || !o || !dout || !lse || !rel_pos || !pos || !dqkv || !workspace) return VTX_ERR_NULL;   // drel_pos NULL: deferred reduce
  WinGeom g;
  int rc = win_geom(g, L, nH, H, W, win, shift);
  if (rc) return rc;
  const int nbn = B * g.nW;
  if (nbn <= 0) return VTX_OK;
  if (ws_bytes < vtx_wattn_bwd_workspace(B, nH, H, W, win)) return VTX_ERR_WORKSPACE;
  if (inv_cells != nullptr && (inv_count <= 0 || inv_count > L * L)) return VTX_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)workspace;
  if (dtype == VTX_BF16)
    rc = region ? wattn_bwd_launch<bf16, true>(qkv, o, dout, lse, rel_pos, pos, region, dqkv, part, inv_cells, inv_count, nbn, g, st)
                : wattn_bwd_launch<bf16, false>(qkv, o, dout, lse, rel_pos, pos, region, dqkv, part, inv_cells, inv_count, nbn, g, st);
  else if (dtype == VTX_F32)
    rc = region ? wattn_bwd_launch<float, true>(qkv, o, dout, lse, rel_pos, pos, region, dqkv, part, inv_cells, inv_count, nbn, g, st)
                : wattn_bwd_launch<float, false>(qkv, o, dout, lse, rel_pos, pos, region, dqkv, part, inv_cells, inv_count, nbn, g, st);
  else return VTX_ERR_DTYPE;
  if (rc || drel_pos == nullptr) return rc;              // deferred: partials stay in the workspace
  const int ntab = (2 * win - 1) * (2 * win - 1);
  const int nwaves = wattn_bwd_rows(nbn, nH);
  hipLaunchKernelGGL(colreduce_kernel, colreduce_grid(ntab * nH), dim3(1024), 0, st, (const float*)part, drel_pos,
                     (float*)nullptr, nwaves, ntab * nH, WA_NBIN * nH);
  return vtx_check_launch();
}

/* Backward over Bk <= B images (see vtx_wattn_fwd_mapped); the rel_pos-gradient partials stay in the workspace
 * (vtx_wattn_bwd_parts(Bk, ...) rows: deferred reduce).  dqkv rows of the other images are not written. */
int vtx_wattn_bwd_mapped(const void* qkv, const void* o, const void* dout, const float* lse, const float* rel_pos,
                         const int64_t* pos, const uint8_t* region, void* dqkv, void* workspace, size_t ws_bytes,
                         const int* inv_cells, int inv_count, const int* perm, int Bk, int L, int nH, int H, int W, int win,
                         int shift, int dtype, void* stream) {
  if (!qkv || !o || !dout || !lse || !rel_pos || !pos || !dqkv || !workspace || !perm) return VTX_ERR_NULL;
  WinGeom g;
  int rc = win_geom(g, L, nH, H, W, win, shift);
  if (rc) return rc;
  g.perm = perm;
  const int nbn = Bk * g.nW;
  if (nbn <= 0) return VTX_ERR_SHAPE;
  if (ws_bytes < vtx_wattn_bwd_workspace(Bk, nH, H, W, win)) return VTX_ERR_WORKSPACE;
  if (inv_cells != nullptr && (inv_count <= 0 || inv_count > L * L)) return VTX_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)workspace;
  if (dtype == VTX_BF16)
    return region ? wattn_bwd_launch<bf16, true>(qkv, o, dout, lse, rel_pos, pos, region, dqkv, part, inv_cells, inv_count, nbn, g, st)
                  : wattn_bwd_launch<bf16, false>(qkv, o, dout, lse, rel_pos, pos, region, dqkv, part, inv_cells, inv_count, nbn, g, st);
  if (dtype == VTX_F32)
    return region ? wattn_bwd_launch<float, true>(qkv, o, dout, lse, rel_pos, pos, region, dqkv, part, inv_cells, inv_count, nbn, g, st)
                  : wattn_bwd_launch<float, false>(qkv, o, dout, lse, rel_pos, pos, region, dqkv, part, inv_cells, inv_count, nbn, g, st);
  return VTX_ERR_DTYPE;
}

}  // extern "C"
