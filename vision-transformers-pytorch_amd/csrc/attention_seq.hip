// Global (ViT) attention for gfx950, bf16, head dim 64, sequences of <= 224 tokens (ViT-S/16: L = 197; 96^2 crops: 37).
//
// Replaces reference models/vit.py:30-42 per (image, head): S = q k^T / sqrt(d), softmax, O = P v, reading
// q/k/v from the QKV projection output [tokens, 3*h*64] and writing [tokens, h*64]; scores stay in registers.
// One workgroup (4 waves) per (image, head):
//   * K and V of the head (L x 64 each) are brought into LDS ONCE with global_load_lds_dwordx4 as plain
//     row-major images (128-byte rows, 16-byte chunk q of row r in slot q ^ (r & 7)); the same image serves
//     the row-contiguous MFMA operands (ds_read_b128) and, through ds_read_b64_tr_b16, the operands that are
//     contracted over tokens (V in P.V; K, Q, dO in the backward) -- no transposed copies, no ds_write;
//   * each wave owns PAIRS of 16-query tiles so every K / V fragment read from LDS feeds two MFMAs;
//   * swapped product S^T = K Q^T: a lane holds 4 keys of every key tile for one query, so softmax is
//     lane-local + two shuffles and P is already the A operand of P.V;
//   * backward: phase A (waves <-> query-tile pairs, K/V in LDS) gives dQ, phase B (waves <-> key-tile pairs,
//     Q/dO re-staged into the same LDS) gives dK, dV; P is recomputed from the saved log-sum-exp and
//     D = rowsum(dO o O).
// fp32 (parity mode) and other head dims use the generic kernels in attention.hip.
#include <stdlib.h>

#include <type_traits>

// (round 6) the o / dqkv stores of this file are non-temporal: -0.4 % on the ViT-S/16 step, same-box A/B; the same switch on the other
// kernel files measured neutral or slower (profiles/round6_nt_store_screen.txt)
#ifndef VTX_NT_STORE8
#define VTX_NT_STORE8 1
#endif
#include "options.h"
#include "vtx_common.h"

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// phase ablation of the backward for timing (separate builds; 0 in the library): 1 no phase A (dQ) | 2 no phase B (dK, dV)
#ifndef SA_ABLATE
#define SA_ABLATE 0
#endif
#define SA_D 64
#define SA_ROWB 128

__device__ __attribute__((aligned(256))) unsigned int sa_zero_row[64];   // 256 zero bytes

struct SeqGeom { int L, nH, hd; float scale; const int* perm; };   // perm: the b-th image worked on is image perm[b] (null: identity)

// DMA `rows` rows x 64 bf16 of one head (token rows base .. ) into a swizzled row-major LDS image
template <int LP, int NW>
__device__ __forceinline__ void sa_stage(unsigned char* lds, const bf16* src, int64_t ld, int L, int wave, int lane) {
  const int lr = lane >> 3, slot = lane & 7;
  const bf16* zero = reinterpret_cast<const bf16*>(sa_zero_row);
  for (int pc = wave; pc < LP / 8; pc += NW) {
    const int r = pc * 8 + lr;
    const int q = slot ^ (r & 7);
    const bf16* s = r < L ? src + (int64_t)r * ld + (q << 3) : zero + (q << 3);
    __builtin_amdgcn_global_load_lds((gbl_void_t*)s, (lds_void_t*)(lds + pc * 8 * SA_ROWB), 16, 0, 0);
  }
}

// row-contiguous fragment: 8 d-slots (32*ds + 8g ..) of token row r
__device__ __forceinline__ Vec8<bf16> sa_frag_row(const unsigned char* img, int r, int ds, int g) {
  return load8<bf16>(reinterpret_cast<const bf16*>(img + r * SA_ROWB + (((ds * 4 + g) ^ (r & 7)) << 4)));
}
// token-contracted fragment for lane (c, g): k-slots j < 4 <-> tokens t0 + 4g + j, j >= 4 <-> t0 + 16 + 4g + (j-4)
// (the slot <-> token map of fragments built from accumulator pairs), column d = d0 + c
__device__ __forceinline__ Vec8<bf16> sa_frag_tr(const unsigned char* img, int t0, int d0, int lane) {
  const int p = lane & 15, g = lane >> 4;
  const int n = d0 + ((p & 3) << 2);
  s16x4 v[2];
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int r = t0 + half * 16 + g * 4 + (p >> 2);
    const unsigned char* a = img + r * SA_ROWB + ((((n >> 3) ^ (r & 7))) << 4) + ((n & 7) << 1);
    v[half] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a);
  }
  s16x8 w = __builtin_shufflevector(v[0], v[1], 0, 1, 2, 3, 4, 5, 6, 7);   // whole-vector cast only (see gemm_wgrad_glds.hip)
  Vec8<bf16> f;
  f.v = __builtin_bit_cast(bf16x8, w);
  return f;
}
// Same, as the A operand of a TRANSPOSED output product (O^T = V^T P^T etc.) with the row permutation that makes
// the result store-friendly: A row c <-> column d = 32 dp + 8 (c >> 2) + 4 dtl + (c & 3), so that accumulator
// (dp, dtl) register r of lane (c, g) is the output of token c for d = 32 dp + 8 g + 4 dtl + r -- the pair dtl = 0, 1
// gives every lane 8 contiguous channels = one 16-byte global store (the 4 x 16 transpose-read block of each
// 16-lane group takes its four 4-column groups from lanes p & 3 -- any four column bases are allowed).
__device__ __forceinline__ Vec8<bf16> sa_frag_trp(const unsigned char* img, int t0, int dp, int dtl, int lane) {
  const int p = lane & 15, g = lane >> 4;
  const int n = 32 * dp + ((p & 3) << 3) + 4 * dtl;
  s16x4 v[2];
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int r = t0 + half * 16 + g * 4 + (p >> 2);
    const unsigned char* a = img + r * SA_ROWB + ((((n >> 3) ^ (r & 7))) << 4) + ((n & 7) << 1);
    v[half] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a);
  }
  s16x8 w = __builtin_shufflevector(v[0], v[1], 0, 1, 2, 3, 4, 5, 6, 7);
  Vec8<bf16> f;
  f.v = __builtin_bit_cast(bf16x8, w);
  return f;
}
__device__ __forceinline__ Vec8<bf16> sa_out8(const f32x4& a0, const f32x4& a1, float scale) {
  Vec8<bf16> f;
#pragma unroll
  for (int j = 0; j < 4; ++j) { f.set(j, a0[j] * scale); f.set(4 + j, a1[j] * scale); }
  return f;
}
__device__ __forceinline__ Vec8<bf16> sa_frag_acc(const f32x4& lo, const f32x4& hi) {
  Vec8<bf16> f;
#pragma unroll
  for (int j = 0; j < 4; ++j) { f.set(j, lo[j]); f.set(4 + j, hi[j]); }
  return f;
}
__device__ __forceinline__ Vec8<bf16> sa_gload(const bf16* p, bool valid) {
  return valid ? load8<bf16>(p) : vec8_zero<bf16>();
}

// ------------------------------------------------------------------------------------------------ forward
// TP = query tiles a wave works on at a time (2: every K / V fragment read feeds two MFMAs | 1), NW = waves per workgroup
#ifndef SA_DQ_INIT
#define SA_DQ_INIT 1     // 0: the backward as it was before the fix (tests/test_gpu_lds_poison.py must then fail at L = 197)
#endif
#ifndef SA_SKIP_DEAD
#define SA_SKIP_DEAD 1   // 0: the round-4 forward, which computes the all-padding last tile too (tools/r5/sattn_dead_tile_check.py compares)
#endif
template <int NKT, int TP, int NW>
__global__ __launch_bounds__(64 * NW, 2) void sattn_fwd_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ o,
                                                       float* __restrict__ lse, SeqGeom g) {
  constexpr int LP = NKT * 16, IMG = LP * SA_ROWB, KSN = NKT / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char sa_smem[];
  unsigned char* ks = sa_smem;            // K image
  unsigned char* vs = sa_smem + IMG;      // V image
  const int prob = blockIdx.x;
  const int h = prob % g.nH, b = g.perm ? g.perm[prob / g.nH] : prob / g.nH;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c_ = lane & 15, g_ = lane >> 4;
  const int64_t ld = 3 * (int64_t)g.hd;
  const bf16* base = qkv + (int64_t)b * g.L * ld + h * SA_D;

  sa_stage<LP, NW>(ks, base + g.hd, ld, g.L, wave, lane);
  sa_stage<LP, NW>(vs, base + 2 * g.hd, ld, g.L, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int nqt = (g.L + 15) >> 4;
  // the LAST key tile of the instantiation holds no key at all for L <= 16 (NKT - 1) (ViT-S/16: L = 197 -> 13 live tiles of 14; the 96 x 96
  // DINO crops: L = 37 -> 3 of 4): its scores are exact zeros after the softmax -- skipped (round 5; same bits, 1 / NKT of the score work)
  const bool last_dead = SA_SKIP_DEAD && nqt < NKT;
  for (int qp = wave; qp * TP < nqt; qp += NW) {
    Vec8<bf16> qf[TP][2];
    bool qv[TP];
#pragma unroll
    for (int t = 0; t < TP; ++t) {
      const int q = (qp * TP + t) * 16 + c_;
      qv[t] = q < g.L;
#pragma unroll
      for (int ds = 0; ds < 2; ++ds) qf[t][ds] = sa_gload(base + (int64_t)(qv[t] ? q : 0) * ld + ds * 32 + g_ * 8, qv[t]);
    }
    f32x4 st[TP][NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
      for (int t = 0; t < TP; ++t) st[t][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (kt == NKT - 1 && last_dead) continue;
#pragma unroll
      for (int ds = 0; ds < 2; ++ds) {
        Vec8<bf16> kf = sa_frag_row(ks, kt * 16 + c_, ds, g_);
#pragma unroll
        for (int t = 0; t < TP; ++t) mma16(kf, qf[t][ds], st[t][kt]);   // st[t][kt][r] = S[q = 16 (TP qp + t) + c][key = 16 kt + 4 g + r]
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    float inv[TP];
#pragma unroll
    for (int t = 0; t < TP; ++t) {
      float m = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        if (kt == NKT - 1 && last_dead) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float s = (kt * 16 + g_ * 4 + r) < g.L ? st[t][kt][r] * g.scale : -INFINITY;
          st[t][kt][r] = s;
          m = fmaxf(m, s);
        }
      }
      m = fmaxf(m, shfl_xor_f(m, 16));
      m = fmaxf(m, shfl_xor_f(m, 32));
      float l = 0.f;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        if (kt == NKT - 1 && last_dead) continue;       // (st[t][NKT - 1] is still the zero it was initialised with)
#pragma unroll
        for (int r = 0; r < 4; ++r) { st[t][kt][r] = __expf(st[t][kt][r] - m); l += st[t][kt][r]; }
      }
      l += shfl_xor_f(l, 16);
      l += shfl_xor_f(l, 32);
      inv[t] = 1.f / l;
      const int q = (qp * TP + t) * 16 + c_;
      if (qv[t] && g_ == 0) lse[(int64_t)prob * g.L + q] = m + __logf(l);
    }
    f32x4 oacc[TP][4];
#pragma unroll
    for (int t = 0; t < TP; ++t)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) oacc[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k2 = 0; k2 < KSN; ++k2) {
      Vec8<bf16> pf[TP];
#pragma unroll
      for (int t = 0; t < TP; ++t) pf[t] = sa_frag_acc(st[t][2 * k2] * inv[t], st[t][2 * k2 + 1] * inv[t]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        Vec8<bf16> vf = sa_frag_trp(vs, k2 * 32, dt >> 1, dt & 1, lane);
#pragma unroll
        for (int t = 0; t < TP; ++t) mma16(vf, pf[t], oacc[t][dt]);   // oacc[t][2 dp + dtl][r] = O[q = 16 (TP qp + t) + c][32 dp + 8 g + 4 dtl + r]
      }
      __builtin_amdgcn_sched_barrier(0);     // keep the unrolled iterations apart (register pressure)
    }
#pragma unroll
    for (int t = 0; t < TP; ++t) {
      const int q = (qp * TP + t) * 16 + c_;
      if (qv[t]) {
        bf16* op = o + ((int64_t)b * g.L + q) * g.hd + h * SA_D + g_ * 8;
        store8<bf16>(op, sa_out8(oacc[t][0], oacc[t][1], 1.f));
        store8<bf16>(op + 32, sa_out8(oacc[t][2], oacc[t][3], 1.f));
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward
template <int NKT, int TP, int NW>
__global__ __launch_bounds__(64 * NW, 2) void sattn_bwd_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ oin,
                                                       const bf16* __restrict__ dout, const float* __restrict__ lse,
                                                       bf16* __restrict__ dqkv, SeqGeom g) {
  constexpr int LP = NKT * 16, IMG = LP * SA_ROWB, KSN = NKT / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char sa_smem[];
  unsigned char* im0 = sa_smem;           // K (phase A) then Q (phase B)
  unsigned char* im1 = sa_smem + IMG;     // V (phase A) then dO (phase B)
  float* dq_s = reinterpret_cast<float*>(sa_smem + 2 * IMG);   // D[q]
  float* lse_s = dq_s + LP;
  const int prob = blockIdx.x;
  const int h = prob % g.nH, b = g.perm ? g.perm[prob / g.nH] : prob / g.nH;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c_ = lane & 15, g_ = lane >> 4;
  const int64_t ld = 3 * (int64_t)g.hd;
  const bf16* qb = qkv + (int64_t)b * g.L * ld + h * SA_D;
  const bf16* ob = oin + (int64_t)b * g.L * g.hd + h * SA_D;
  const bf16* dob = dout + (int64_t)b * g.L * g.hd + h * SA_D;
  bf16* dqb = dqkv + (int64_t)b * g.L * ld + h * SA_D;
  const int nt = (g.L + 15) >> 4;

  sa_stage<LP, NW>(im0, qb + g.hd, ld, g.L, wave, lane);
  sa_stage<LP, NW>(im1, qb + 2 * g.hd, ld, g.L, wave, lane);
  // lse * log2 e, +inf for padded query rows: exp2(. - inf) = 0 masks them in both phases without a select
  for (int i = threadIdx.x; i < LP; i += 64 * NW) {
    lse_s[i] = i < g.L ? lse[(int64_t)prob * g.L + i] * 1.4426950408889634f : INFINITY;
    // D[q] of a query tile WITHOUT queries (L = 197: rows 208 .. 223) is never written by phase A, and phase B multiplies (dP - D) of those
    // rows by p = 0: whatever the previous workgroup left in this LDS would reach dK as 0 x NaN (round 5: seen once, on one box, as
    // non-finite dqkv of the very first launch of a process; tests/test_gpu_lds_poison.py)
    if (SA_DQ_INIT && i >= nt * 16) dq_s[i] = 0.f;
  }
  const float sl = g.scale * 1.4426950408889634f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---------------- phase A: wave <-> pairs of query tiles; dQ = scale * dS K
  for (int qp = wave; qp * TP < nt && !(SA_ABLATE & 1); qp += NW) {
    Vec8<bf16> qf[TP][2], dof[TP][2];
    bool qv[TP];
    float dsum[TP], lq[TP];
#pragma unroll
    for (int t = 0; t < TP; ++t) {
      const int q = (qp * TP + t) * 16 + c_;
      qv[t] = q < g.L;
      const int qq = qv[t] ? q : 0;
      float s = 0.f;
#pragma unroll
      for (int ds = 0; ds < 2; ++ds) {
        qf[t][ds] = sa_gload(qb + (int64_t)qq * ld + ds * 32 + g_ * 8, qv[t]);
        dof[t][ds] = sa_gload(dob + (int64_t)qq * g.hd + ds * 32 + g_ * 8, qv[t]);
        Vec8<bf16> of = sa_gload(ob + (int64_t)qq * g.hd + ds * 32 + g_ * 8, qv[t]);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += of.get(e) * dof[t][ds].get(e);
      }
      s += shfl_xor_f(s, 16);
      s += shfl_xor_f(s, 32);
      dsum[t] = s;
      lq[t] = lse_s[q];                             // log2 domain; +inf on padded rows (q < LP always)
      if (g_ == 0) dq_s[q] = qv[t] ? s : 0.f;       // q < LP always
    }
    f32x4 dqacc[TP][4];
#pragma unroll
    for (int t = 0; t < TP; ++t)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) dqacc[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // (the key tile without keys that the forward skips is computed here: skipping it behind a run-time branch costs the ViT-S/16 backward
    //  more than the tile saves -- 128 vs 124 us -- and peeling the last step into a second instantiation of the loop body costs registers:
    //  148 vs 119 us; tools/r5/sattn_dead_tile_check.py)
#pragma unroll 1
    for (int k2 = 0; k2 < KSN; ++k2) {
      f32x4 dsv[TP][2];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int kt = 2 * k2 + half;
        f32x4 pt[TP], dpt[TP];
#pragma unroll
        for (int t = 0; t < TP; ++t) { pt[t] = f32x4{0.f, 0.f, 0.f, 0.f}; dpt[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ds = 0; ds < 2; ++ds) {
          Vec8<bf16> kf = sa_frag_row(im0, kt * 16 + c_, ds, g_);
          Vec8<bf16> vf = sa_frag_row(im1, kt * 16 + c_, ds, g_);
#pragma unroll
          for (int t = 0; t < TP; ++t) { mma16(kf, qf[t][ds], pt[t]); mma16(vf, dof[t][ds], dpt[t]); }
        }
        // padded keys have zero K / V rows (their dS only has to stay finite): masked in the straddling tile only
        const bool edge = kt * 16 + 16 > g.L;           // uniform
#pragma unroll
        for (int t = 0; t < TP; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float p = __builtin_amdgcn_exp2f(fmaf(pt[t][r], sl, -lq[t]));
            if (edge && (kt * 16 + g_ * 4 + r) >= g.L) p = 0.f;
            dsv[t][half][r] = p * (dpt[t][r] - dsum[t]);
          }
      }
      Vec8<bf16> dsf[TP];
#pragma unroll
      for (int t = 0; t < TP; ++t) dsf[t] = sa_frag_acc(dsv[t][0], dsv[t][1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        Vec8<bf16> kf = sa_frag_trp(im0, k2 * 32, dt >> 1, dt & 1, lane);
#pragma unroll
        for (int t = 0; t < TP; ++t) mma16(kf, dsf[t], dqacc[t][dt]);   // dqacc[t][2 dp + dtl][r] = dQ[q = .. + c][32 dp + 8 g + 4 dtl + r] / scale
      }
    }
#pragma unroll
    for (int t = 0; t < TP; ++t) {
      const int q = (qp * TP + t) * 16 + c_;
      if (qv[t]) {
        bf16* p = dqb + (int64_t)q * ld + g_ * 8;
        store8<bf16>(p, sa_out8(dqacc[t][0], dqacc[t][1], g.scale));
        store8<bf16>(p + 32, sa_out8(dqacc[t][2], dqacc[t][3], g.scale));
      }
    }
  }
  __syncthreads();                                   // K / V images are dead; dq_s complete
  sa_stage<LP, NW>(im0, qb, ld, g.L, wave, lane);                       // Q
  sa_stage<LP, NW>(im1, dob, (int64_t)g.hd, g.L, wave, lane);           // dO
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---------------- phase B: wave <-> pairs of key tiles; dV = P^T dO, dK = scale * dS^T Q
  for (int kp = wave; kp * TP < nt && !(SA_ABLATE & 2); kp += NW) {
    Vec8<bf16> kf[TP][2], vf[TP][2];
    bool kv[TP];
#pragma unroll
    for (int t = 0; t < TP; ++t) {
      const int key = (kp * TP + t) * 16 + c_;
      kv[t] = key < g.L;
      const int kk = kv[t] ? key : 0;
#pragma unroll
      for (int ds = 0; ds < 2; ++ds) {
        kf[t][ds] = sa_gload(qb + g.hd + (int64_t)kk * ld + ds * 32 + g_ * 8, kv[t]);
        vf[t][ds] = sa_gload(qb + 2 * g.hd + (int64_t)kk * ld + ds * 32 + g_ * 8, kv[t]);
      }
    }
    f32x4 dkacc[TP][4], dvacc[TP][4];
#pragma unroll
    for (int t = 0; t < TP; ++t)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) { dkacc[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dvacc[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll 1
    for (int q2 = 0; q2 < KSN; ++q2) {
      f32x4 pp[TP][2], dss[TP][2];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int qt = 2 * q2 + half;
        f32x4 s[TP], dp[TP];
#pragma unroll
        for (int t = 0; t < TP; ++t) { s[t] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ds = 0; ds < 2; ++ds) {
          Vec8<bf16> qf = sa_frag_row(im0, qt * 16 + c_, ds, g_);
          Vec8<bf16> dof = sa_frag_row(im1, qt * 16 + c_, ds, g_);
#pragma unroll
          for (int t = 0; t < TP; ++t) { mma16(qf, kf[t][ds], s[t]); mma16(dof, vf[t][ds], dp[t]); }
        }
        const int q0 = qt * 16 + g_ * 4;                     // rows of these accumulators: q0 + r
        const f32x4 ls = *reinterpret_cast<const f32x4*>(lse_s + q0);
        const f32x4 dd = *reinterpret_cast<const f32x4*>(dq_s + q0);
#pragma unroll
        for (int t = 0; t < TP; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            // padded query rows: ls = +inf -> p = 0; padded key columns (zero K / V fragments) only feed their own,
            // never stored, dK / dV columns
            const float p = __builtin_amdgcn_exp2f(fmaf(s[t][r], sl, -ls[r]));
            pp[t][half][r] = p;
            dss[t][half][r] = p * (dp[t][r] - dd[r]);
          }
      }
      Vec8<bf16> pf[TP], sf[TP];
#pragma unroll
      for (int t = 0; t < TP; ++t) { pf[t] = sa_frag_acc(pp[t][0], pp[t][1]); sf[t] = sa_frag_acc(dss[t][0], dss[t][1]); }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        Vec8<bf16> dotf = sa_frag_trp(im1, q2 * 32, dt >> 1, dt & 1, lane);
        Vec8<bf16> qtf = sa_frag_trp(im0, q2 * 32, dt >> 1, dt & 1, lane);
#pragma unroll
        for (int t = 0; t < TP; ++t) {         // d{k,v}acc[t][2 dp + dtl][r] = d{K,V}[key = .. + c][32 dp + 8 g + 4 dtl + r]
          mma16(dotf, pf[t], dvacc[t][dt]);
          mma16(qtf, sf[t], dkacc[t][dt]);
        }
      }
    }
#pragma unroll
    for (int t = 0; t < TP; ++t) {
      const int key = (kp * TP + t) * 16 + c_;
      if (kv[t]) {
        bf16* p = dqb + (int64_t)key * ld + g_ * 8;
        store8<bf16>(p + g.hd, sa_out8(dkacc[t][0], dkacc[t][1], g.scale));
        store8<bf16>(p + g.hd + 32, sa_out8(dkacc[t][2], dkacc[t][3], g.scale));
        store8<bf16>(p + 2 * g.hd, sa_out8(dvacc[t][0], dvacc[t][1], 1.f));
        store8<bf16>(p + 2 * g.hd + 32, sa_out8(dvacc[t][2], dvacc[t][3], 1.f));
      }
    }
  }
}

bool sattn_ok(int dtype, int L, int D, int swin, const void* bias) {
  return vtx_opt(VTX_OPT_SATTN) && dtype == VTX_BF16 && D == 64 && !swin && bias == nullptr && L >= 1 && L <= 224;
}

// option SATTN_WAVES: 4 = four waves on PAIRS of 16-token tiles (round 1) | 8 = eight waves on single tiles: half the tiles per
// wave (13 tiles at L = 197: 2 + 2 + ... vs 4 + 4 + 4 + 2) and twice the waves per SIMD to hide the LDS / exp latencies.
// (Eight waves on pairs -- half the LDS fragment traffic per MFMA, but 170 registers = one workgroup per CU: backward
//  112 -> 132 us.  Phase ablation, SA_ABLATE: staging + loads 23 us, phase A 44-54, phase B 36-46 of the 113-us backward.)
// Round 6: waves per workgroup chosen so that the LIVE 16-token tiles divide among them.  Every wave walks tiles wave, wave + NW, ...: at
// L = 197 (13 live tiles) eight waves take 2 + 2 + 2 + 2 + 2 + 1 + 1 + 1 -- both phases last two tile periods with three waves idle in the
// second (13 of 16 slots) -- seven waves take 2 x 6 + 1 (13 of 14).  Candidates 6, 7, 8 (two workgroups per CU either way); the
// fewest idle slots wins, ties go to more waves.  A tile's arithmetic does not depend on the wave that runs it: bit-identical.
// SATTN_WAVES: 8 (default) | 7, 6 = forced | 1 = this rule | 4 = four waves on pairs of tiles (round 1).
// MEASURED (profiles/round6_sattn_wave_counts.txt): no gain -- ViT-S/16 B = 256 forward 44.6 (8 waves) / 45.8 (7) / 48.5 us (6), backward
// 119.3 / 121.5 / 127.9 us, ViT-S/16 step 12.02 -> 12.11 ms with the rule: the idle waves of the second tile period cost nothing (the
// other workgroup of the CU fills the SIMDs), fewer waves per SIMD hide less of the LDS -> MFMA -> exp chain.  The default stays 8.
static int sa_pick_nw(int L, int nkt) {
  const int o = vtx_opt(VTX_OPT_SATTN_WAVES);
  if (o == 4 || o == 6 || o == 7 || o == 8) return (nkt != 14 && (o == 6 || o == 7)) ? 8 : o;
  if (nkt != 14) return 8;
  const int nt = (L + 15) >> 4;
  int best = 8, waste = ((nt + 7) / 8) * 8 - nt;
  for (int w = 7; w >= 6; --w) {
    const int ws = ((nt + w - 1) / w) * w - nt;
    if (ws < waste) { waste = ws; best = w; }
  }
  return best;
}
extern "C" int vtx_sattn_waves(int L) { return sa_pick_nw(L, L <= 64 ? 4 : (L <= 128 ? 8 : 14)); }

template <int NKT, int TP, int NW> static int sattn_fwd_k(const void* qkv, void* o, float* lse, int B, const SeqGeom& g, hipStream_t st) {
  constexpr size_t smem = (size_t)2 * NKT * 16 * SA_ROWB;
  hipLaunchKernelGGL((sattn_fwd_kernel<NKT, TP, NW>), dim3(B * g.nH), dim3(64 * NW), smem, st, (const bf16*)qkv, (bf16*)o, lse, g);
  return vtx_check_launch();
}
template <int NKT> static int sattn_fwd_t(const void* qkv, void* o, float* lse, int B, const SeqGeom& g, hipStream_t st) {
  switch (sa_pick_nw(g.L, NKT)) {
    case 4: return sattn_fwd_k<NKT, 2, 4>(qkv, o, lse, B, g, st);
    case 6: if constexpr (NKT == 14) return sattn_fwd_k<NKT, 1, 6>(qkv, o, lse, B, g, st);
    case 7: if constexpr (NKT == 14) return sattn_fwd_k<NKT, 1, 7>(qkv, o, lse, B, g, st);
    default: return sattn_fwd_k<NKT, 1, 8>(qkv, o, lse, B, g, st);
  }
}
template <int NKT, int TP, int NW> static int sattn_bwd_k(const void* qkv, const void* o, const void* dout, const float* lse, void* dqkv,
                                                          int B, const SeqGeom& g, hipStream_t st) {
  constexpr size_t smem = (size_t)2 * NKT * 16 * SA_ROWB + 2 * NKT * 16 * sizeof(float);
  hipLaunchKernelGGL((sattn_bwd_kernel<NKT, TP, NW>), dim3(B * g.nH), dim3(64 * NW), smem, st, (const bf16*)qkv, (const bf16*)o,
                     (const bf16*)dout, lse, (bf16*)dqkv, g);
  return vtx_check_launch();
}
template <int NKT> static int sattn_bwd_t(const void* qkv, const void* o, const void* dout, const float* lse, void* dqkv,
                                          int B, const SeqGeom& g, hipStream_t st) {
  switch (sa_pick_nw(g.L, NKT)) {
    case 4: return sattn_bwd_k<NKT, 2, 4>(qkv, o, dout, lse, dqkv, B, g, st);
    case 6: if constexpr (NKT == 14) return sattn_bwd_k<NKT, 1, 6>(qkv, o, dout, lse, dqkv, B, g, st);
    case 7: if constexpr (NKT == 14) return sattn_bwd_k<NKT, 1, 7>(qkv, o, dout, lse, dqkv, B, g, st);
    default: return sattn_bwd_k<NKT, 1, 8>(qkv, o, dout, lse, dqkv, B, g, st);
  }
}

// key-tile count of the instantiation: 14 (L <= 224: ViT-S/16 at 224^2, L = 197), 8 (L <= 128), 4 (L <= 64: the 96^2 DINO
// crops, L = 37)
int sattn_fwd_launch(const void* qkv, void* o, float* lse, int B, int L, int nH, hipStream_t st, const int* perm) {
  SeqGeom g; g.L = L; g.nH = nH; g.hd = nH * SA_D; g.scale = 1.0f / sqrtf((float)SA_D); g.perm = perm;
  if (L <= 64) return sattn_fwd_t<4>(qkv, o, lse, B, g, st);
  if (L <= 128) return sattn_fwd_t<8>(qkv, o, lse, B, g, st);
  return sattn_fwd_t<14>(qkv, o, lse, B, g, st);
}

int sattn_bwd_launch(const void* qkv, const void* o, const void* dout, const float* lse, void* dqkv, int B, int L, int nH,
                     hipStream_t st, const int* perm) {
  SeqGeom g; g.L = L; g.nH = nH; g.hd = nH * SA_D; g.scale = 1.0f / sqrtf((float)SA_D); g.perm = perm;
  if (L <= 64) return sattn_bwd_t<4>(qkv, o, dout, lse, dqkv, B, g, st);
  if (L <= 128) return sattn_bwd_t<8>(qkv, o, dout, lse, dqkv, B, g, st);
  return sattn_bwd_t<14>(qkv, o, dout, lse, dqkv, B, g, st);
}
