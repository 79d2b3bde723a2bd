// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the ViT / Swin hot path.
//
// Every kernel is templated on ONE element type T in {float, bf16}:
//   * bf16  = the benchmark / training mode (bf16 storage, fp32 accumulation, fp32 statistics)
//   * float = the parity mode (exact-fp32 MFMA, used to check against the CPU oracle at 1e-3 rel)
// gfx950 only: 64-wide wavefronts, v_mfma_f32_16x16x32_bf16 / v_mfma_f32_16x16x4_f32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define VTX_WAVE 64

enum { VTX_OK = 0, VTX_ERR_SHAPE = -1, VTX_ERR_DTYPE = -2, VTX_ERR_ALIGN = -3, VTX_ERR_LAUNCH = -4,
       VTX_ERR_WORKSPACE = -5, VTX_ERR_NULL = -6 };
enum { VTX_F32 = 0, VTX_BF16 = 1 };

#ifndef VTX_PK_DS_GUARD
#define VTX_PK_DS_GUARD 1          // see shfl_xor_f below
#endif
// A value that goes to the vector-memory pipe as store data right after packed-fp32 math (see the HAZARD note at shfl_xor_f:
// no failure has been observed for stores -- they read their data registers later than a DS instruction does -- but the
// guard costs one s_nop)
template <typename V> __device__ __forceinline__ void vmem_guard(V& v) {
#if VTX_PK_DS_GUARD
  asm volatile("s_nop 0" : "+v"(v));
#endif
}

// ---------------------------------------------------------------- 8-element vectors of T
template <typename T> struct Vec8;
template <> struct Vec8<float> {
  float v[8];
  __device__ __forceinline__ float get(int i) const { return v[i]; }
  __device__ __forceinline__ void set(int i, float x) { v[i] = x; }
};
template <> struct alignas(16) Vec8<bf16> {
  bf16x8 v;
  __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
  __device__ __forceinline__ void set(int i, float x) { v[i] = (bf16)x; }
};

template <typename T> __device__ __forceinline__ Vec8<T> vec8_zero() {
  Vec8<T> r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.set(i, 0.f);
  return r;
}
// 8 consecutive T from global / LDS memory (16 B for bf16, 32 B for float); p must be 16-B aligned
template <typename T> __device__ __forceinline__ Vec8<T> load8(const T* p);
// load8 from an address that is ALWAYS readable, zeros when !valid -- WITHOUT a branch around the load (the load is issued by every
// lane, the zeros are a select).  Round 5 (tools/r5/win12_debug*.py, profiles/round5_mfma_branch_hazard.md): with `valid ?
// load8(p) : zero` hipcc (ROCm 7.2) puts the load of the NEXT operand tile under an exec-mask branch that sits between a chain of
// MFMAs and the v_accvgpr_read of their result; when every lane skips the load (a fully padded 16-token tile) the read follows the
// last MFMA by three instructions and returns the accumulator BEFORE the chain has drained -- the wait states the hazard
// recogniser inserts in straight-line code are missing on that path (fp32 attention with 10 key tiles: scores off by 5 %;
// -O1, or reading the accumulator in the MFMA's own block, is correct).  Callers clamp the address (row 0 of the operand).
template <typename T> __device__ __forceinline__ Vec8<T> load8_clamped(const T* p, bool valid) {
  Vec8<T> v = load8<T>(p);
  if (!valid) v = vec8_zero<T>();
  return v;
}
template <> __device__ __forceinline__ Vec8<bf16> load8<bf16>(const bf16* p) {
  Vec8<bf16> r; r.v = *reinterpret_cast<const bf16x8*>(p); return r;
}
template <> __device__ __forceinline__ Vec8<float> load8<float>(const float* p) {
  Vec8<float> r;
  f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) { r.v[i] = a[i]; r.v[4 + i] = b[i]; }
  return r;
}
template <typename T> __device__ __forceinline__ void store8(T* p, const Vec8<T>& x);
// VTX_NT_STORE8 (per translation unit; tools/r6/build_variant.sh A/B, profiles/round6_nt_store_screen.txt): every 16-byte bf16 output store of
// the file non-temporal -- outputs that stream past the L2 instead of displacing the operands it holds (what ASTAT_NT does for gemm_astat.hip)
#ifndef VTX_NT_STORE8
#define VTX_NT_STORE8 0
#endif
template <> __device__ __forceinline__ void store8<bf16>(bf16* p, const Vec8<bf16>& x) {
  if constexpr (VTX_NT_STORE8 != 0) __builtin_nontemporal_store(x.v, reinterpret_cast<bf16x8*>(p));
  else *reinterpret_cast<bf16x8*>(p) = x.v;
}
template <> __device__ __forceinline__ void store8<float>(float* p, const Vec8<float>& x) {
  f32x4 a, b;
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = x.v[i]; b[i] = x.v[4 + i]; }
  vmem_guard(a); vmem_guard(b);
  *reinterpret_cast<f32x4*>(p) = a; *reinterpret_cast<f32x4*>(p + 4) = b;
}

template <typename T> __device__ __forceinline__ float to_f32(T x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x) { return (T)x; }
// value as the product path would store it (round to T, back to fp32)
template <typename T> __device__ __forceinline__ float round_to(float x) { return (float)((T)x); }

// ---------------------------------------------------------------- MFMA 16x16 "k-slot" contraction
// acc[r] (row 4*(lane>>4)+r, col lane&15) += sum over g in 0..3, j in 0..7 of
//        A(lane with lane&15 == row, lane>>4 == g)[j] * B(lane with lane&15 == col, lane>>4 == g)[j]
// i.e. each lane supplies 8 "k-slots" (g,j) of one A row / one B column; any assignment of real k
// indices to slots is valid as long as A and B use the same one.
//   bf16 : one v_mfma_f32_16x16x32_bf16
//   float: eight v_mfma_f32_16x16x4_f32 (k index = lane>>4 per instruction), exact fp32
__device__ __forceinline__ void mma16(const Vec8<bf16>& a, const Vec8<bf16>& b, f32x4& c) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c, 0, 0, 0);
}
__device__ __forceinline__ void mma16(const Vec8<float>& a, const Vec8<float>& b, f32x4& c) {
#pragma unroll
  for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[j], b.v[j], c, 0, 0, 0);
}

// Read an MFMA accumulator in the block of its MFMAs (the compiler pads THAT read with the wait states the matrix pipe needs; a
// read it places behind a branch is not padded on the taken path: profiles/round5_mfma_branch_hazard.md).  Used where branches
// follow the products closely and speed does not matter: the fp32 parity mode and the dropout variants of the generic kernels.
__device__ __forceinline__ void acc_settle(f32x4& c) { asm volatile("" : "+v"(c)); }

// ---------------------------------------------------------------- wave reductions (64 lanes)
// HAZARD (found in round 4, profiles/round4_nondeterminism_root_cause.md): on gfx950 a packed-fp32 VOP3P instruction
// (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 -- two passes over the register pair) followed IN THE NEXT ISSUE SLOT by an LDS
// instruction that reads its result as data (ds_bpermute_b32 = __shfl_xor) now and then hands the LDS pipe the OLD register
// contents.  hipcc (ROCm 7.2) pads a dependent VALU consumer of a VOP3P result with `s_nop 0`, not a DS consumer, and its SLP
// vectoriser turns two interleaved scalar reduction chains (LayerNorm backward: the sums s1, s2) into exactly this pair of
// instructions.  Seen only while another kernel shares the CU (the side-stream weight gradient): one row of a LayerNorm
// backward's dx off by a few bf16 ulps in ~10 % of full-size train steps.  ONE wait state between the two removes it (0 of 900
// trials); the guard below supplies it for every cross-lane exchange of the library (VTX_PK_DS_GUARD=0: the unguarded code,
// for tools/probe/merge_bisect.py).  tools/probe/scan_pk_hazard.py scans the generated ISA for remaining sites.
__device__ __forceinline__ float shfl_xor_f(float v, int m) {
#if VTX_PK_DS_GUARD
  asm volatile("s_nop 0" : "+v"(v));
#endif
  return __shfl_xor(v, m, 64);
}
__device__ __forceinline__ float shfl_f(float v, int src) {          // value of lane `src`
#if VTX_PK_DS_GUARD
  asm volatile("s_nop 0" : "+v"(v));
#endif
  return __shfl(v, src, 64);
}
// DPP lane exchange inside a 16-lane row (no LDS round trip, unlike __shfl_xor which lowers to ds_bpermute_b32)
template <int CTRL> __device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// Sum over groups of G = 16, 32 or 64 consecutive lanes, result in every lane of the group: the four steps inside a
// 16-lane row are DPP (quad_perm xor 1, xor 2, row_half_mirror, row_mirror); only the steps across rows go through LDS.
template <int G> __device__ __forceinline__ float group_sum(float v) {
  v += dpp_f<0xB1>(v);       // quad_perm [1,0,3,2]
  v += dpp_f<0x4E>(v);       // quad_perm [2,3,0,1]
  v += dpp_f<0x141>(v);      // row_half_mirror: 8-lane sums
  v += dpp_f<0x140>(v);      // row_mirror: 16-lane sums
  if constexpr (G >= 32) v += shfl_xor_f(v, 16);
  if constexpr (G >= 64) v += shfl_xor_f(v, 32);
  return v;
}

// LayerNorm backward, element level -- shared by ln_bwd_kernel (csrc/layernorm.hip) and the fold of that kernel into the fused-MLP
// backward (csrc/mlp_fused.hip, round 6): the SAME expressions (hence the same fp contraction by hipcc) in both, so that the folded
// path reproduces the stand-alone launch's dx bit for bit as long as the row sums are formed in group_sum<16>'s order.
//   accum: xh = (x - mu) rs;  gv = d gamma;  s1 += gv;  s2 += gv xh      (d = the incoming gradient dln, x = the LayerNorm's input)
//   out:   dx = dres + rs (gv - c1 - xh c2)   with c1 = mean(gv), c2 = mean(gv xh) over the row
__device__ __forceinline__ void ln_bwd_elem_accum(float d, float x, float mu, float rs, float gm, float& s1, float& s2, float& xh, float& gv) {
  xh = (x - mu) * rs;
  gv = d * gm;
  s1 += gv;
  s2 += gv * xh;
}
__device__ __forceinline__ float ln_bwd_elem_out(float dres, float rs, float gv, float c1, float xh, float c2) {
  return dres + rs * (gv - c1 - xh * c2);
}
__device__ __forceinline__ float ln_bwd_elem_out(float rs, float gv, float c1, float xh, float c2) {
  return rs * (gv - c1 - xh * c2);
}

// LayerNorm forward, element level -- shared by ln_fwd_kernel and its fold into the row-streaming kernels that consume the normalised
// rows (ln_fold.h LnFwdFold): same expressions, same contraction.
__device__ __forceinline__ void ln_fwd_elem_sq(float x, float mu, float& q) { const float d = x - mu; q += d * d; }
__device__ __forceinline__ float ln_fwd_elem_out(float x, float mu, float rs, float gm, float bt) { return (x - mu) * rs * gm + bt; }

// v_rcp_f32 (1 ulp) instead of the ~10-instruction IEEE division: these run once per output element in GEMM epilogues
__device__ __forceinline__ float sigmoidf_(float z) { return __builtin_amdgcn_rcpf(1.f + __expf(-z)); }
__device__ __forceinline__ float silu_f(float z) { return z * sigmoidf_(z); }
__device__ __forceinline__ float dsilu_f(float z) { float s = sigmoidf_(z); return s * (1.f + z * (1.f - s)); }
// exact (erf) GELU of nn.GELU() and its derivative 0.5 (1 + erf(z / sqrt 2)) + z exp(-z^2 / 2) / sqrt(2 pi).
// erf by Abramowitz-Stegun 7.1.26 (absolute error <= 1.5e-7, below bf16 AND fp32-parity resolution of the products it
// feeds): branch-free, one v_exp_f32 and one v_rcp_f32 -- libm's erff is ~3x the instructions with a divergent branch,
// and these run once per MLP hidden element in the GEMM epilogues (50432 x 384 -> 1536 MLP: fc1 forward 213 -> 186 us, fc2 dgrad 197 -> 166 us).
// With u = z / sqrt 2 the same exponential exp(-u^2) = exp(-z^2 / 2) serves erf and the Gaussian density.
__device__ __forceinline__ float erf_as(float au, float e) {          // au = |u|, e = exp(-u^2)  ->  erf(|u|)
  const float t = __builtin_amdgcn_rcpf(1.f + 0.3275911f * au);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  return 1.f - poly * e;
}
__device__ __forceinline__ float gelu_f(float z) {
  const float au = fabsf(z) * 0.70710678118654752f;
  const float er = copysignf(erf_as(au, __expf(-au * au)), z);
  return 0.5f * z * (1.f + er);
}
__device__ __forceinline__ float dgelu_f(float z) {
  const float au = fabsf(z) * 0.70710678118654752f;
  const float e = __expf(-au * au);
  const float er = copysignf(erf_as(au, e), z);
  return 0.5f * (1.f + er) + z * 0.39894228040143268f * e;
}

// out[c] = sum_b part[b][c] for c < C (out0) and C <= c < 2C (out1, optional).  Launch with 1024 threads and
// ceil(ncols / 32) blocks: 32 columns x 32 row lanes per block, fixed summation order (deterministic).
static __global__ __launch_bounds__(1024) void colreduce_kernel(const float* __restrict__ part, float* __restrict__ out0,
                                                                float* __restrict__ out1, int nb, int C, int ld) {
  __shared__ float red[32][33];
  const int ncols = out1 ? 2 * C : C;
  const int nrl = blockDim.x >> 5;                       // row lanes: 32 with the 1024-thread launch
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  float s = 0.f;
  if (c < ncols) {
    // 4 independent accumulators in a fixed interleave: the loads of 4 rows are in flight together
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = rl;
    for (; b + 3 * nrl < nb; b += 4 * nrl) {
      s0 += part[(int64_t)b * ld + c];
      s1 += part[(int64_t)(b + nrl) * ld + c];
      s2 += part[(int64_t)(b + 2 * nrl) * ld + c];
      s3 += part[(int64_t)(b + 3 * nrl) * ld + c];
    }
    for (; b < nb; b += nrl) s0 += part[(int64_t)b * ld + c];
    s = (s0 + s1) + (s2 + s3);
  }
  red[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && c < ncols) {
    float t = 0.f;
    for (int r = 0; r < nrl; ++r) t += red[r][cl];
    if (c < C) out0[c] = t; else out1[c - C] = t;
  }
}
static inline dim3 colreduce_grid(int ncols) { return dim3((ncols + 31) / 32); }

// Up to 4 such reductions in ONE launch (the LayerNorm dgamma / dbeta partials and the rel_pos gradient partials of a
// transformer layer's backward: three 5-us launches become one): segment s owns blocks [blk0[s], blk0[s + 1]).
struct ColReduceMulti {
  const float* part[4]; float* out0[4]; float* out1[4];
  int nb[4], C[4], ld[4], blk0[5];
  int n;
};
static __global__ __launch_bounds__(1024) void colreduce_multi_kernel(ColReduceMulti m) {
  __shared__ float red[32][33];
  int sgm = 0;
#pragma unroll
  for (int i = 1; i < 4; ++i)
    if (i < m.n && (int)blockIdx.x >= m.blk0[i]) sgm = i;
  const float* __restrict__ part = m.part[sgm];
  float* __restrict__ out0 = m.out0[sgm];
  float* __restrict__ out1 = m.out1[sgm];
  const int nb = m.nb[sgm], C = m.C[sgm], ld = m.ld[sgm];
  const int ncols = out1 ? 2 * C : C;
  const int nrl = blockDim.x >> 5;
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c = ((int)blockIdx.x - m.blk0[sgm]) * 32 + cl;
  float s = 0.f;
  if (c < ncols) {                                       // the same fixed interleave as colreduce_kernel: identical bits
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = rl;
    for (; b + 3 * nrl < nb; b += 4 * nrl) {
      s0 += part[(int64_t)b * ld + c];
      s1 += part[(int64_t)(b + nrl) * ld + c];
      s2 += part[(int64_t)(b + 2 * nrl) * ld + c];
      s3 += part[(int64_t)(b + 3 * nrl) * ld + c];
    }
    for (; b < nb; b += nrl) s0 += part[(int64_t)b * ld + c];
    s = (s0 + s1) + (s2 + s3);
  }
  red[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && c < ncols) {
    float t = 0.f;
    for (int r = 0; r < nrl; ++r) t += red[r][cl];
    if (c < C) out0[c] = t; else out1[c - C] = t;
  }
}

// out[i] = sum_z slab[z][i]   (fixed order: deterministic).  n must be a multiple of 4.
static __global__ void slab_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ out, int64_t n, int nz) {
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    const f32x4* p = reinterpret_cast<const f32x4*>(slabs) + i;
    int z = 0;
    for (; z + 4 <= nz; z += 4) {
      f32x4 a = p[(int64_t)z * n4], b = p[(int64_t)(z + 1) * n4], c2 = p[(int64_t)(z + 2) * n4], d = p[(int64_t)(z + 3) * n4];
      s += a; s += b; s += c2; s += d;
    }
    for (; z < nz; ++z) s += p[(int64_t)z * n4];
    vmem_guard(s);
    reinterpret_cast<f32x4*>(out)[i] = s;
  }
}
// The same for up to 16 (slab set, output) pairs in ONE launch -- the weight and bias gradients of a grouped weight-gradient
// launch: seg[s] covers blocks blk0[s] .. blk0[s + 1] - 1, each block 256 threads x one float4.
struct SlabReduceMulti {
  const float* slabs[16];
  float* out[16];
  long long n[16];          // elements per slab (multiples of 4)
  int blk0[17];
  int nseg, nz;
};
static __global__ __launch_bounds__(256) void slab_reduce_multi_kernel(SlabReduceMulti a) {
  // (static indices only: a dynamically indexed by-value kernel argument is copied to scratch)
  const float* slabs = a.slabs[0];
  float* out = a.out[0];
  long long n = a.n[0];
  int b0 = 0;
#pragma unroll
  for (int i = 1; i < 16; ++i)
    if (i < a.nseg && (int)blockIdx.x >= a.blk0[i]) { slabs = a.slabs[i]; out = a.out[i]; n = a.n[i]; b0 = a.blk0[i]; }
  const int64_t n4 = n >> 2;
  const int64_t i = (int64_t)((int)blockIdx.x - b0) * 256 + threadIdx.x;
  if (i >= n4) return;
  const f32x4* p = reinterpret_cast<const f32x4*>(slabs) + i;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  const int nz = a.nz;
  int z = 0;
  for (; z + 8 <= nz; z += 8) {
    f32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = p[(int64_t)(z + j) * n4];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j];
  }
  for (; z + 4 <= nz; z += 4) {
    f32x4 b0 = p[(int64_t)z * n4], b1 = p[(int64_t)(z + 1) * n4], b2 = p[(int64_t)(z + 2) * n4], b3 = p[(int64_t)(z + 3) * n4];
    s += b0; s += b1; s += b2; s += b3;
  }
  for (; z < nz; ++z) s += p[(int64_t)z * n4];
  vmem_guard(s);
  reinterpret_cast<f32x4*>(out)[i] = s;
}

// ONE launch for every small reduction a transformer layer's backward leaves behind (round 3): the split-K slabs of the
// grouped weight gradient (up to 16 (slab set, output) pairs, as slab_reduce_multi_kernel) AND up to 4 column reductions
// (LayerNorm dgamma / dbeta partials x 2, rel_pos gradient partials, as colreduce_multi_kernel).  1024-thread blocks;
// segment s owns blocks [blk0[s], blk0[s + 1]): slab segments first (1024 float4 per block), then the column segments
// (32 columns x 32 row lanes per block).  Per output element the summation orders are those of the two kernels it
// replaces: identical bits.
struct LayerReduce {
  const float* slabs[16]; float* out[16]; long long n[16];
  const float* part[4]; float* out0[4]; float* out1[4];
  int nb[4], C[4], ld[4];
  int blk0[17];             // slab segments: blocks [blk0[s], blk0[s + 1])
  int cblk0[5];             // column segments, behind all slab blocks (cblk0[0] = number of slab blocks)
  int nseg, nz, ncol;
  int accumulate;           // 1: every output is out + sum (a parameter's SECOND gradient of one backward lands on the first)
};
static __global__ __launch_bounds__(1024) void layer_reduce_kernel(LayerReduce a) {
  __shared__ float red[32][33];
  const int bid = (int)blockIdx.x;
  if (bid < a.cblk0[0]) {                                  // ---- a split-K slab segment (block-uniform branch; static indices only)
    const float* slabs = a.slabs[0];
    float* out = a.out[0];
    long long n = a.n[0];
    int b0 = 0;
#pragma unroll
    for (int i = 1; i < 16; ++i)
      if (i < a.nseg && bid >= a.blk0[i]) { slabs = a.slabs[i]; out = a.out[i]; n = a.n[i]; b0 = a.blk0[i]; }
    const int64_t n4 = n >> 2;
    const int64_t i = (int64_t)(bid - b0) * 1024 + threadIdx.x;
    if (i >= n4) return;
    const f32x4* p = reinterpret_cast<const f32x4*>(slabs) + i;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    const int nz = a.nz;
    int z = 0;
    for (; z + 8 <= nz; z += 8) {
      f32x4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = p[(int64_t)(z + j) * n4];
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[j];
    }
    for (; z + 4 <= nz; z += 4) {
      f32x4 c0 = p[(int64_t)z * n4], c1 = p[(int64_t)(z + 1) * n4], c2 = p[(int64_t)(z + 2) * n4], c3 = p[(int64_t)(z + 3) * n4];
      s += c0; s += c1; s += c2; s += c3;
    }
    for (; z < nz; ++z) s += p[(int64_t)z * n4];
    if (a.accumulate) s = reinterpret_cast<const f32x4*>(out)[i] + s;      // one add of two finished sums = autograd's a + b
    vmem_guard(s);
    reinterpret_cast<f32x4*>(out)[i] = s;
    return;
  }
  // ---- a column-reduce segment (the code of colreduce_multi_kernel)
  int sgm = 0;
#pragma unroll
  for (int i = 1; i < 4; ++i)
    if (i < a.ncol && bid >= a.cblk0[i]) sgm = i;
  const float* __restrict__ part = a.part[0];
  float* __restrict__ out0 = a.out0[0];
  float* __restrict__ out1 = a.out1[0];
  int nb = a.nb[0], C = a.C[0], ld = a.ld[0], cb0 = a.cblk0[0];
#pragma unroll
  for (int i = 1; i < 4; ++i)
    if (i == sgm) { part = a.part[i]; out0 = a.out0[i]; out1 = a.out1[i]; nb = a.nb[i]; C = a.C[i]; ld = a.ld[i]; cb0 = a.cblk0[i]; }
  const int ncols = out1 ? 2 * C : C;
  const int nrl = 32;
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c = (bid - cb0) * 32 + cl;
  float s = 0.f;
  if (c < ncols) {                                       // the same fixed interleave as colreduce_kernel: identical bits
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = rl;
    for (; b + 3 * nrl < nb; b += 4 * nrl) {
      s0 += part[(int64_t)b * ld + c];
      s1 += part[(int64_t)(b + nrl) * ld + c];
      s2 += part[(int64_t)(b + 2 * nrl) * ld + c];
      s3 += part[(int64_t)(b + 3 * nrl) * ld + c];
    }
    for (; b < nb; b += nrl) s0 += part[(int64_t)b * ld + c];
    s = (s0 + s1) + (s2 + s3);
  }
  red[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && c < ncols) {
    float t = 0.f;
    for (int r = 0; r < nrl; ++r) t += red[r][cl];
    float* dst = c < C ? out0 + c : out1 + (c - C);
    *dst = a.accumulate ? *dst + t : t;
  }
}

// ---------------------------------------------------------------- attention-probability dropout
// Reference: F.dropout(attn, p) on the softmax output (models/vit.py:39, swin_transformer.py:144, pvt.py:60, twins.py:88,147).
// The keep decision of cell (problem, query, key) is a counter-based hash of (seed, problem, query * Lk + key): the backward
// regenerates it, no mask tensor exists in HBM.  keep != NULL: an explicit keep mask [problems][Lq][Lk] (bytes, 1 = keep)
// replaces the hash -- the parity tests feed the mask the reference drew.  scale = 1 / (1 - p).
struct DropArgs {
  float scale;
  unsigned thresh, s0, s1;
  const uint8_t* keep;
  int Lq, Lk;
};
__device__ __forceinline__ unsigned drop_hash(unsigned s0, unsigned s1, unsigned prob, unsigned cell) {
  unsigned h = s0 ^ (prob * 0x9E3779B1u);
  h ^= h >> 15; h *= 0x85EBCA6Bu; h ^= h >> 13;
  h += cell * 0xC2B2AE35u + s1;
  h ^= h >> 16; h *= 0x27D4EB2Fu; h ^= h >> 15; h *= 0x165667B1u; h ^= h >> 16;
  return h;
}
// factor of attention probability (problem, q, key): 0 (dropped) or 1 / (1 - p); cells outside [Lq) x [Lk) are never used
__device__ __forceinline__ float drop_factor(const DropArgs& d, unsigned prob, int q, int key) {
  if (d.keep != nullptr) {
    if (q >= d.Lq || key >= d.Lk) return 0.f;
    return d.keep[((int64_t)prob * d.Lq + q) * d.Lk + key] ? d.scale : 0.f;
  }
  return drop_hash(d.s0, d.s1, prob, (unsigned)(q * d.Lk + key)) >= d.thresh ? d.scale : 0.f;
}
static inline int drop_args(DropArgs& d, float p, uint64_t seed, const uint8_t* keep, int Lq, int Lk) {
  if (!(p > 0.f) || !(p < 1.f)) return VTX_ERR_SHAPE;
  d.scale = 1.f / (1.f - p);
  const double t = (double)p * 4294967296.0;
  d.thresh = t >= 4294967295.0 ? 4294967295u : (unsigned)t;
  d.s0 = (unsigned)seed; d.s1 = (unsigned)(seed >> 32);
  d.keep = keep; d.Lq = Lq; d.Lk = Lk;
  return VTX_OK;
}

static inline dim3 slab_reduce_grid(int64_t n) {
  int64_t nb = ((n >> 2) + 255) / 256;
  return dim3((unsigned)(nb > 4096 ? 4096 : (nb < 1 ? 1 : nb)));
}

static inline int vtx_check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? VTX_OK : VTX_ERR_LAUNCH;
}
