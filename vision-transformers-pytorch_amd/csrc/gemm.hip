// MFMA GEMM family for gfx950 with fused prologues / epilogues -- every dense contraction of the hot
// path except the attention cores (QKV / proj / MLP / patch-embed / PatchMerge / classifier linears of
// reference models/vit.py:23-25,31,43, models/swin_transformer.py:34-35,128,155,205,222,
// models/layer.py:191-196) and their dgrad / wgrad.
//
//   C[M,N] = epilogue( sum_k opA(A)[m,k] * opB(B)[n,k] )
//     TA = false: A[m*lda + k]   (k contiguous)      TA = true: A[k*lda + m]   (m contiguous)
//     TB = false: B[n*ldb + k]                       TB = true: B[k*ldb + n]
//   forward  y = x W^T      : TA=0 TB=0  A=x[M,K]   B=W[N,K]
//   dgrad    dx = dy W      : TA=0 TB=1  A=dy[M,N'] B=W[N',K'] (contraction over W's rows)
//   wgrad    dW = dy^T x    : TA=1 TB=1  A=dy[M',N] B=x[M',K'] (contraction over tokens, split-K slabs)
//
// Block = 256 threads = 4 waves (2x2); block tile 128 x BN (BN = 128 / 96 / 64), LDS k-tile of 128 bytes
// per row (64 bf16 / 32 fp32) at a 160-byte row stride (conflict-free ds_read_b128 fragment reads);
// global -> registers -> LDS staging with register prefetch of the next k-tile; transposed operands are
// transposed in registers (4x8 micro-tiles) on their way into LDS so both MFMA operands are always read
// as 8 contiguous k-slots per lane.  bias / SiLU (+ saved pre-activation) / silu' / DropPath scale / residual
// add are applied in the epilogue, which goes through LDS so that every global access is a full 16-byte
// vector per lane.  (SiLU as an operand PROLOGUE was measured 2.7x slower on fc2: the transcendental work
// is repeated per N-tile and serialises with the staging; writing h once from fc1's epilogue is cheaper.)
// Workgroup ids are remapped so that tiles sharing an operand panel run on the same XCD (private L2).
#include "gemm_common.h"
#include "options.h"

template <typename T> struct GemmGeom {
  static constexpr int BK = 128 / (int)sizeof(T);      // elements per LDS k-tile row
  static constexpr int STRIDE = 160 / (int)sizeof(T);  // LDS row stride in elements
  static constexpr int VPR = BK / 8;                   // 8-element vectors per row
  static constexpr int KS = BK / 32;                   // mma16 k-steps per LDS tile
  static constexpr int KG = BK / 4;                    // 4-row contraction groups (transposed staging)
};

// ---- staging of one operand tile (R tile rows x BK contraction) -------------------------------
template <typename T, int R, bool TR> struct Stage {
  using G = GemmGeom<T>;
  static constexpr int NITEMS = TR ? (G::KG * (R / 8)) : (R * G::VPR);
  static constexpr int NIT = (NITEMS + 255) / 256;
  static constexpr int NREG = TR ? 4 : 1;
  Vec8<T> reg[NIT][NREG];

  // global -> registers.  base: operand pointer; t0: first tile row (m or n); k0: first contraction index
  __device__ __forceinline__ void gload(const T* __restrict__ base, int64_t ld, int t0, int tdim, int k0, int kend,
                                        const float* __restrict__, int) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = threadIdx.x + it * 256;
      if (!TR) {
        const int r = idx / G::VPR, kv = idx % G::VPR;
        const int row = t0 + r, k = k0 + kv * 8;
        if (idx < NITEMS && row < tdim && k < kend) reg[it][0] = load8<T>(base + (int64_t)row * ld + k);
        else reg[it][0] = vec8_zero<T>();
      } else {
        const int kg = idx % G::KG, rc = idx / G::KG;
        const int col = t0 + rc * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int k = k0 + kg * 4 + i;
          if (idx < NITEMS && col < tdim && k < kend) reg[it][i] = load8<T>(base + (int64_t)k * ld + col);
          else reg[it][i] = vec8_zero<T>();
        }
      }
    }
  }

  // transposed operand only: per-sample scale along the contraction index (weight gradient through DropPath).
  // The scale values are requested with the tile and applied when the registers go to LDS (kscale_apply), so the
  // prefetch of the tile itself is not waited for early.
  float ksv[TR ? NIT : 1][4];
  __device__ __forceinline__ void kscale_fetch(int k0, int kend, const float* __restrict__ kscale, int k_per_scale) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = threadIdx.x + it * 256;
      const int kg = idx % G::KG;
      // the 4 contraction rows of this item are consecutive: one division, then compare
      const int kb = k0 + kg * 4;
      const int sq = kb / k_per_scale, sr = kb - sq * k_per_scale;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = kb + i;
        const int si = k_per_scale >= 4 ? sq + (sr + i >= k_per_scale ? 1 : 0) : k / k_per_scale;
        ksv[it][i] = (idx < NITEMS && k < kend) ? kscale[si] : 0.f;
      }
    }
  }
  // kmask: every scale is 0 or one constant -> rows are only masked here, the caller scales its accumulators
  __device__ __forceinline__ void kscale_apply(bool kmask) {
#pragma unroll
    for (int it = 0; it < NIT; ++it)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (kmask) {
          if (ksv[it][i] == 0.f) reg[it][i] = vec8_zero<T>();
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) reg[it][i].set(e, reg[it][i].get(e) * ksv[it][i]);
        }
      }
  }

  // registers -> LDS tile [R][STRIDE]
  __device__ __forceinline__ void lstore(T* __restrict__ lds) const {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = threadIdx.x + it * 256;
      if (idx >= NITEMS) continue;
      if (!TR) {
        const int r = idx / G::VPR, kv = idx % G::VPR;
        store8<T>(lds + r * G::STRIDE + kv * 8, reg[it][0]);
      } else {
        const int kg = idx % G::KG, rc = idx / G::KG;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          T* p = lds + (rc * 8 + e) * G::STRIDE + kg * 4;
          if constexpr (sizeof(T) == 2) {
            bf16x4 w;
#pragma unroll
            for (int i = 0; i < 4; ++i) w[i] = reg[it][i].v[e];
            *reinterpret_cast<bf16x4*>(p) = w;
          } else {
            f32x4 w;
#pragma unroll
            for (int i = 0; i < 4; ++i) w[i] = reg[it][i].v[e];
            *reinterpret_cast<f32x4*>(p) = w;
          }
        }
      }
    }
  }
};

template <typename T, typename TO, int BM, int BN, bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs p) {
  using G = GemmGeom<T>;
  constexpr int WM = BM / 32, WN = BN / 32;   // 16x16 MFMA tiles per wave along M / N
  constexpr int CSTR = BN + 4;                // fp32 C-staging row stride (floats)
  static_assert((BM + BN) * 160 >= (BM / 2) * CSTR * 4, "C staging must fit in the operand LDS");
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[(BM + BN) * 160];
  T* ldsA = reinterpret_cast<T*>(lds_raw);
  T* ldsB = ldsA + BM * G::STRIDE;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int c_ = lane & 15, g_ = lane >> 4;

  // ---- XCD-aware tile mapping: dispatch id d runs on XCD d % 8; give each XCD a contiguous range of
  //      (slice, m-tile, n-tile) ids so blocks that share an operand panel share an L2 (bijective remap).
  const int ntn = gridDim.x, ntm = gridDim.y;
  const int nblk = ntn * ntm * gridDim.z;
  const int did = (blockIdx.z * ntm + blockIdx.y) * ntn + blockIdx.x;
  const int xq = nblk >> 3, xr = nblk & 7, xcd = did & 7;
  const int lid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (did >> 3);
  const int tn = lid % ntn;
  const int tm = (lid / ntn) % ntm;
  const int tz = lid / (ntn * ntm);
  const int m0 = tm * BM, n0 = tn * BN;
  const int kbeg = tz * p.kchunk;
  const int kend = min(p.K, kbeg + p.kchunk);
  const int nk = (kend - kbeg + G::BK - 1) / G::BK;

  const T* A = (const T*)p.A;
  const T* B = (const T*)p.B;

  f32x4 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  Stage<T, BM, TA> sa;
  Stage<T, BN, TB> sb;
  const bool kmask = TA && p.kscale != nullptr && p.kscale_const > 0.f;
  sa.gload(A, p.lda, m0, p.M, kbeg, kend, nullptr, 1);
  if constexpr (TA) { if (p.kscale != nullptr) sa.kscale_fetch(kbeg, kend, p.kscale, p.k_per_scale); }
  sb.gload(B, p.ldb, n0, p.N, kbeg, kend, nullptr, 1);
  if constexpr (TA) { if (p.kscale != nullptr) sa.kscale_apply(kmask); }
  sa.lstore(ldsA);
  sb.lstore(ldsB);
  __syncthreads();

  // bias gradient rides on the wgrad kernel: row sums of the (already DropPath-scaled) dy^T tile in LDS,
  // done by the blocks of the first column tile only
  const bool do_ksum = TA && p.ksum_out != nullptr && tn == 0 && threadIdx.x < BM;
  float ksum = 0.f;

  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    if (more) {
      const int k0 = kbeg + (kt + 1) * G::BK;
      sa.gload(A, p.lda, m0, p.M, k0, kend, nullptr, 1);
      if constexpr (TA) { if (p.kscale != nullptr) sa.kscale_fetch(k0, kend, p.kscale, p.k_per_scale); }
      sb.gload(B, p.ldb, n0, p.N, k0, kend, nullptr, 1);
    }
    if (do_ksum) {
      const T* row = ldsA + threadIdx.x * G::STRIDE;
#pragma unroll
      for (int v = 0; v < G::VPR; ++v) {
        Vec8<T> t = load8<T>(row + v * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) ksum += t.get(e);
      }
    }
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) {
      Vec8<T> fa[WM], fb[WN];
#pragma unroll
      for (int i = 0; i < WM; ++i)
        fa[i] = load8<T>(ldsA + (wm * (BM / 2) + i * 16 + c_) * G::STRIDE + ks * 32 + g_ * 8);
#pragma unroll
      for (int j = 0; j < WN; ++j)
        fb[j] = load8<T>(ldsB + (wn * (BN / 2) + j * 16 + c_) * G::STRIDE + ks * 32 + g_ * 8);
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) mma16(fa[i], fb[j], acc[i][j]);
    }
    __syncthreads();
    if (more) {
      if constexpr (TA) { if (p.kscale != nullptr) sa.kscale_apply(kmask); }
      sa.lstore(ldsA);
      sb.lstore(ldsB);
      __syncthreads();
    }
  }

  if (kmask) {
    ksum *= p.kscale_const;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j) acc[i][j] *= p.kscale_const;
  }
  if (do_ksum && m0 + (int)threadIdx.x < p.M) p.ksum_out[(int64_t)tz * p.M + m0 + threadIdx.x] = ksum;

  EpiOperands<T, BM, BN> eo;
  eo.load(p, m0, n0, wn, c_);
  gemm_epilogue<T, TO, BM, BN>(p, acc, lds_raw, m0, n0, tz, wm, wn, c_, g_, eo);
}

template <typename T, typename TO, int BN, bool TA, bool TB>
static int gemm_launch(const GemmArgs& a, int nz, hipStream_t st) {
  constexpr int BM = 128;
  dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM, nz);
  hipLaunchKernelGGL((gemm_kernel<T, TO, BM, BN, TA, TB>), grid, dim3(256), 0, st, a);
  return vtx_check_launch();
}

template <typename T, typename TO, bool TA, bool TB>
static int gemm_pick_bn(const GemmArgs& a, int nz, hipStream_t st) {
  if (a.N % 128 == 0) return gemm_launch<T, TO, 128, TA, TB>(a, nz, st);
  if (a.N % 96 == 0) return gemm_launch<T, TO, 96, TA, TB>(a, nz, st);
  if (a.N <= 64) return gemm_launch<T, TO, 64, TA, TB>(a, nz, st);
  return gemm_launch<T, TO, 128, TA, TB>(a, nz, st);
}

static int gemm_validate(const GemmArgs& a, int mode) {
  if (!a.A || !a.B || !a.C) return VTX_ERR_NULL;
  if (a.M <= 0 || a.N <= 0 || a.K <= 0) return VTX_ERR_SHAPE;
  // 8-element vector granularity of the contiguous dimensions (loads) and of the output rows (stores)
  if ((a.N & 7) || (a.ldc & 7) || (a.lda & 7) || (a.ldb & 7)) return VTX_ERR_ALIGN;
  if ((mode == 0 || mode == 1) && (a.K & 7)) return VTX_ERR_ALIGN;   // A[m][k] rows
  if (mode == 2 && (a.M & 7)) return VTX_ERR_ALIGN;                   // A^T[k][m] rows
  if ((a.act == 2 || a.act == 4) && !a.aux_in) return VTX_ERR_NULL;
  return VTX_OK;
}

extern "C" {

// mode 0: C = epi(A W^T)    mode 1: C = epi(A W)     (see GemmArgs for the fused epilogue)
int vtx_gemm(int mode, int dtype, const void* A, const void* B, void* C, int M, int N, int K, int64_t lda,
             int64_t ldb, int64_t ldc, const float* bias, const void* resid, const float* rowscale,
             int rows_per_scale, void* aux_out, const void* aux_in, int act, void* stream) {
  GemmArgs a;
  a.A = A; a.B = B; a.C = C; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
  a.bias = bias; a.resid = resid; a.rowscale = rowscale; a.rows_per_scale = rows_per_scale > 0 ? rows_per_scale : 1;
  a.aux_out = aux_out; a.aux_in = aux_in; a.act = act; a.kscale = nullptr; a.k_per_scale = 1; a.kscale_const = 0.f;
  a.ksum_out = nullptr;
  a.kchunk = ((K + 127) / 128) * 128;
  gemm_args_nomap(a);
  if (mode != 0 && mode != 1) return VTX_ERR_SHAPE;
  if (act < 0 || act > 4) return VTX_ERR_SHAPE;
  int rc = gemm_validate(a, mode);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VTX_BF16 && mode == 0 && gemm_skinny_ok(a)) return gemm_skinny_launch(a, st);
  if (dtype == VTX_BF16 && mode == 0 && gemm_glds_ok(N, K) && gemm_glds_enabled()) {
    return gemm_glds_launch(a, st);
  }
  if (dtype == VTX_BF16)
    return mode == 0 ? gemm_pick_bn<bf16, bf16, false, false>(a, 1, st) : gemm_pick_bn<bf16, bf16, false, true>(a, 1, st);
  if (dtype == VTX_F32)
    return mode == 0 ? gemm_pick_bn<float, float, false, false>(a, 1, st) : gemm_pick_bn<float, float, false, true>(a, 1, st);
  return VTX_ERR_DTYPE;
}

// Number of contraction slices (split-K over tokens) of the register-staged wgrad kernels: tiles x slices should just
// fill a whole number of resident block rounds (256 CUs x 2 blocks) -- a partly filled round leaves CUs idle for the
// whole kernel because every block runs the same long k-loop.  (The LDS-DMA kernel: wgrad_glds_slices.)
static int wgrad_slices(int64_t mtok, int N, int Kin) { return wgrad_glds_slices(mtok, wgrad_glds_tiles(N, Kin)); }

static int64_t chunk_of(int64_t mtok, int nz) {
  int64_t chunk = (mtok + nz - 1) / nz;
  return ((chunk + 127) / 128) * 128;
}

size_t vtx_wgrad_workspace(int64_t mtok, int N, int Kin) {
  const int nz = wgrad_slices(mtok, N, Kin);
  return ((size_t)nz * (size_t)N * (size_t)Kin + (size_t)nz * (size_t)N) * sizeof(float);
}

// weight (+ bias) slabs of one weight gradient -> dW (+ dbias), one launch
static int reduce_slabs(const float* slabs, const float* bias_part, float* dW, float* dbias, int N, int Kin, int nz,
                        hipStream_t st) {
  const long long nw = (long long)N * Kin;
  if ((nw & 3) || (dbias && (N & 3))) {                       // (float4 path needs multiples of 4: odd shapes keep the scalar-tail kernel)
    hipLaunchKernelGGL(slab_reduce_kernel, slab_reduce_grid(nw), dim3(256), 0, st, slabs, dW, (int64_t)nw, nz);
    int rc = vtx_check_launch();
    if (rc || !dbias) return rc;
    hipLaunchKernelGGL(slab_reduce_kernel, slab_reduce_grid(N), dim3(256), 0, st, bias_part, dbias, (int64_t)N, nz);
    return vtx_check_launch();
  }
  SlabReduceMulti m;
  for (int i = 0; i < 16; ++i) { m.slabs[i] = nullptr; m.out[i] = nullptr; m.n[i] = 0; }
  m.nz = nz; m.blk0[0] = 0;
  m.slabs[0] = slabs; m.out[0] = dW; m.n[0] = nw;
  m.blk0[1] = (int)(((nw >> 2) + 255) / 256);
  m.nseg = 1;
  if (dbias) {
    m.slabs[1] = bias_part; m.out[1] = dbias; m.n[1] = N;
    m.blk0[2] = m.blk0[1] + (int)((((long long)N >> 2) + 255) / 256);
    m.nseg = 2;
  }
  for (int i = m.nseg; i < 16; ++i) m.blk0[i + 1] = m.blk0[m.nseg];
  hipLaunchKernelGGL(slab_reduce_multi_kernel, dim3((unsigned)m.blk0[m.nseg]), dim3(256), 0, st, m);
  return vtx_check_launch();
}

// dW[N,Kin] = sum_m s[m] * dy[m,N]^T x[m,Kin]  (fp32 out);  dbias[N] = sum_m s[m] * dy[m,:] (same kernel)
int vtx_wgrad(int dtype, const void* dy, const void* x, float* dW, float* dbias, int64_t mtok, int N, int Kin,
              int64_t ld_dy, int64_t ld_x, const float* rowscale, int rows_per_scale, float scale_const,
              void* workspace, size_t ws_bytes, void* stream) {
  if (!dy || !x || !dW || !workspace) return VTX_ERR_NULL;
  if (((uintptr_t)dW & 15) || ((uintptr_t)workspace & 15)) return VTX_ERR_ALIGN;   // 16-byte vector stores into dW / the slabs
  if (mtok <= 0 || mtok > 0x7fffffff) return VTX_ERR_SHAPE;
  if (ws_bytes < vtx_wgrad_workspace(mtok, N, Kin)) return VTX_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int nz = wgrad_slices(mtok, N, Kin);
  GemmArgs a;
  a.A = dy; a.B = x; a.C = (nz == 1) ? (void*)dW : workspace;
  a.M = N; a.N = Kin; a.K = (int)mtok; a.lda = ld_dy; a.ldb = ld_x; a.ldc = Kin;
  a.bias = nullptr; a.resid = nullptr; a.rowscale = nullptr; a.rows_per_scale = 1;
  a.aux_out = nullptr; a.aux_in = nullptr; a.act = 0;
  a.kscale = rowscale; a.k_per_scale = rows_per_scale > 0 ? rows_per_scale : 1; a.kscale_const = scale_const;
  a.kchunk = (int)chunk_of(mtok, nz);
  gemm_args_nomap(a);
  float* bias_part = (float*)workspace + (size_t)nz * N * Kin;
  a.ksum_out = dbias ? (nz == 1 ? dbias : bias_part) : nullptr;
  int rc = gemm_validate(a, 2);
  if (rc) return rc;
  // (the LDS-DMA kernel keeps the DropPath liveness of a slice's samples in a 512-entry table: slices spanning more
  //  samples -- sequences of one or two tokens -- take the register-staged kernel)
  if (wgrad_glds_ok(dtype, N, Kin, rowscale, scale_const) && (rowscale == nullptr || a.kchunk / a.k_per_scale + 2 <= 512)) {
    WgradProbHost hp;
    hp.dy = dy; hp.x = x; hp.slab = (float*)workspace; hp.out = dW; hp.ksum_part = bias_part; hp.ksum_out = dbias;
    hp.rowscale = rowscale; hp.ld_dy = ld_dy; hp.ld_x = ld_x; hp.N = N; hp.Kin = Kin; hp.live_only = 0; hp.perm = nullptr; hp.Mtok = (int)mtok; hp.scale = 1.f;
    rc = wgrad_glds_group_launch(1, &hp, mtok, a.k_per_scale, scale_const, nz, a.kchunk, st);
    if (rc || nz == 1) return rc;
    return reduce_slabs((const float*)workspace, bias_part, dW, dbias, N, Kin, nz, st);
  }
  if (dtype == VTX_BF16) rc = gemm_pick_bn<bf16, float, true, true>(a, nz, st);
  else if (dtype == VTX_F32) rc = gemm_pick_bn<float, float, true, true>(a, nz, st);
  else return VTX_ERR_DTYPE;
  if (rc || nz == 1) return rc;
  return reduce_slabs((const float*)workspace, bias_part, dW, dbias, N, Kin, nz, st);
}

// ---- grouped weight gradients (bf16, every problem eligible for the LDS-DMA kernel: vtx_wgrad_group_ok)
int vtx_wgrad_group_max(void) { return wgrad_glds_max_problems(); }

int vtx_wgrad_group_ok(int dtype, int nprob, const int* N, const int* Kin, int64_t mtok, int has_rowscale,
                       int rows_per_scale, float scale_const) {
  if (nprob < 1 || nprob > wgrad_glds_max_problems() || mtok <= 0 || mtok > 0x7fffffff) return 0;
  int tiles = 0;
  static const float one = 1.f;
  const float* rs = has_rowscale ? &one : nullptr;
  for (int i = 0; i < nprob; ++i) {
    if (!wgrad_glds_ok(dtype, N[i], Kin[i], rs, scale_const)) return 0;
    tiles += wgrad_glds_tiles(N[i], Kin[i]);
  }
  const int wt = wgrad_wide_tiles(nprob, N, Kin);
  const int nz = wt ? wgrad_glds_slices(mtok, wt, true) : wgrad_glds_slices(mtok, tiles);
  if (has_rowscale && chunk_of(mtok, nz) / (rows_per_scale > 0 ? rows_per_scale : 1) + 2 > 512) return 0;
  return 1;
}

static int group_tiles(int nprob, const int* N, const int* Kin) {
  int tiles = 0;
  for (int i = 0; i < nprob; ++i) tiles += wgrad_glds_tiles(N[i], Kin[i]);
  return tiles;
}

// slices of a group under the current options: 128 x 384 tiles (wgrad_wide_tiles) or 128 x 128
static int group_slices(int nprob, const int* N, const int* Kin, int64_t mtok, int* wide) {
  int J = 0;
  const int wt = wgrad_wide_tiles(nprob, N, Kin, &J);
  if (wide) *wide = wt > 0 ? J : 0;
  return wt ? wgrad_glds_slices(mtok, wt, true) : wgrad_glds_slices(mtok, group_tiles(nprob, N, Kin));
}

int vtx_wgrad_group_slices(int nprob, const int* N, const int* Kin, int64_t mtok) {
  if (nprob < 1 || nprob > wgrad_glds_max_problems() || !N || !Kin || mtok <= 0) return 0;
  return group_slices(nprob, N, Kin, mtok, nullptr);
}

size_t vtx_wgrad_group_workspace(int nprob, const int* N, const int* Kin, int64_t mtok) {
  // (the larger of the two tilings' slice counts: a size taken under one setting of WGRAD_WIDE serves the other)
  int nz = wgrad_glds_slices(mtok, group_tiles(nprob, N, Kin));
  const int wt = wgrad_wide_tiles_any(nprob, N, Kin);
  if (wt >= 1) { const int nzw = wgrad_glds_slices(mtok, wt, true); nz = nzw > nz ? nzw : nz; }
  size_t fl = 0;
  for (int i = 0; i < nprob; ++i) fl += (size_t)nz * ((size_t)N[i] * Kin[i] + (size_t)N[i]);
  return (fl + 4) * sizeof(float);
}

int vtx_wgrad_group_mapped(int dtype, int nprob, const void* const* dy, const void* const* x, float* const* dW,
                           float* const* dbias, const int* N, const int* Kin, const int64_t* ld_dy, const int64_t* ld_x,
                           const float* const* rowscale, const int* const* perm, const int* mtok_kept, const float* scale,
                           int rows_per_scale, float scale_const, int64_t mtok, void* workspace, size_t ws_bytes, int ncol,
                           const float* const* col_part, float* const* col_out0, float* const* col_out1, const int* col_nb,
                           const int* col_C, const int* col_ld, int accumulate, void* stream);

int vtx_wgrad_group(int dtype, int nprob, const void* const* dy, const void* const* x, float* const* dW,
                    float* const* dbias, const int* N, const int* Kin, const int64_t* ld_dy, const int64_t* ld_x,
                    const float* const* rowscale, int rows_per_scale, float scale_const, int64_t mtok,
                    void* workspace, size_t ws_bytes, int ncol, const float* const* col_part, float* const* col_out0,
                    float* const* col_out1, const int* col_nb, const int* col_C, const int* col_ld, int accumulate,
                    void* stream) {
  return vtx_wgrad_group_mapped(dtype, nprob, dy, x, dW, dbias, N, Kin, ld_dy, ld_x, rowscale, nullptr, nullptr, nullptr,
                                rows_per_scale, scale_const, mtok, workspace, ws_bytes, ncol, col_part, col_out0, col_out1, col_nb,
                                col_C, col_ld, accumulate, stream);
}

/* vtx_wgrad_group over the KEPT samples of stochastic-depth branches: problem i with perm[i] != NULL contracts over the
 * mtok_kept[i] tokens of its kept samples only, in perm[i] order (logical token t = row perm[i][t / rows_per_scale] *
 * rows_per_scale + t % rows_per_scale of dy[i] and x[i]; dropped samples' rows are never read) and multiplies its result by
 * scale[i]; rowscale[i] must be NULL for it.  perm == NULL: exactly vtx_wgrad_group.  Slices / workspace as for mtok tokens. */
int vtx_wgrad_group_mapped(int dtype, int nprob, const void* const* dy, const void* const* x, float* const* dW,
                           float* const* dbias, const int* N, const int* Kin, const int64_t* ld_dy, const int64_t* ld_x,
                           const float* const* rowscale, const int* const* perm, const int* mtok_kept, const float* scale,
                           int rows_per_scale, float scale_const, int64_t mtok, void* workspace, size_t ws_bytes, int ncol,
                           const float* const* col_part, float* const* col_out0, float* const* col_out1, const int* col_nb,
                           const int* col_C, const int* col_ld, int accumulate, void* stream) {
  if (!dy || !x || !dW || !N || !Kin || !ld_dy || !ld_x || !workspace) return VTX_ERR_NULL;
  bool any_scale = false;
  if (nprob >= 1 && nprob <= wgrad_glds_max_problems())
    for (int i = 0; i < nprob; ++i) any_scale = any_scale || (rowscale && rowscale[i]);
  if (!vtx_wgrad_group_ok(dtype, nprob, N, Kin, mtok, any_scale, rows_per_scale, scale_const)) return VTX_ERR_SHAPE;
  if (ws_bytes < vtx_wgrad_group_workspace(nprob, N, Kin, mtok)) return VTX_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  int wide = 0;
  const int nz = group_slices(nprob, N, Kin, mtok, &wide);
  if (accumulate && nz < 2) return VTX_ERR_SHAPE;          // accumulation lives in the slab reduce (vtx_wgrad_group_slices tells)
  WgradProbHost hp[8];
  float* w = (float*)workspace;
  for (int i = 0; i < nprob; ++i) {
    if (!dy[i] || !x[i] || !dW[i]) return VTX_ERR_NULL;
    if ((uintptr_t)dW[i] & 15) return VTX_ERR_ALIGN;                                 // 16-byte vector stores into dW
    if ((N[i] & 7) || (Kin[i] & 7) || (ld_dy[i] & 7) || (ld_x[i] & 7)) return VTX_ERR_ALIGN;
    hp[i].dy = dy[i]; hp[i].x = x[i]; hp[i].out = dW[i]; hp[i].ksum_out = dbias ? dbias[i] : nullptr;
    hp[i].rowscale = rowscale ? rowscale[i] : nullptr; hp[i].ld_dy = ld_dy[i]; hp[i].ld_x = ld_x[i];
    hp[i].N = N[i]; hp[i].Kin = Kin[i];
    hp[i].live_only = 0;
    hp[i].perm = perm ? perm[i] : nullptr; hp[i].Mtok = (int)mtok; hp[i].scale = 1.f;
    if (hp[i].perm) {
      if (hp[i].rowscale || !mtok_kept || !scale || mtok_kept[i] <= 0 || mtok_kept[i] > mtok || rows_per_scale <= 0 ||
          mtok_kept[i] % rows_per_scale)
        return VTX_ERR_SHAPE;
      hp[i].Mtok = mtok_kept[i]; hp[i].scale = scale[i];
    }
    hp[i].slab = w; w += (size_t)nz * N[i] * Kin[i];
    hp[i].ksum_part = w; w += (size_t)nz * N[i];
  }
  if (ncol < 0 || ncol > 4 || (ncol > 0 && (!col_part || !col_out0 || !col_nb || !col_C || !col_ld))) return VTX_ERR_SHAPE;
  int64_t mmax = 0;                                         // (mapped problems contract over their kept tokens only)
  for (int i = 0; i < nprob; ++i) mmax = hp[i].Mtok > mmax ? hp[i].Mtok : mmax;
  const int kchunk = (int)chunk_of(mmax, nz);
  int rc = wgrad_glds_group_launch(nprob, hp, mmax, rows_per_scale, scale_const, nz, kchunk, st, wide);
  if (rc || (nz == 1 && ncol == 0)) return rc;
  // ONE reduction launch behind the group: all weight and bias slabs (kernel boundary = visibility: the slabs were written
  // with plain stores) and the layer's deferred column reductions (LayerNorm dgamma / dbeta, rel_pos gradient)
  LayerReduce m;
  m.nseg = 0; m.nz = nz; m.ncol = ncol; m.blk0[0] = 0; m.accumulate = accumulate ? 1 : 0;
  if (nz > 1) {
    for (int i = 0; i < nprob; ++i) {
      const long long nw = (long long)N[i] * Kin[i];
      m.slabs[m.nseg] = hp[i].slab; m.out[m.nseg] = hp[i].out; m.n[m.nseg] = nw;
      m.blk0[m.nseg + 1] = m.blk0[m.nseg] + (int)(((nw >> 2) + 1023) / 1024);
      ++m.nseg;
      if (hp[i].ksum_out) {
        m.slabs[m.nseg] = hp[i].ksum_part; m.out[m.nseg] = hp[i].ksum_out; m.n[m.nseg] = N[i];
        m.blk0[m.nseg + 1] = m.blk0[m.nseg] + (int)((((long long)N[i] >> 2) + 1023) / 1024);
        ++m.nseg;
      }
    }
  }
  for (int i = m.nseg; i < 16; ++i) { m.slabs[i] = nullptr; m.out[i] = nullptr; m.n[i] = 0; m.blk0[i + 1] = m.blk0[m.nseg]; }
  m.cblk0[0] = m.blk0[m.nseg];
  for (int i = 0; i < 4; ++i) {
    const bool on = i < ncol;
    if (on && (!col_part[i] || !col_out0[i] || col_nb[i] <= 0 || col_C[i] <= 0)) return VTX_ERR_NULL;
    m.part[i] = on ? col_part[i] : nullptr; m.out0[i] = on ? col_out0[i] : nullptr;
    m.out1[i] = (on && col_out1) ? col_out1[i] : nullptr;
    m.nb[i] = on ? col_nb[i] : 0; m.C[i] = on ? col_C[i] : 0; m.ld[i] = on ? col_ld[i] : 0;
    const int cols = on ? (m.out1[i] ? 2 * m.C[i] : m.C[i]) : 0;
    m.cblk0[i + 1] = m.cblk0[i] + (cols + 31) / 32;
  }
  hipLaunchKernelGGL(layer_reduce_kernel, dim3((unsigned)m.cblk0[4]), dim3(1024), 0, st, m);
  return vtx_check_launch();
}

}  // extern "C"
