// MFMA GEMM family for gfx950 with fused epilogues -- every dense contraction of the hot path
// except the attention cores (QKV / proj / MLP / patch-embed / PatchMerge / classifier linears of
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
// Block = 256 threads = 4 waves (2x2); block tile BM x BN, LDS k-tile of 128 bytes per row
// (64 bf16 / 32 fp32) at a 160-byte row stride (conflict-free ds_read_b128 fragment reads);
// global -> registers -> LDS staging with register prefetch of the next k-tile; transposed
// operands are transposed in registers (4x8 micro-tiles) on their way into LDS so both MFMA
// operands are always read as 8 contiguous k-slots per lane.
#include "vtx_common.h"

struct GemmArgs {
  const void* A; const void* B; void* C;
  int M, N, K;
  int64_t lda, ldb, ldc;
  const float* bias;        // [N] fp32 or null
  const void* resid;        // T [M, ldc] or null: C = resid + rowscale * (acc + bias)
  const float* rowscale;    // per-sample DropPath scale on OUTPUT rows (index row / rows_per_scale) or null
  int rows_per_scale;
  void* aux_out;            // T [M, ldc] or null: pre-activation z when act == 1
  const void* aux_in;       // T [M, ldc]: z when act == 2
  int act;                  // 0 none | 1 C = silu(z), z = acc + bias | 2 C = acc * silu'(aux_in)
  const float* kscale;      // per-sample scale along the CONTRACTION index of a transposed A (wgrad through DropPath)
  int k_per_scale;
  int kchunk;               // contraction length per grid.z slice (multiple of the LDS k-tile)
};

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
                                        const float* __restrict__ kscale, int k_per_scale) {
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
          if (idx < NITEMS && col < tdim && k < kend) {
            Vec8<T> v = load8<T>(base + (int64_t)k * ld + col);
            if (kscale != nullptr) {
              const float s = kscale[k / k_per_scale];
#pragma unroll
              for (int e = 0; e < 8; ++e) v.set(e, v.get(e) * s);
            }
            reg[it][i] = v;
          } else {
            reg[it][i] = vec8_zero<T>();
          }
        }
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
  __shared__ __attribute__((aligned(16))) T lds[(BM + BN) * G::STRIDE];
  T* ldsA = lds;
  T* ldsB = lds + BM * G::STRIDE;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int c_ = lane & 15, g_ = lane >> 4;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int kbeg = blockIdx.z * p.kchunk;
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
  sa.gload(A, p.lda, m0, p.M, kbeg, kend, TA ? p.kscale : nullptr, p.k_per_scale);
  sb.gload(B, p.ldb, n0, p.N, kbeg, kend, nullptr, 1);
  sa.lstore(ldsA);
  sb.lstore(ldsB);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    if (more) {
      const int k0 = kbeg + (kt + 1) * G::BK;
      sa.gload(A, p.lda, m0, p.M, k0, kend, TA ? p.kscale : nullptr, p.k_per_scale);
      sb.gload(B, p.ldb, n0, p.N, k0, kend, nullptr, 1);
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
      sa.lstore(ldsA);
      sb.lstore(ldsB);
      __syncthreads();
    }
  }

  // ---------------- epilogue: acc[i][j][r] = C[row = .. + 4*g_ + r][col = .. + c_]
  TO* Cout = (TO*)p.C + (int64_t)blockIdx.z * p.M * p.ldc;
  const T* resid = (const T*)p.resid;
  const T* aux_in = (const T*)p.aux_in;
  T* aux_out = (T*)p.aux_out;
#pragma unroll
  for (int i = 0; i < WM; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + wm * (BM / 2) + i * 16 + g_ * 4 + r;
      if (row >= p.M) continue;
      const float rsc = p.rowscale ? p.rowscale[row / p.rows_per_scale] : 1.f;
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int col = n0 + wn * (BN / 2) + j * 16 + c_;
        if (col >= p.N) continue;
        const int64_t off = (int64_t)row * p.ldc + col;
        float v = acc[i][j][r];
        if (p.bias) v += p.bias[col];
        if (p.act == 1) {
          const float z = round_to<T>(v);
          if (aux_out) aux_out[off] = from_f32<T>(v);
          v = silu_f(z);
        } else if (p.act == 2) {
          v *= dsilu_f(to_f32<T>(aux_in[off]));
        }
        v *= rsc;
        if (resid) v += to_f32<T>(resid[off]);
        Cout[off] = from_f32<TO>(v);
      }
    }
  }
}

// bias gradient: out[n] = sum_m scale[m / rows_per_scale] * dy[m, n]  -- two-stage, deterministic.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ dy, float* __restrict__ part, int64_t M, int N,
                                                    int64_t ld, const float* __restrict__ rowscale,
                                                    int rows_per_scale, int rows_per_block) {
  // thread -> 8 consecutive columns; threads of a block stride over the rows of the block's slab
  const int nvec = N >> 3;
  const int vpb = min(nvec, 256);                  // vectors handled side by side
  const int rlanes = 256 / vpb;                    // row-parallelism inside the block
  __shared__ float red[256 * 8];
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = min(M, r0 + rows_per_block);
  for (int vbase = 0; vbase < nvec; vbase += vpb) {
    const int v = vbase + threadIdx.x % vpb;
    const int rl = threadIdx.x / vpb;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (v < nvec && rl < rlanes) {
      for (int64_t r = r0 + rl; r < r1; r += rlanes) {
        Vec8<T> t = load8<T>(dy + r * ld + v * 8);
        const float sc = rowscale ? rowscale[r / rows_per_scale] : 1.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += sc * t.get(e);
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[threadIdx.x * 8 + e] = s[e];
    __syncthreads();
    if (threadIdx.x < vpb && v < nvec) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float a = 0.f;
        for (int q = 0; q < rlanes; ++q) a += red[(q * vpb + threadIdx.x) * 8 + e];
        part[(int64_t)blockIdx.x * N + v * 8 + e] = a;
      }
    }
    __syncthreads();
  }
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
  // contiguous-dimension granularity of the 8-element vector loads
  if (mode == 0 && ((a.K & 7) || (a.lda & 7) || (a.ldb & 7))) return VTX_ERR_ALIGN;            // NT
  if (mode == 1 && ((a.K & 7) || (a.N & 7) || (a.lda & 7) || (a.ldb & 7))) return VTX_ERR_ALIGN;  // NN: B[k][n]
  if (mode == 2 && ((a.M & 7) || (a.N & 7) || (a.lda & 7) || (a.ldb & 7))) return VTX_ERR_ALIGN;  // TN
  if (a.act == 2 && !a.aux_in) return VTX_ERR_NULL;
  return VTX_OK;
}

extern "C" {

// y = epilogue(x W^T): mode 0.   dx = epilogue(dy W): mode 1.   (see GemmArgs for the fused epilogue)
int vtx_gemm(int mode, int dtype, const void* A, const void* B, void* C, int M, int N, int K, int64_t lda,
             int64_t ldb, int64_t ldc, const float* bias, const void* resid, const float* rowscale,
             int rows_per_scale, void* aux_out, const void* aux_in, int act, void* stream) {
  GemmArgs a;
  a.A = A; a.B = B; a.C = C; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
  a.bias = bias; a.resid = resid; a.rowscale = rowscale; a.rows_per_scale = rows_per_scale > 0 ? rows_per_scale : 1;
  a.aux_out = aux_out; a.aux_in = aux_in; a.act = act; a.kscale = nullptr; a.k_per_scale = 1;
  a.kchunk = ((K + 127) / 128) * 128;
  if (mode != 0 && mode != 1) return VTX_ERR_SHAPE;
  int rc = gemm_validate(a, mode);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VTX_BF16)
    return mode == 0 ? gemm_pick_bn<bf16, bf16, false, false>(a, 1, st) : gemm_pick_bn<bf16, bf16, false, true>(a, 1, st);
  if (dtype == VTX_F32)
    return mode == 0 ? gemm_pick_bn<float, float, false, false>(a, 1, st) : gemm_pick_bn<float, float, false, true>(a, 1, st);
  return VTX_ERR_DTYPE;
}

// Number of contraction slices the wgrad kernel will use for (Mtok tokens, N x Kin weight).
static int wgrad_slices(int64_t mtok, int N, int Kin) {
  const int tiles = ((N + 127) / 128) * ((Kin + 127) / 128);
  int nz = (512 + tiles - 1) / tiles;
  const int64_t maxz = (mtok + 255) / 256;
  if (nz > maxz) nz = (int)maxz;
  if (nz < 1) nz = 1;
  if (nz > 256) nz = 256;
  return nz;
}

size_t vtx_wgrad_workspace(int64_t mtok, int N, int Kin) {
  const int nz = wgrad_slices(mtok, N, Kin);
  size_t slabs = (size_t)nz * (size_t)N * (size_t)Kin * sizeof(float);
  size_t bias = (size_t)1024 * (size_t)N * sizeof(float);
  return slabs > bias ? slabs : bias;
}

// dW[N,Kin] = sum_m s[m] * dy[m,N]^T x[m,Kin]   (fp32 out);  dbias[N] = sum_m s[m] * dy[m,:] (optional)
int vtx_wgrad(int dtype, const void* dy, const void* x, float* dW, float* dbias, int64_t mtok, int N, int Kin,
              int64_t ld_dy, int64_t ld_x, const float* rowscale, int rows_per_scale, void* workspace, size_t ws_bytes,
              void* stream) {
  if (!dy || !x || !dW || !workspace) return VTX_ERR_NULL;
  if (mtok <= 0 || mtok > 0x7fffffff) return VTX_ERR_SHAPE;
  if (ws_bytes < vtx_wgrad_workspace(mtok, N, Kin)) return VTX_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int nz = wgrad_slices(mtok, N, Kin);
  GemmArgs a;
  a.A = dy; a.B = x; a.C = (nz == 1) ? (void*)dW : workspace;
  a.M = N; a.N = Kin; a.K = (int)mtok; a.lda = ld_dy; a.ldb = ld_x; a.ldc = Kin;
  a.bias = nullptr; a.resid = nullptr; a.rowscale = nullptr; a.rows_per_scale = 1; a.aux_out = nullptr;
  a.aux_in = nullptr; a.act = 0; a.kscale = rowscale; a.k_per_scale = rows_per_scale > 0 ? rows_per_scale : 1;
  int64_t chunk = (mtok + nz - 1) / nz;
  chunk = ((chunk + 127) / 128) * 128;
  a.kchunk = (int)chunk;
  int rc = gemm_validate(a, 2);
  if (rc) return rc;
  if (dtype == VTX_BF16) rc = gemm_pick_bn<bf16, float, true, true>(a, nz, st);
  else if (dtype == VTX_F32) rc = gemm_pick_bn<float, float, true, true>(a, nz, st);
  else return VTX_ERR_DTYPE;
  if (rc) return rc;
  if (nz > 1) {
    const int64_t n = (int64_t)N * Kin;
    int nb = (int)((n + 255) / 256);
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(nb), dim3(256), 0, st, (const float*)workspace, dW, n, nz);
    rc = vtx_check_launch();
    if (rc) return rc;
  }
  if (dbias) {
    if (N & 7) return VTX_ERR_ALIGN;
    int nb = (int)((mtok + 511) / 512);
    if (nb > 1024) nb = 1024;
    const int rpb = (int)((mtok + nb - 1) / nb);
    float* part = (float*)workspace;   // slabs are dead after slab_reduce (same stream => ordered)
    if (dtype == VTX_BF16)
      hipLaunchKernelGGL((colsum_kernel<bf16>), dim3(nb), dim3(256), 0, st, (const bf16*)dy, part, mtok, N, ld_dy,
                         rowscale, a.k_per_scale, rpb);
    else
      hipLaunchKernelGGL((colsum_kernel<float>), dim3(nb), dim3(256), 0, st, (const float*)dy, part, mtok, N, ld_dy,
                         rowscale, a.k_per_scale, rpb);
    rc = vtx_check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(colreduce_kernel, dim3((N + 255) / 256), dim3(256), 0, st, part, dbias, (float*)nullptr, nb, N, N);
    rc = vtx_check_launch();
  }
  return rc;
}

}  // extern "C"
