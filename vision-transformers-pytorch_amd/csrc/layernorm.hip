// LayerNorm forward / backward for gfx950 -- HBM-bound row kernels.
//
// Replaces nn.LayerNorm on the hot path (reference: models/vit.py:13, models/swin_transformer.py:12
// block norms eps 1e-6; swin_transformer.py:206,221,277 embed / merge / final norms eps 1e-5).
// A row of C elements is owned by a sub-wave group of G lanes (16/32/64), each lane holding NV
// 8-element vectors in registers, so x is read from HBM exactly once; mean / variance use two
// register passes (biased variance, fp32 statistics) and sub-wave __shfl_xor reductions.
// "merge" addressing folds PatchMerge's 2x2 patchify gather (swin_transformer.py:15-22,224) into
// the row load (forward) and the dx scatter (backward): the 4C-wide row never exists in HBM.
#include "options.h"
#include "vtx_common.h"

struct LnAddr {
  int merge;   // 0: x is [rows, C];  1: x is (B, H, W, Cs) with C = 4*Cs and row = (b, i, j) of the H/2 x W/2 grid
  int H, W, Cs;
  // stochastic-depth compaction (round 3, plain layout only): logical row r is row perm[r / T] * T + r % T of EVERY
  // row-indexed tensor; the forward runs over the kept samples' rows only; the backward runs over all rows, [live, rows)
  // being the rows of dropped samples whose gradient is the residual stream's alone (dx = dres)
  const int* perm;
  int T;
  int64_t live;
  unsigned magic;    // ceil(2^32 / T): r / T == __umulhi(r, magic) (rows x T < 2^32, checked by the entry points)
};
template <bool MAPPED> __device__ __forceinline__ int64_t ln_orow(const LnAddr& a, int64_t row) {
  if constexpr (!MAPPED) return row;
  const int s = (int)__umulhi((unsigned)row, a.magic);
  return (int64_t)a.perm[s] * a.T + ((int)row - s * a.T);
}

__device__ __forceinline__ int64_t ln_src_offset(const LnAddr& a, int64_t row, int col, int C) {
  if (!a.merge) return row * (int64_t)C + col;
  const int Ho = a.H >> 1, Wo = a.W >> 1;
  const int j = (int)(row % Wo);
  const int64_t t = row / Wo;
  const int i = (int)(t % Ho);
  const int64_t b = t / Ho;
  const int s = col / a.Cs, c = col - s * a.Cs;        // segment order (py, px), swin_transformer.py:19-21
  const int py = s >> 1, px = s & 1;
  return ((b * a.H + 2 * i + py) * a.W + 2 * j + px) * (int64_t)a.Cs + c;
}

template <typename T, int G, int NV, bool MAPPED = false, int ROWS = (NV == 1 ? 4 : (NV == 2 ? 2 : 1))>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, T* __restrict__ y,
                                                    float* __restrict__ mean, float* __restrict__ rstd,
                                                    int64_t rows, int C, float eps, LnAddr addr) {
  constexpr int GPB = 256 / G;
  const int lig = threadIdx.x % G, grp = threadIdx.x / G;
  const int nvec = C >> 3;
  float gm[NV][8], bt[NV][8];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int v = lig + k * G;
    // 16-byte loads (parameter tensors are at least 16-byte aligned; scalar loads here cost 16 load instructions per k)
    f32x4 g0 = {0.f, 0.f, 0.f, 0.f}, g1 = g0, b0 = g0, b1 = g0;
    if (v < nvec) {
      g0 = *reinterpret_cast<const f32x4*>(gamma + v * 8); g1 = *reinterpret_cast<const f32x4*>(gamma + v * 8 + 4);
      b0 = *reinterpret_cast<const f32x4*>(beta + v * 8); b1 = *reinterpret_cast<const f32x4*>(beta + v * 8 + 4);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { gm[k][e] = g0[e]; gm[k][4 + e] = g1[e]; bt[k][e] = b0[e]; bt[k][4 + e] = b1[e]; }
  }
  const float invC = 1.f / (float)C;
  // ROWS rows per group per iteration: their loads are issued back to back before any reduction starts, so
  // each lane keeps ROWS x NV 16-byte loads in flight (memory-level parallelism for an HBM-bound kernel)
  const int64_t stride = (int64_t)gridDim.x * GPB;
  for (int64_t row0 = (int64_t)blockIdx.x * GPB + grp; row0 < rows; row0 += stride * ROWS) {
    float xv[ROWS][NV][8];
    int64_t orow[ROWS];
#pragma unroll
    for (int rr = 0; rr < ROWS; ++rr) {
      const int64_t row = row0 + rr * stride;
      orow[rr] = row < rows ? ln_orow<MAPPED>(addr, row) : 0;
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int v = lig + k * G;
        if (v < nvec && row < rows) {
          Vec8<T> t = load8<T>(x + ln_src_offset(addr, orow[rr], v * 8, C));
#pragma unroll
          for (int e = 0; e < 8; ++e) xv[rr][k][e] = t.get(e);
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) xv[rr][k][e] = 0.f;
        }
      }
    }
#pragma unroll
    for (int rr = 0; rr < ROWS; ++rr) {
      const int64_t row = row0 + rr * stride;
      if (row >= rows) break;                       // uniform within the group
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < NV; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) s += xv[rr][k][e];
      const float mu = group_sum<G>(s) * invC;
      float q = 0.f;
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int v = lig + k * G;
        if (v < nvec) {
#pragma unroll
          for (int e = 0; e < 8; ++e) ln_fwd_elem_sq(xv[rr][k][e], mu, q);            // (vtx_common.h: shared with ln_fold.h)
        }
      }
      const float rs = rsqrtf(group_sum<G>(q) * invC + eps);
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int v = lig + k * G;
        if (v < nvec) {
          Vec8<T> o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o.set(e, ln_fwd_elem_out(xv[rr][k][e], mu, rs, gm[k][e], bt[k][e]));
          store8<T>(y + orow[rr] * (int64_t)C + v * 8, o);
        }
      }
      if (lig == 0) { mean[orow[rr]] = mu; rstd[orow[rr]] = rs; }
    }
  }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma  (+ dres: the residual-stream
// gradient that bypasses the norm, fused so the stream gradient is written once).
// dgamma / dbeta: per-lane register accumulation over the block's rows, LDS reduce across the
// block's groups, one deterministic partial row per block; ln_colreduce_kernel sums the partials.
template <typename T, int G, int NV, bool MAPPED = false, int ROWS = 1>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                                    const float* __restrict__ gamma, const T* __restrict__ dres,
                                                    T* __restrict__ dx, float* __restrict__ part,
                                                    int64_t rows, int C, LnAddr addr) {
  constexpr int GPB = 256 / G;
  extern __shared__ __attribute__((aligned(16))) float ln_smem[];   // [2][GPB][C]
  const int lig = threadIdx.x % G, grp = threadIdx.x / G;
  const int nvec = C >> 3;
  float gm[NV][8], dg[NV][8], db[NV][8];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int v = lig + k * G;
    f32x4 g0 = {0.f, 0.f, 0.f, 0.f}, g1 = g0;
    if (v < nvec) { g0 = *reinterpret_cast<const f32x4*>(gamma + v * 8); g1 = *reinterpret_cast<const f32x4*>(gamma + v * 8 + 4); }
#pragma unroll
    for (int e = 0; e < 4; ++e) { gm[k][e] = g0[e]; gm[k][4 + e] = g1[e]; }
#pragma unroll
    for (int e = 0; e < 8; ++e) { dg[k][e] = 0.f; db[k][e] = 0.f; }
  }
  const float invC = 1.f / (float)C;
  // ROWS rows per group and iteration, every load of theirs (x, dy, the residual-stream gradient, the statistics) issued
  // before the first reduction: ROWS x 3 NV 16-byte loads per lane in flight (round 5; one row and the residual gradient
  // loaded after the reductions before).  Rows are visited in the same order by the same lanes: same bits.
  const int64_t stride = (int64_t)gridDim.x * GPB;
  for (int64_t lrow0 = (int64_t)blockIdx.x * GPB + grp; lrow0 < rows; lrow0 += stride * ROWS) {
    Vec8<T> tx[ROWS][NV], td[ROWS][NV], tr[ROWS][NV];
    int64_t orow[ROWS];
    float mu[ROWS], rs[ROWS];
    bool live[ROWS];
#pragma unroll
    for (int rr = 0; rr < ROWS; ++rr) {
      const int64_t lrow = lrow0 + rr * stride;
      const bool in = lrow < rows;
      orow[rr] = in ? ln_orow<MAPPED>(addr, lrow) : 0;
      live[rr] = in && !(MAPPED && lrow >= addr.live);   // (a dropped sample's row: dx = dres; dy, x, statistics never read)
      mu[rr] = live[rr] ? mean[orow[rr]] : 0.f;
      rs[rr] = live[rr] ? rstd[orow[rr]] : 0.f;
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int v = lig + k * G;
        const int64_t off = ln_src_offset(addr, orow[rr], v * 8, C);
        if (v < nvec && live[rr]) {
          tx[rr][k] = load8<T>(x + off);
          td[rr][k] = load8<T>(dy + orow[rr] * (int64_t)C + v * 8);
        } else {
          tx[rr][k] = vec8_zero<T>();
          td[rr][k] = vec8_zero<T>();
        }
        tr[rr][k] = (v < nvec && in && dres != nullptr) ? load8<T>(dres + off) : vec8_zero<T>();
      }
    }
#pragma unroll
    for (int rr = 0; rr < ROWS; ++rr) {
      const int64_t lrow = lrow0 + rr * stride;
      if (lrow >= rows) break;                       // uniform within the group
      if (!live[rr]) {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
          const int v = lig + k * G;
          if (v < nvec) store8<T>(dx + orow[rr] * (int64_t)C + v * 8, tr[rr][k]);
        }
        continue;
      }
      float xh[NV][8], gv[NV][8];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int v = lig + k * G;
        if (v < nvec) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float d = td[rr][k].get(e);
            ln_bwd_elem_accum(d, tx[rr][k].get(e), mu[rr], rs[rr], gm[k][e], s1, s2, xh[k][e], gv[k][e]);   // (vtx_common.h: shared with mlp_fused.hip)
            dg[k][e] += d * xh[k][e];
            db[k][e] += d;
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) { xh[k][e] = 0.f; gv[k][e] = 0.f; }
        }
      }
      const float c1 = group_sum<G>(s1) * invC, c2 = group_sum<G>(s2) * invC;
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int v = lig + k * G;
        if (v < nvec) {
          const int64_t off = ln_src_offset(addr, orow[rr], v * 8, C);
          Vec8<T> o;
          if (dres != nullptr) {
#pragma unroll
            for (int e = 0; e < 8; ++e) o.set(e, ln_bwd_elem_out(tr[rr][k].get(e), rs[rr], gv[k][e], c1, xh[k][e], c2));
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) o.set(e, ln_bwd_elem_out(rs[rr], gv[k][e], c1, xh[k][e], c2));
          }
          store8<T>(dx + off, o);
        }
      }
    }
  }
  float* sg = ln_smem;
  float* sb = ln_smem + GPB * C;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int v = lig + k * G;
    if (v < nvec) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { sg[grp * C + v * 8 + e] = dg[k][e]; sb[grp * C + v * 8 + e] = db[k][e]; }
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int g = 0; g < GPB; ++g) { a += sg[g * C + c]; b += sb[g * C + c]; }
    part[(int64_t)blockIdx.x * 2 * C + c] = a;
    part[(int64_t)blockIdx.x * 2 * C + C + c] = b;
  }
}

// HBM-bound row kernels: up to 1024 blocks of 256 threads, grid-stride over the rows.  The backward's cap is structural
// (every block writes one row of dgamma/dbeta partials, the workspace is sized for 1024); sizing the forward for one
// sweep over 2048 resident blocks measured no better (13.4 -> 14.1 us at 25 088 x 384).
static int ln_grid(int64_t rows, int gpb, int cap) {
  int64_t nb = (rows + gpb - 1) / gpb;
  if (nb > cap) nb = cap;
  if (nb < 1) nb = 1;
  return (int)nb;
}

template <typename T, int G, int NV>
static int ln_fwd_launch(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                         int64_t rows, int C, float eps, LnAddr a, hipStream_t st) {
  const int nb = ln_grid(rows, 256 / G, 1024);
  // (the row map of stochastic-depth compaction is a compile-time variant: the plain kernels carry none of its code)
  // LN_ROWS bit 0: two rows of the three-vector groups (C = 384 on 16 lanes, 768 on 32) in flight per iteration
  if constexpr (NV == 3) {
    if (vtx_opt(VTX_OPT_LN_ROWS) & 1) {
      if (a.perm != nullptr)
        hipLaunchKernelGGL((ln_fwd_kernel<T, G, NV, true, 2>), dim3(nb), dim3(256), 0, st, (const T*)x, gamma, beta, (T*)y, mean,
                           rstd, rows, C, eps, a);
      else
        hipLaunchKernelGGL((ln_fwd_kernel<T, G, NV, false, 2>), dim3(nb), dim3(256), 0, st, (const T*)x, gamma, beta, (T*)y, mean,
                           rstd, rows, C, eps, a);
      return vtx_check_launch();
    }
  }
  if (a.perm != nullptr)
    hipLaunchKernelGGL((ln_fwd_kernel<T, G, NV, true>), dim3(nb), dim3(256), 0, st, (const T*)x, gamma, beta, (T*)y, mean,
                       rstd, rows, C, eps, a);
  else
    hipLaunchKernelGGL((ln_fwd_kernel<T, G, NV, false>), dim3(nb), dim3(256), 0, st, (const T*)x, gamma, beta, (T*)y, mean,
                       rstd, rows, C, eps, a);
  return vtx_check_launch();
}

template <typename T, int G, int NV>
static int ln_bwd_launch(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                         const void* dres, void* dx, float* dgamma, float* dbeta, float* ws, int64_t rows, int C,
                         LnAddr a, hipStream_t st) {
  const int nb = ln_grid(rows, 256 / G, 1024);
  const size_t smem = (size_t)2 * (256 / G) * C * sizeof(float);
  // LN_ROWS bit 1: two rows per group and iteration in the backward (one- and two-vector groups)
  constexpr int R2 = NV <= 2 ? 2 : 1;
  const bool two = (vtx_opt(VTX_OPT_LN_ROWS) & 2) != 0 && R2 == 2;
  if (a.perm != nullptr) {
    if (two)
      hipLaunchKernelGGL((ln_bwd_kernel<T, G, NV, true, R2>), dim3(nb), dim3(256), smem, st, (const T*)dy, (const T*)x, mean, rstd,
                         gamma, (const T*)dres, (T*)dx, ws, rows, C, a);
    else
      hipLaunchKernelGGL((ln_bwd_kernel<T, G, NV, true, 1>), dim3(nb), dim3(256), smem, st, (const T*)dy, (const T*)x, mean, rstd,
                         gamma, (const T*)dres, (T*)dx, ws, rows, C, a);
  } else {
    if (two)
      hipLaunchKernelGGL((ln_bwd_kernel<T, G, NV, false, R2>), dim3(nb), dim3(256), smem, st, (const T*)dy, (const T*)x, mean, rstd,
                         gamma, (const T*)dres, (T*)dx, ws, rows, C, a);
    else
      hipLaunchKernelGGL((ln_bwd_kernel<T, G, NV, false, 1>), dim3(nb), dim3(256), smem, st, (const T*)dy, (const T*)x, mean, rstd,
                         gamma, (const T*)dres, (T*)dx, ws, rows, C, a);
  }
  int rc = vtx_check_launch();
  if (rc || dgamma == nullptr) return rc;                // deferred: the partials stay in ws ([nb][2C]), see vtx_colreduce_multi
  hipLaunchKernelGGL(colreduce_kernel, colreduce_grid(2 * C), dim3(1024), 0, st, ws, dgamma, dbeta, nb, C, 2 * C);
  return vtx_check_launch();
}

// EXACT-FIT groups (option LN_FIT bit FITBIT): C = 384 = 16 lanes x 3 vectors, C = 768 = 32 x 3 -- every lane of the group
// holds data (48 of 64 lanes with <64, 1>), four / two rows per wavefront, and the 16-lane sums stay inside DPP rows
#define LN_DISPATCH(FN, T, FITBIT, ...)                                  \
  do {                                                                   \
    const int nvec = C >> 3;                                             \
    if (vtx_opt(VTX_OPT_LN_FIT) & (FITBIT)) {                            \
      if (nvec == 48) return FN<T, 16, 3>(__VA_ARGS__);                  \
      if (nvec == 96) return FN<T, 32, 3>(__VA_ARGS__);                  \
    }                                                                    \
    if (nvec <= 16) return FN<T, 16, 1>(__VA_ARGS__);                    \
    if (nvec <= 32) return FN<T, 32, 1>(__VA_ARGS__);                    \
    if (nvec <= 64) return FN<T, 64, 1>(__VA_ARGS__);                    \
    if (nvec <= 128) return FN<T, 64, 2>(__VA_ARGS__);                   \
    if (nvec <= 192) return FN<T, 64, 3>(__VA_ARGS__);                   \
    if (nvec <= 256) return FN<T, 64, 4>(__VA_ARGS__);                   \
    return VTX_ERR_SHAPE;                                                \
  } while (0)

static int ln_make_addr(LnAddr& a, int64_t rows, int C, int merge, int H, int W) {
  a.merge = merge; a.H = H; a.W = W; a.Cs = C / 4;
  a.perm = nullptr; a.T = 1; a.live = rows; a.magic = 0;
  if (C <= 0 || (C & 7)) return VTX_ERR_SHAPE;
  if (merge) {
    if ((C & 31) || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return VTX_ERR_SHAPE;
    if (rows % ((int64_t)(H / 2) * (W / 2))) return VTX_ERR_SHAPE;
  }
  return VTX_OK;
}

extern "C" {

int vtx_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                      int64_t rows, int C, float eps, int dtype, int merge, int H, int W, void* stream) {
  if (!x || !gamma || !beta || !y || !mean || !rstd) return VTX_ERR_NULL;
  if (rows <= 0) return VTX_OK;
  LnAddr a;
  int rc = ln_make_addr(a, rows, C, merge, H, W);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VTX_BF16) LN_DISPATCH(ln_fwd_launch, bf16, 1, x, gamma, beta, y, mean, rstd, rows, C, eps, a, st);
  if (dtype == VTX_F32) LN_DISPATCH(ln_fwd_launch, float, 1, x, gamma, beta, y, mean, rstd, rows, C, eps, a, st);
  return VTX_ERR_DTYPE;
}

/* The same over the KEPT samples of a stochastic-depth branch only: logical row r (r < rows = kept samples x T) is row
 * perm[r / T] * T + r % T of x, y, mean and rstd (perm [samples] int32 on the device, kept samples first). */
int vtx_layernorm_fwd_mapped(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                             int64_t rows, int C, float eps, int dtype, const int* perm, int T, void* stream) {
  if (!x || !gamma || !beta || !y || !mean || !rstd || !perm) return VTX_ERR_NULL;
  if (rows <= 0) return VTX_OK;
  if (T <= 0 || rows % T) return VTX_ERR_SHAPE;
  LnAddr a;
  int rc = ln_make_addr(a, rows, C, 0, 0, 0);
  if (rc) return rc;
  a.perm = perm; a.T = T; a.magic = (unsigned)((0x100000000ull + (unsigned)T - 1) / (unsigned)T);
  if ((uint64_t)rows * (uint64_t)T >= 0x100000000ull) return VTX_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VTX_BF16) LN_DISPATCH(ln_fwd_launch, bf16, 1, x, gamma, beta, y, mean, rstd, rows, C, eps, a, st);
  if (dtype == VTX_F32) LN_DISPATCH(ln_fwd_launch, float, 1, x, gamma, beta, y, mean, rstd, rows, C, eps, a, st);
  return VTX_ERR_DTYPE;
}

/* number of partial rows ([blocks][2C] fp32) vtx_layernorm_bwd leaves in its workspace when dgamma == dbeta == NULL */
int vtx_layernorm_bwd_blocks(int64_t rows, int C) {
  const int nvec = C >> 3;
  const int G = nvec <= 16 ? 16 : (nvec <= 32 ? 32 : 64);
  int g2 = G;
  if ((vtx_opt(VTX_OPT_LN_FIT) & 2) && nvec == 48) g2 = 16;
  if ((vtx_opt(VTX_OPT_LN_FIT) & 2) && nvec == 96) g2 = 32;
  return ln_grid(rows, 256 / g2, 1024);
}

size_t vtx_layernorm_bwd_workspace(int64_t rows, int C) {
  (void)rows;
  return (size_t)1024 * 2 * (size_t)C * sizeof(float);
}

int vtx_layernorm_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                      const void* dres, void* dx, float* dgamma, float* dbeta, void* workspace, size_t ws_bytes,
                      int64_t rows, int C, int dtype, int merge, int H, int W, void* stream) {
  if (!dy || !x || !mean || !rstd || !gamma || !dx || !workspace) return VTX_ERR_NULL;
  if ((dgamma == nullptr) != (dbeta == nullptr)) return VTX_ERR_NULL;      // both (reduce now) or neither (deferred)
  if (ws_bytes < vtx_layernorm_bwd_workspace(rows, C)) return VTX_ERR_WORKSPACE;
  LnAddr a;
  int rc = ln_make_addr(a, rows, C, merge, H, W);
  if (rc) return rc;
  if (merge && dres) return VTX_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  float* ws = (float*)workspace;
  if (rows <= 0) return VTX_OK;
  if (dtype == VTX_BF16)
    LN_DISPATCH(ln_bwd_launch, bf16, 2, dy, x, mean, rstd, gamma, dres, dx, dgamma, dbeta, ws, rows, C, a, st);
  if (dtype == VTX_F32)
    LN_DISPATCH(ln_bwd_launch, float, 2, dy, x, mean, rstd, gamma, dres, dx, dgamma, dbeta, ws, rows, C, a, st);
  return VTX_ERR_DTYPE;
}

/* Backward over a stochastic-depth branch: all `rows` logical rows are visited in perm order (see _fwd_mapped); rows
 * [0, live) are the kept samples' (dx = dres + LN'(dy)), rows [live, rows) the dropped samples' (dx = dres: dy, mean, rstd are
 * never read there).  Partials stay in the workspace (deferred reduce, vtx_layernorm_bwd_blocks(rows, C) rows). */
int vtx_layernorm_bwd_mapped(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                             const void* dres, void* dx, void* workspace, size_t ws_bytes, int64_t rows, int64_t live, int C,
                             int dtype, const int* perm, int T, void* stream) {
  if (!dy || !x || !mean || !rstd || !gamma || !dx || !workspace || !perm || !dres) return VTX_ERR_NULL;
  if (ws_bytes < vtx_layernorm_bwd_workspace(rows, C)) return VTX_ERR_WORKSPACE;
  if (T <= 0 || rows <= 0 || rows % T || live < 0 || live > rows || live % T) return VTX_ERR_SHAPE;
  LnAddr a;
  int rc = ln_make_addr(a, rows, C, 0, 0, 0);
  if (rc) return rc;
  a.perm = perm; a.T = T; a.live = live; a.magic = (unsigned)((0x100000000ull + (unsigned)T - 1) / (unsigned)T);
  if ((uint64_t)rows * (uint64_t)T >= 0x100000000ull) return VTX_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  float* ws = (float*)workspace;
  float* nul = nullptr;
  if (dtype == VTX_BF16)
    LN_DISPATCH(ln_bwd_launch, bf16, 2, dy, x, mean, rstd, gamma, dres, dx, nul, nul, ws, rows, C, a, st);
  if (dtype == VTX_F32)
    LN_DISPATCH(ln_bwd_launch, float, 2, dy, x, mean, rstd, gamma, dres, dx, nul, nul, ws, rows, C, a, st);
  return VTX_ERR_DTYPE;
}

/* Several fixed-order column reductions in one launch: out0[i][c] = sum_b part[i][b][c] (c < C[i]), out1[i][c - C[i]] for
 * C[i] <= c < 2 C[i] when out1[i] != NULL; nb[i] partial rows of stride ld[i].  n <= 4.  Same summation order (same bits)
 * as the reductions vtx_layernorm_bwd / vtx_wattn_bwd run themselves when given their outputs. */
int vtx_colreduce_multi(int n, const float* const* part, float* const* out0, float* const* out1, const int* nb,
                        const int* C, const int* ld, void* stream) {
  if (!part || !out0 || !nb || !C || !ld) return VTX_ERR_NULL;
  if (n < 1 || n > 4) return VTX_ERR_SHAPE;
  ColReduceMulti m;
  int blk = 0;
  for (int i = 0; i < 4; ++i) {
    const int k = i < n ? i : 0;
    if (!part[k] || !out0[k] || nb[k] <= 0 || C[k] <= 0) return VTX_ERR_NULL;
    m.part[i] = part[k]; m.out0[i] = out0[k]; m.out1[i] = out1 ? out1[k] : nullptr;
    m.nb[i] = nb[k]; m.C[i] = C[k]; m.ld[i] = ld[k];
    m.blk0[i] = blk;
    if (i < n) blk += ((m.out1[i] ? 2 * C[k] : C[k]) + 31) / 32;
  }
  m.blk0[4] = blk;
  for (int i = n; i < 4; ++i) m.blk0[i] = blk;
  m.n = n;
  hipLaunchKernelGGL(colreduce_multi_kernel, dim3(blk), dim3(1024), 0, (hipStream_t)stream, m);
  return vtx_check_launch();
}

}  // extern "C"
