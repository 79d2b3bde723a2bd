// Halo (blocked local) attention for gfx950 -- reference models/halo_transformer.py:22-115 (MultiHeadedHaloAttention): the queries of
// a window x window block attend to the keys / values of the (window + 2 halo)^2 neighbourhood around it, out-of-image positions
// being ZERO key / value rows that still take part in the softmax (F.unfold's zero padding, halo_transformer.py:70-76), plus a
// relative-position term rel_pos[pos[q][k]][head].
//
// Data layout in HBM: the QKV projection stays the channels-last map [B, H, W, 3 h D] (q | k | v channel blocks); this file gathers
//   q_win  [B nW, window^2, h D]                  (the window partition: a permutation)
//   kv_win [B nW, (window + 2 halo)^2, 2 h D]     (the neighbourhoods: every map token appears in up to ceil((w + 2a) / w)^2 of them)
// and the attention itself is the cross-attention of csrc/attention_long.hip (key blocks of 64, online softmax, bias term, dK / dV
// per neighbourhood, bias gradient summed over all problems in fixed order).  The backward scatters dq_win back (permutation) and
// SUMS dkv_win over the neighbourhoods that contain a token -- one thread per map token and 8 channels walks the covering windows in
// a fixed order: deterministic, no atomics.  Every kernel here is an HBM stream.
#include "vtx_common.h"

struct HaloGeom {
  int B, H, W, win, halo, nWy, nWx, side, nc;   // side = win + 2 halo; nc = channels gathered
  int64_t ld;                                   // row stride of the map (elements), c0 already added to the base pointer
};

// dst[((b nW + n) side^2 + kk) nc + c] = map[b, wi win + ky - halo, wj win + kx - halo, c]  (0 outside the map)
template <typename T>
__global__ __launch_bounds__(256) void halo_gather_kernel(const T* __restrict__ map, T* __restrict__ dst, int64_t total, HaloGeom g) {
  const int nv = g.nc >> 3;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int v = (int)(i % nv);
    int64_t r = i / nv;
    const int kk = (int)(r % (g.side * g.side));
    r /= g.side * g.side;
    const int n = (int)(r % (g.nWy * g.nWx)), b = (int)(r / (g.nWy * g.nWx));
    const int wi = n / g.nWx, wj = n - wi * g.nWx, ky = kk / g.side, kx = kk - ky * g.side;
    const int y = wi * g.win + ky - g.halo, x = wj * g.win + kx - g.halo;
    Vec8<T> val = vec8_zero<T>();
    if (y >= 0 && y < g.H && x >= 0 && x < g.W) val = load8<T>(map + (((int64_t)b * g.H + y) * g.W + x) * g.ld + v * 8);
    store8<T>(dst + i * 8, val);
  }
}

// map[b, y, x, c] = sum over the windows (wi, wj) whose neighbourhood contains (y, x), in ascending (wi, wj) order, of
// src[((b nW + n) side^2 + kk) nc + c] -- the adjoint of the gather (halo = 0: the inverse permutation)
template <typename T>
__global__ __launch_bounds__(256) void halo_scatter_kernel(const T* __restrict__ src, T* __restrict__ map, int64_t total, HaloGeom g) {
  const int nv = g.nc >> 3;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int v = (int)(i % nv);
    int64_t r = i / nv;
    const int x = (int)(r % g.W);
    r /= g.W;
    const int y = (int)(r % g.H), b = (int)(r / g.H);
    // windows wi with wi win - halo <= y < wi win + win + halo
    const int wi0 = max(0, (y - g.halo - g.win + 1 + g.win * 4) / g.win - 4), wi1 = min(g.nWy - 1, (y + g.halo) / g.win);
    const int wj0 = max(0, (x - g.halo - g.win + 1 + g.win * 4) / g.win - 4), wj1 = min(g.nWx - 1, (x + g.halo) / g.win);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int wi = wi0; wi <= wi1; ++wi)
      for (int wj = wj0; wj <= wj1; ++wj) {
        const int ky = y - wi * g.win + g.halo, kx = x - wj * g.win + g.halo;
        if (ky < 0 || ky >= g.side || kx < 0 || kx >= g.side) continue;
        const int64_t row = ((int64_t)b * g.nWy * g.nWx + wi * g.nWx + wj) * (g.side * g.side) + ky * g.side + kx;
        const Vec8<T> t = load8<T>(src + row * g.nc + v * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += t.get(e);
      }
    Vec8<T> o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o.set(e, acc[e]);
    store8<T>(map + (((int64_t)b * g.H + y) * g.W + x) * g.ld + v * 8, o);
  }
}

static int halo_geom(HaloGeom& g, int B, int H, int W, int64_t ld, int c0, int nc, int win, int halo) {
  if (B <= 0 || H <= 0 || W <= 0 || win <= 0 || halo < 0 || H % win || W % win) return VTX_ERR_SHAPE;
  if (nc <= 0 || (nc & 7) || (c0 & 7) || (ld & 7) || c0 < 0 || c0 + nc > ld) return VTX_ERR_ALIGN;
  if (halo > 3 * win) return VTX_ERR_SHAPE;          // (the covering-window search above looks 4 windows back)
  g.B = B; g.H = H; g.W = W; g.win = win; g.halo = halo; g.nWy = H / win; g.nWx = W / win; g.side = win + 2 * halo; g.nc = nc; g.ld = ld;
  return VTX_OK;
}
static unsigned halo_grid(int64_t total) {
  int64_t nb = (total + 255) / 256;
  return (unsigned)(nb > 65536 ? 65536 : (nb < 1 ? 1 : nb));
}

bool lattn_ok(int dtype, int D);
int lattn_cross_fwd_launch(const void* q, const void* kv, void* o, float* lse, int B, int Lq, int Lk, int nH, int D, int dtype,
                           hipStream_t st, const float* bias = nullptr, const DropArgs* da = nullptr);
int lattn_cross_bwd_launch(const void* q, const void* kv, const void* o, const void* dout, const float* lse, void* dq, void* dkv,
                           float* ws, int B, int Lq, int Lk, int nH, int D, int dtype, hipStream_t st, const float* bias = nullptr,
                           float* dbias = nullptr, const DropArgs* da = nullptr);

extern "C" {

/* Window / neighbourhood gather of a channels-last map: dst [B * (H / win) * (W / win), (win + 2 halo)^2, nc] from channels
 * [c0, c0 + nc) of map [B, H, W, ld]; positions outside the map give zero rows (reference models/halo_transformer.py:70-76:
 * F.unfold(kernel win + 2 halo, stride win, padding halo); halo = 0 is the plain window partition of halo_transformer.py:63-68). */
int vtx_window_gather(const void* map, void* dst, int B, int H, int W, int64_t ld, int c0, int nc, int win, int halo, int dtype,
                      void* stream) {
  if (!map || !dst) return VTX_ERR_NULL;
  HaloGeom g;
  int rc = halo_geom(g, B, H, W, ld, c0, nc, win, halo);
  if (rc) return rc;
  const int64_t total = (int64_t)B * g.nWy * g.nWx * g.side * g.side * (nc >> 3);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VTX_BF16)
    hipLaunchKernelGGL(halo_gather_kernel<bf16>, dim3(halo_grid(total)), dim3(256), 0, st, (const bf16*)map + c0, (bf16*)dst, total, g);
  else if (dtype == VTX_F32)
    hipLaunchKernelGGL(halo_gather_kernel<float>, dim3(halo_grid(total)), dim3(256), 0, st, (const float*)map + c0, (float*)dst, total, g);
  else
    return VTX_ERR_DTYPE;
  return vtx_check_launch();
}

/* The adjoint: channels [c0, c0 + nc) of map [B, H, W, ld] = sum over the windows whose neighbourhood holds the token of the rows of
 * src (fixed order: deterministic); other channels of the map are not touched.  halo = 0: the inverse window partition. */
int vtx_window_scatter(const void* src, void* map, int B, int H, int W, int64_t ld, int c0, int nc, int win, int halo, int dtype,
                       void* stream) {
  if (!map || !src) return VTX_ERR_NULL;
  HaloGeom g;
  int rc = halo_geom(g, B, H, W, ld, c0, nc, win, halo);
  if (rc) return rc;
  const int64_t total = (int64_t)B * H * W * (nc >> 3);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VTX_BF16)
    hipLaunchKernelGGL(halo_scatter_kernel<bf16>, dim3(halo_grid(total)), dim3(256), 0, st, (const bf16*)src, (bf16*)map + c0, total, g);
  else if (dtype == VTX_F32)
    hipLaunchKernelGGL(halo_scatter_kernel<float>, dim3(halo_grid(total)), dim3(256), 0, st, (const float*)src, (float*)map + c0, total, g);
  else
    return VTX_ERR_DTYPE;
  return vtx_check_launch();
}

/* Cross attention with an additive score term: q [B * Lq, nH * D], kv [B * Lk, 2 * nH * D] (k | v halves), bias [nH][Lq][Lk] fp32 or
 * NULL, o [B * Lq, nH * D], lse [B * nH * Lq]; softmax(q k^T / sqrt(D) + bias) v per (problem, head), any Lq / Lk, D = 32 | 64
 * (reference models/halo_transformer.py:93-104 with B = images x windows). */
int vtx_xattn_fwd(const void* q, const void* kv, void* o, float* lse, const float* bias, int B, int Lq, int Lk, int nH, int D, int dtype,
                  void* stream) {
  if (!q || !kv || !o || !lse) return VTX_ERR_NULL;
  if (B <= 0 || Lq <= 0 || Lk <= 0 || nH <= 0 || !lattn_ok(dtype, D) || (int64_t)B * Lq >= 0x7fffffff || (int64_t)B * Lk >= 0x7fffffff)
    return VTX_ERR_SHAPE;
  return lattn_cross_fwd_launch(q, kv, o, lse, B, Lq, Lk, nH, D, dtype, (hipStream_t)stream, bias);
}
size_t vtx_xattn_bwd_workspace(int B, int Lq, int nH) { return (size_t)B * nH * Lq * sizeof(float); }
/* dq, dkv fully written; dbias [nH][Lq][Lk] fp32 = sum over the B problems of dS (required with a bias); deterministic. */
int vtx_xattn_bwd(const void* q, const void* kv, const void* o, const void* dout, const float* lse, const float* bias, void* dq,
                  void* dkv, float* dbias, void* workspace, size_t ws_bytes, int B, int Lq, int Lk, int nH, int D, int dtype,
                  void* stream) {
  if (!q || !kv || !o || !dout || !lse || !dq || !dkv || !workspace) return VTX_ERR_NULL;
  if ((bias != nullptr) != (dbias != nullptr)) return VTX_ERR_NULL;
  if (B <= 0 || Lq <= 0 || Lk <= 0 || nH <= 0 || !lattn_ok(dtype, D) || (int64_t)B * Lq >= 0x7fffffff || (int64_t)B * Lk >= 0x7fffffff)
    return VTX_ERR_SHAPE;
  if (ws_bytes < vtx_xattn_bwd_workspace(B, Lq, nH)) return VTX_ERR_WORKSPACE;
  return lattn_cross_bwd_launch(q, kv, o, dout, lse, dq, dkv, (float*)workspace, B, Lq, Lk, nH, D, dtype, (hipStream_t)stream, bias, dbias);
}

/* The same with dropout of the attention probabilities (halo_transformer.py:101: F.dropout(attn, p, training)): see
 * vtx_attention_fwd_drop; problem = b * nH + head over the B problems, cell = query * Lk + key, keep [B * nH][Lq][Lk]. */
int vtx_xattn_fwd_drop(const void* q, const void* kv, void* o, float* lse, const float* bias, int B, int Lq, int Lk, int nH, int D,
                       int dtype, float drop_p, uint64_t seed, const uint8_t* keep, void* stream) {
  if (!q || !kv || !o || !lse) return VTX_ERR_NULL;
  if (B <= 0 || Lq <= 0 || Lk <= 0 || nH <= 0 || !lattn_ok(dtype, D) || (int64_t)B * Lq >= 0x7fffffff || (int64_t)B * Lk >= 0x7fffffff)
    return VTX_ERR_SHAPE;
  DropArgs da;
  int rc = drop_args(da, drop_p, seed, keep, Lq, Lk);
  if (rc) return rc;
  return lattn_cross_fwd_launch(q, kv, o, lse, B, Lq, Lk, nH, D, dtype, (hipStream_t)stream, bias, &da);
}
int vtx_xattn_bwd_drop(const void* q, const void* kv, const void* o, const void* dout, const float* lse, const float* bias, void* dq,
                       void* dkv, float* dbias, void* workspace, size_t ws_bytes, int B, int Lq, int Lk, int nH, int D, int dtype,
                       float drop_p, uint64_t seed, const uint8_t* keep, void* stream) {
  if (!q || !kv || !o || !dout || !lse || !dq || !dkv || !workspace) return VTX_ERR_NULL;
  if ((bias != nullptr) != (dbias != nullptr)) return VTX_ERR_NULL;
  if (B <= 0 || Lq <= 0 || Lk <= 0 || nH <= 0 || !lattn_ok(dtype, D) || (int64_t)B * Lq >= 0x7fffffff || (int64_t)B * Lk >= 0x7fffffff)
    return VTX_ERR_SHAPE;
  if (ws_bytes < vtx_xattn_bwd_workspace(B, Lq, nH)) return VTX_ERR_WORKSPACE;
  DropArgs da;
  int rc = drop_args(da, drop_p, seed, keep, Lq, Lk);
  if (rc) return rc;
  return lattn_cross_bwd_launch(q, kv, o, dout, lse, dq, dkv, (float*)workspace, B, Lq, Lk, nH, D, dtype, (hipStream_t)stream, bias, dbias,
                                &da);
}

}  // extern "C"
