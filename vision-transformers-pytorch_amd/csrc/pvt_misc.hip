// Data-movement kernels of the PVT path (all HBM-bound, 16-byte vectors): non-overlapping patch gather / scatter on
// token-major features (the im2col of the stride = kernel convolutions) and the position-embedding add with an
// optional cls token.
#include "vtx_common.h"

// ---------------------------------------------------------------------------------------------
// x [B, skip + H*W, C]  ->  out [B*(H/p)*(W/p), p*p*C], column order (py, px, c).
// One thread per 8-channel vector, indexed in INPUT order (coalesced reads; writes are C*sizeof(T)-byte runs).
template <typename T, bool BWD, bool ACC>
__global__ void patchify_kernel(const T* __restrict__ src, T* __restrict__ dst, int H, int W, int C, int p, int skip,
                                int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int cv = C >> 3;
  const int v = (int)(idx % cv);
  int64_t t = idx / cv;
  const int x = (int)(t % W); t /= W;
  const int y = (int)(t % H);
  const int64_t b = t / H;
  const int i = y / p, py = y - i * p, j = x / p, px = x - j * p;
  const int64_t tok = (b * (skip + (int64_t)H * W) + skip + (int64_t)y * W + x) * C + v * 8;          // feature side
  const int64_t pat = ((b * (H / p) + i) * (W / p) + j) * ((int64_t)p * p * C) + (int64_t)(py * p + px) * C + v * 8;
  if (!BWD) {
    store8<T>(dst + pat, load8<T>(src + tok));
  } else {
    Vec8<T> g = load8<T>(src + pat);
    if (ACC) {
      Vec8<T> a = load8<T>(dst + tok);
#pragma unroll
      for (int e = 0; e < 8; ++e) g.set(e, g.get(e) + a.get(e));
    }
    store8<T>(dst + tok, g);
  }
}

// ---------------------------------------------------------------------------------------------
// out[b, 0] = cls + pos[0] (cls != null); out[b, s + t] = x[b, t] + pos[s + t]   (pvt.py:133-137)
template <typename T>
__global__ void add_pos_fwd_kernel(const T* __restrict__ x, const float* __restrict__ cls, const float* __restrict__ pos,
                                   T* __restrict__ out, int T_, int C, int s, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // over (b, token of out, vector)
  if (idx >= total) return;
  const int cv = C >> 3;
  const int v = (int)(idx % cv);
  const int64_t bt = idx / cv;
  const int L = T_ + s;
  const int t = (int)(bt % L);
  const int64_t b = bt / L;
  const float* pp = pos + (int64_t)t * C + v * 8;
  Vec8<T> o;
  if (s && t == 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) o.set(e, cls[v * 8 + e] + pp[e]);
  } else {
    Vec8<T> a = load8<T>(x + (b * T_ + (t - s)) * C + v * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) o.set(e, a.get(e) + pp[e]);
  }
  store8<T>(out + bt * C + v * 8, o);
}

// dx[b, t] = dout[b, s + t]; dpos[t'] = sum_b dout[b, t'] (fixed order over b); dcls = sum_b dout[b, 0].
// 16 column vectors x 16 batch lanes per workgroup, partial sums added in lane order through LDS (misc.hip has the twin).
template <typename T>
__global__ __launch_bounds__(256) void add_pos_bwd_kernel(const T* __restrict__ dout, T* __restrict__ dx,
                                                         float* __restrict__ dcls, float* __restrict__ dpos, int B, int T_,
                                                         int C, int s) {
  __shared__ float red[16][16][9];
  const int cv = C >> 3;
  const int L = T_ + s;
  const int vl = threadIdx.x & 15, bl = threadIdx.x >> 4;
  const int idx = blockIdx.x * 16 + vl;
  const bool live = idx < L * cv;
  const int t = live ? idx / cv : 0, v = live ? idx - t * cv : 0;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (live)
    for (int b = bl; b < B; b += 16) {
      Vec8<T> a = load8<T>(dout + ((int64_t)b * L + t) * C + v * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += a.get(e);
      if (t >= s) store8<T>(dx + ((int64_t)b * T_ + (t - s)) * C + v * 8, a);
    }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[bl][vl][e] = acc[e];
  __syncthreads();
  if (bl == 0 && live) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float sum = 0.f;
      for (int j = 0; j < 16; ++j) sum += red[j][vl][e];
      dpos[(int64_t)t * C + v * 8 + e] = sum;
      if (s && t == 0) dcls[v * 8 + e] = sum;
    }
  }
}

extern "C" {

int vtx_patchify_fwd(const void* x, void* out, int B, int H, int W, int C, int p, int skip, int dtype, void* stream) {
  if (!x || !out) return VTX_ERR_NULL;
  if (B <= 0 || p <= 0 || H % p || W % p || skip < 0) return VTX_ERR_SHAPE;
  if (C & 7) return VTX_ERR_ALIGN;
  const int64_t total = (int64_t)B * H * W * (C >> 3);
  dim3 grid((unsigned)((total + 255) / 256));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VTX_BF16)
    hipLaunchKernelGGL((patchify_kernel<bf16, false, false>), grid, dim3(256), 0, st, (const bf16*)x, (bf16*)out, H, W, C, p, skip, total);
  else if (dtype == VTX_F32)
    hipLaunchKernelGGL((patchify_kernel<float, false, false>), grid, dim3(256), 0, st, (const float*)x, (float*)out, H, W, C, p, skip, total);
  else return VTX_ERR_DTYPE;
  return vtx_check_launch();
}

int vtx_patchify_bwd(const void* dout, void* dx, int B, int H, int W, int C, int p, int skip, int accumulate, int dtype,
                     void* stream) {
  if (!dout || !dx) return VTX_ERR_NULL;
  if (B <= 0 || p <= 0 || H % p || W % p || skip < 0) return VTX_ERR_SHAPE;
  if (C & 7) return VTX_ERR_ALIGN;
  const int64_t total = (int64_t)B * H * W * (C >> 3);
  dim3 grid((unsigned)((total + 255) / 256));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VTX_BF16) {
    if (accumulate) hipLaunchKernelGGL((patchify_kernel<bf16, true, true>), grid, dim3(256), 0, st, (const bf16*)dout, (bf16*)dx, H, W, C, p, skip, total);
    else hipLaunchKernelGGL((patchify_kernel<bf16, true, false>), grid, dim3(256), 0, st, (const bf16*)dout, (bf16*)dx, H, W, C, p, skip, total);
  } else if (dtype == VTX_F32) {
    if (accumulate) hipLaunchKernelGGL((patchify_kernel<float, true, true>), grid, dim3(256), 0, st, (const float*)dout, (float*)dx, H, W, C, p, skip, total);
    else hipLaunchKernelGGL((patchify_kernel<float, true, false>), grid, dim3(256), 0, st, (const float*)dout, (float*)dx, H, W, C, p, skip, total);
  } else return VTX_ERR_DTYPE;
  return vtx_check_launch();
}

int vtx_add_pos_fwd(const void* x, const float* cls, const float* pos, void* out, int B, int T, int C, int dtype,
                    void* stream) {
  if (!x || !pos || !out) return VTX_ERR_NULL;
  if (B <= 0 || T <= 0) return VTX_ERR_SHAPE;
  if (C & 7) return VTX_ERR_ALIGN;
  const int s = cls ? 1 : 0;
  const int64_t total = (int64_t)B * (T + s) * (C >> 3);
  dim3 grid((unsigned)((total + 255) / 256));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VTX_BF16)
    hipLaunchKernelGGL((add_pos_fwd_kernel<bf16>), grid, dim3(256), 0, st, (const bf16*)x, cls, pos, (bf16*)out, T, C, s, total);
  else if (dtype == VTX_F32)
    hipLaunchKernelGGL((add_pos_fwd_kernel<float>), grid, dim3(256), 0, st, (const float*)x, cls, pos, (float*)out, T, C, s, total);
  else return VTX_ERR_DTYPE;
  return vtx_check_launch();
}

int vtx_add_pos_bwd(const void* dout, void* dx, float* dcls, float* dpos, int B, int T, int C, int dtype, void* stream) {
  if (!dout || !dx || !dpos) return VTX_ERR_NULL;
  if (B <= 0 || T <= 0) return VTX_ERR_SHAPE;
  if (C & 7) return VTX_ERR_ALIGN;
  const int s = dcls ? 1 : 0;
  const int n = (T + s) * (C >> 3);
  dim3 grid((n + 15) / 16);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VTX_BF16)
    hipLaunchKernelGGL((add_pos_bwd_kernel<bf16>), grid, dim3(256), 0, st, (const bf16*)dout, (bf16*)dx, dcls, dpos, B, T, C, s);
  else if (dtype == VTX_F32)
    hipLaunchKernelGGL((add_pos_bwd_kernel<float>), grid, dim3(256), 0, st, (const float*)dout, (float*)dx, dcls, dpos, B, T, C, s);
  else return VTX_ERR_DTYPE;
  return vtx_check_launch();
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// Row-wise L2 normalisation  y = x / max(||x||_2, eps)  (F.normalize(dim=-1, p=2) of the DINO head, reference
// models/vit.py:258) and its backward  dx = (dy - y * sum(y o dy)) / max(||x||, eps); one wavefront per row.
template <typename T>
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, float* __restrict__ nrm,
                                                        int64_t rows, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* xr = x + row * C;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) { const float v = to_f32<T>(xr[c]); s += v * v; }
  const float n = fmaxf(sqrtf(group_sum<64>(s)), eps);
  const float inv = 1.f / n;
  for (int c = lane; c < C; c += 64) {
    float o = to_f32<T>(xr[c]) * inv;
    vmem_guard(o);                       // (packed math in front of a store: vtx_common.h)
    y[row * C + c] = from_f32<T>(o);
  }
  if (lane == 0) nrm[row] = n;
}

template <typename T>
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y,
                                                        const float* __restrict__ nrm, T* __restrict__ dx, int64_t rows,
                                                        int C) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += to_f32<T>(y[row * C + c]) * to_f32<T>(dy[row * C + c]);
  s = group_sum<64>(s);
  const float inv = 1.f / nrm[row];
  for (int c = lane; c < C; c += 64) {
    float o = (to_f32<T>(dy[row * C + c]) - to_f32<T>(y[row * C + c]) * s) * inv;
    vmem_guard(o);
    dx[row * C + c] = from_f32<T>(o);
  }
}

extern "C" {

int vtx_l2norm_fwd(const void* x, void* y, float* nrm, int64_t rows, int C, float eps, int dtype, void* stream) {
  if (!x || !y || !nrm) return VTX_ERR_NULL;
  if (rows <= 0 || C <= 0) return VTX_ERR_SHAPE;
  dim3 grid((unsigned)((rows + 3) / 4));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VTX_BF16) hipLaunchKernelGGL((l2norm_fwd_kernel<bf16>), grid, dim3(256), 0, st, (const bf16*)x, (bf16*)y, nrm, rows, C, eps);
  else if (dtype == VTX_F32) hipLaunchKernelGGL((l2norm_fwd_kernel<float>), grid, dim3(256), 0, st, (const float*)x, (float*)y, nrm, rows, C, eps);
  else return VTX_ERR_DTYPE;
  return vtx_check_launch();
}

int vtx_l2norm_bwd(const void* dy, const void* y, const float* nrm, void* dx, int64_t rows, int C, int dtype, void* stream) {
  if (!dy || !y || !nrm || !dx) return VTX_ERR_NULL;
  if (rows <= 0 || C <= 0) return VTX_ERR_SHAPE;
  dim3 grid((unsigned)((rows + 3) / 4));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VTX_BF16) hipLaunchKernelGGL((l2norm_bwd_kernel<bf16>), grid, dim3(256), 0, st, (const bf16*)dy, (const bf16*)y, nrm, (bf16*)dx, rows, C);
  else if (dtype == VTX_F32) hipLaunchKernelGGL((l2norm_bwd_kernel<float>), grid, dim3(256), 0, st, (const float*)dy, (const float*)y, nrm, (float*)dx, rows, C);
  else return VTX_ERR_DTYPE;
  return vtx_check_launch();
}

}  // extern "C"
