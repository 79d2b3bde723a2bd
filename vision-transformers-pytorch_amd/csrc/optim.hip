// Fused optimizer tail of the train step (reference train.py:285-299: clip_grad_norm_ -> optimizer.step() with
// torch.optim.AdamW), as two multi-tensor HBM-bound passes over all parameters instead of ~10 foreach launches:
//
//   pass 1  vtx_grad_sqnorm   partial sums of g^2 per 4096-element chunk (one workgroup per chunk, fixed summation
//                             order inside), then ONE workgroup adds the chunk partials in index order
//                             -> total squared L2 norm (deterministic: no atomics);
//   pass 2  vtx_adamw_step    per element: g' = g * min(1, max_norm / (||g|| + 1e-6))   (clip_grad_norm_ semantics)
//                             p *= 1 - lr * wd;  m = b1 m + (1 - b1) g';  v = b2 v + (1 - b2) g'^2;
//                             p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
//                             reading p, g, m, v once and writing p, m, v once (28 B per parameter).
//
// Tensor addresses travel in the KERNEL ARGUMENTS (up to OPT_NT tensors per launch, ~3.4 KB of kernarg), like the
// pointer tables of torch's own multi-tensor kernels: gradient addresses change every backward, and a device-side
// table would need a host-to-device copy per step -- from pageable memory that copy serialises the host with the GPU
// stream (measured: +1.4 ms/step).  Chunks are numbered tensor-major within a launch; a workgroup finds its tensor by
// binary search over the chunk prefix sums.
#include "vtx_common.h"

#define OPT_CHUNK 4096      // elements per workgroup (256 threads x 4 float4)
#define OPT_NT 64           // tensors per launch

struct OptPack {
  float* p[OPT_NT];
  const float* g[OPT_NT];
  float* m[OPT_NT];
  float* v[OPT_NT];
  int64_t numel[OPT_NT];
  int chunk0[OPT_NT];       // first chunk of tensor i within this launch
  float lr[OPT_NT], wd[OPT_NT];
  int n;
};

struct OptDesc { float* p; const float* g; float* m; float* v; int64_t numel; int chunk0; float lr, wd; };

__device__ __forceinline__ OptDesc opt_find(const OptPack& k, int chunk) {
  int lo = 0, hi = k.n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (k.chunk0[mid] <= chunk) lo = mid; else hi = mid - 1;
  }
  return OptDesc{k.p[lo], k.g[lo], k.m[lo], k.v[lo], k.numel[lo], k.chunk0[lo], k.lr[lo], k.wd[lo]};
}

__device__ __forceinline__ float opt_block_sum(float s, float* red) {
  s = group_sum<64>(s);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) red[wave] = s;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];        // fixed order
}

__global__ __launch_bounds__(256) void grad_sqnorm_partial_kernel(const OptPack k, float* __restrict__ partial) {
  __shared__ float red[4];
  const OptDesc d = opt_find(k, blockIdx.x);
  const int64_t base = (int64_t)(blockIdx.x - d.chunk0) * OPT_CHUNK;
  float s = 0.f;
  const bool al = (reinterpret_cast<uintptr_t>(d.g) & 15) == 0;     // DDP bucket views may start at any 4-byte offset
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t e = base + (int64_t)(i * 256 + threadIdx.x) * 4;
    if (al && e + 4 <= d.numel) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(d.g + e);
      s += g[0] * g[0] + g[1] * g[1] + g[2] * g[2] + g[3] * g[3];
    } else {
      for (int64_t k = e; k < d.numel && k < e + 4; ++k) s += d.g[k] * d.g[k];
    }
  }
  s = opt_block_sum(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// one workgroup: out[0] = sum of partial[0..n) in a fixed order (thread-strided, then the block tree)
__global__ __launch_bounds__(256) void grad_sqnorm_final_kernel(const float* __restrict__ partial, int n,
                                                               float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
  s = opt_block_sum(s, red);
  if (threadIdx.x == 0) { out[0] = s; out[1] = sqrtf(s); }
}

__global__ __launch_bounds__(256) void adamw_step_kernel(const OptPack k, const float* __restrict__ norm, float max_norm,
                                                        float beta1, float beta2, float eps, float bc1,
                                                        float rsqrt_bc2) {
  const OptDesc d = opt_find(k, blockIdx.x);
  const int64_t base = (int64_t)(blockIdx.x - d.chunk0) * OPT_CHUNK;
  float coef = 1.f;
  if (max_norm > 0.f) coef = fminf(max_norm / (norm[1] + 1e-6f), 1.f);     // clip_grad_norm_: clamp(max / (total + 1e-6), max = 1)
  const float decay = 1.f - d.lr * d.wd;
  const float step = d.lr / bc1;
  auto upd = [&](float& p, float g, float& m, float& v) {
    g *= coef;
    p *= decay;
    m = beta1 * m + (1.f - beta1) * g;
    v = beta2 * v + (1.f - beta2) * g * g;
    p -= step * (m / (sqrtf(v) * rsqrt_bc2 + eps));
  };
  const bool al = ((reinterpret_cast<uintptr_t>(d.p) | reinterpret_cast<uintptr_t>(d.g) | reinterpret_cast<uintptr_t>(d.m) |
                    reinterpret_cast<uintptr_t>(d.v)) & 15) == 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t e = base + (int64_t)(i * 256 + threadIdx.x) * 4;
    if (al && e + 4 <= d.numel) {
      f32x4 p = *reinterpret_cast<const f32x4*>(d.p + e), g = *reinterpret_cast<const f32x4*>(d.g + e);
      f32x4 m = *reinterpret_cast<const f32x4*>(d.m + e), v = *reinterpret_cast<const f32x4*>(d.v + e);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float pk = p[k], mk = m[k], vk = v[k];
        upd(pk, g[k], mk, vk);
        p[k] = pk; m[k] = mk; v[k] = vk;
      }
      vmem_guard(p); vmem_guard(m); vmem_guard(v);
      *reinterpret_cast<f32x4*>(d.p + e) = p;
      *reinterpret_cast<f32x4*>(d.m + e) = m;
      *reinterpret_cast<f32x4*>(d.v + e) = v;
    } else {
      for (int64_t k = e; k < d.numel && k < e + 4; ++k) {
        float p = d.p[k], m = d.m[k], v = d.v[k];
        upd(p, d.g[k], m, v);
        d.p[k] = p; d.m[k] = m; d.v[k] = v;
      }
    }
  }
}

// EMA / momentum-teacher update  p = m * p + (1 - m) * g   (g = the online parameter): train_dino.py:258-263,
// train_util.py:70-76 -- one multi-tensor pass instead of two foreach launches per call
__global__ __launch_bounds__(256) void ema_kernel(const OptPack k, float m) {
  const OptDesc d = opt_find(k, blockIdx.x);
  const int64_t base = (int64_t)(blockIdx.x - d.chunk0) * OPT_CHUNK;
  const bool al = ((reinterpret_cast<uintptr_t>(d.p) | reinterpret_cast<uintptr_t>(d.g)) & 15) == 0;
  const float w = 1.f - m;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t e = base + (int64_t)(i * 256 + threadIdx.x) * 4;
    if (al && e + 4 <= d.numel) {
      f32x4 p = *reinterpret_cast<const f32x4*>(d.p + e);
      const f32x4 g = *reinterpret_cast<const f32x4*>(d.g + e);
#pragma unroll
      for (int j = 0; j < 4; ++j) p[j] = p[j] * m + g[j] * w;
      vmem_guard(p);
      *reinterpret_cast<f32x4*>(d.p + e) = p;
    } else {
      for (int64_t j = e; j < d.numel && j < e + 4; ++j) d.p[j] = d.p[j] * m + d.g[j] * w;
    }
  }
}

// fill a launch pack from tensors [i0, i0 + cnt); returns its number of chunks
static int opt_pack(OptPack& k, int i0, int cnt, float* const* p, const float* const* g, float* const* m,
                    float* const* v, const int64_t* numel, const float* lr, const float* wd) {
  int chunk0 = 0;
  k.n = cnt;
  for (int j = 0; j < cnt; ++j) {
    const int i = i0 + j;
    k.p[j] = p ? p[i] : nullptr; k.g[j] = g[i]; k.m[j] = m ? m[i] : nullptr; k.v[j] = v ? v[i] : nullptr;
    k.numel[j] = numel[i]; k.chunk0[j] = chunk0;
    k.lr[j] = lr ? lr[i] : 0.f; k.wd[j] = wd ? wd[i] : 0.f;
    chunk0 += (int)((numel[i] + OPT_CHUNK - 1) / OPT_CHUNK);
  }
  return chunk0;
}

extern "C" {

int vtx_opt_chunk(void) { return OPT_CHUNK; }

/* g, numel: HOST arrays of n device pointers / element counts.  norm_out[0] = sum over all tensors of g^2,
 * norm_out[1] = its square root (deterministic); partial: sum_i ceil(numel_i / vtx_opt_chunk()) floats of workspace. */
int vtx_grad_sqnorm(int n, const float* const* g, const int64_t* numel, float* partial, float* norm_out, void* stream) {
  if (!g || !numel || !partial || !norm_out) return VTX_ERR_NULL;
  if (n <= 0) return VTX_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  int done = 0;
  for (int i0 = 0; i0 < n; i0 += OPT_NT) {
    OptPack k;
    const int cnt = n - i0 < OPT_NT ? n - i0 : OPT_NT;
    const int nch = opt_pack(k, i0, cnt, nullptr, g, nullptr, nullptr, numel, nullptr, nullptr);
    if (nch > 0) hipLaunchKernelGGL(grad_sqnorm_partial_kernel, dim3(nch), dim3(256), 0, st, k, partial + done);
    int rc = vtx_check_launch();
    if (rc) return rc;
    done += nch;
  }
  hipLaunchKernelGGL(grad_sqnorm_final_kernel, dim3(1), dim3(256), 0, st, (const float*)partial, done, norm_out);
  return vtx_check_launch();
}

/* One AdamW step (torch.optim.AdamW semantics, step count t >= 1) of n tensors given as HOST arrays of device
 * pointers (p, g, m = exp_avg, v = exp_avg_sq), element counts and per-tensor lr / weight decay; the gradient is scaled
 * by min(1, max_norm / (norm[1] + 1e-6)) when max_norm > 0 (norm = device output of vtx_grad_sqnorm; may be NULL when
 * max_norm <= 0).  Gradients are not modified. */
int vtx_adamw_step(int n, float* const* p, const float* const* g, float* const* m, float* const* v,
                   const int64_t* numel, const float* lr, const float* wd, const float* norm, float max_norm,
                   float beta1, float beta2, float eps, int t, void* stream) {
  if (!p || !g || !m || !v || !numel || !lr || !wd) return VTX_ERR_NULL;
  if (max_norm > 0.f && !norm) return VTX_ERR_NULL;
  if (n <= 0 || t < 1) return VTX_ERR_SHAPE;
  const double bc1 = 1.0 - pow((double)beta1, (double)t);
  const double bc2 = 1.0 - pow((double)beta2, (double)t);
  for (int i0 = 0; i0 < n; i0 += OPT_NT) {
    OptPack k;
    const int cnt = n - i0 < OPT_NT ? n - i0 : OPT_NT;
    const int nch = opt_pack(k, i0, cnt, p, g, m, v, numel, lr, wd);
    if (nch > 0)
      hipLaunchKernelGGL(adamw_step_kernel, dim3(nch), dim3(256), 0, (hipStream_t)stream, k, norm, max_norm, beta1,
                         beta2, eps, (float)bc1, (float)(1.0 / sqrt(bc2)));
    int rc = vtx_check_launch();
    if (rc) return rc;
  }
  return VTX_OK;
}

/* Momentum update of n fp32 tensors (HOST arrays of device pointers): p_i = m * p_i + (1 - m) * g_i. */
int vtx_ema_update(int n, float* const* p, const float* const* g, const int64_t* numel, float m, void* stream) {
  if (!p || !g || !numel) return VTX_ERR_NULL;
  if (n <= 0) return VTX_ERR_SHAPE;
  for (int i0 = 0; i0 < n; i0 += OPT_NT) {
    OptPack k;
    const int cnt = n - i0 < OPT_NT ? n - i0 : OPT_NT;
    const int nch = opt_pack(k, i0, cnt, p, g, nullptr, nullptr, numel, nullptr, nullptr);
    if (nch > 0) hipLaunchKernelGGL(ema_kernel, dim3(nch), dim3(256), 0, (hipStream_t)stream, k, m);
    int rc = vtx_check_launch();
    if (rc) return rc;
  }
  return VTX_OK;
}

}  // extern "C"
