// DINO self-distillation loss (reference loss.py:89-152) as three HBM-bound kernels over the (crops x batch) x K
// logits (K = 65 536 in config/dino_deit-s-16.conf: one student logit tensor is 168 MB in bf16 at 10 crops x 128):
//
//   teacher  q[r][k] = softmax((t[r][k] - center[k]) / temp_t)                    one workgroup per teacher row
//   student  logp = log_softmax(s[row] / temp_s); for row = (crop v, sample b):
//              loss_row = sum_{iq != v} sum_k -q[iq][b][k] logp[k]
//              ds[row][k] = gscale / (n_terms B temp_s) * (cnt_v p[k] - sum_{iq != v} q[iq][b][k]),  p = exp(logp)
//            (forward value AND the gradient w.r.t. the student logits in one sweep: the loss is a leaf of the
//             graph, its backward is this kernel's second output scaled by the incoming gradient)
//   colsum   batch_center[k] = sum_r t[r][k]   (update_center, loss.py:146-152; fixed order)
//
// Rows are too long for registers (K / 256 threads = 256 values): each row is swept twice -- an online (max, sum-exp)
// pass and the output pass; the second read of the 128-KB row comes from L2.  All row statistics are fp32 and the
// reductions are fixed-order (wave shuffles, then 4 wave partials in index order).
#include "vtx_common.h"

__device__ __forceinline__ void dino_combine(float& m, float& l, float m2, float l2) {
  const float M = fmaxf(m, m2);
  l = (m == -INFINITY ? 0.f : l * __expf(m - M)) + (m2 == -INFINITY ? 0.f : l2 * __expf(m2 - M));
  m = M;
}

// block-wide (max, sum-exp) of per-thread partials; result in every thread
__device__ __forceinline__ void dino_block_ml(float& m, float& l, float* red) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const float m2 = shfl_xor_f(m, o), l2 = shfl_xor_f(l, o);
    dino_combine(m, l, m2, l2);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) { red[2 * wave] = m; red[2 * wave + 1] = l; }
  __syncthreads();
  m = red[0]; l = red[1];
#pragma unroll
  for (int w = 1; w < 4; ++w) dino_combine(m, l, red[2 * w], red[2 * w + 1]);
}

template <typename T>
__global__ __launch_bounds__(256) void dino_teacher_kernel(const T* __restrict__ t, const float* __restrict__ center,
                                                          float* __restrict__ q, int K, float inv_temp) {
  __shared__ float red[8];
  const T* row = t + (int64_t)blockIdx.x * K;
  float* out = q + (int64_t)blockIdx.x * K;
  float m = -INFINITY, l = 0.f;
  for (int k0 = threadIdx.x * 8; k0 < K; k0 += 256 * 8) {
    const Vec8<T> v = load8<T>(row + k0);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float x = (v.get(e) - center[k0 + e]) * inv_temp;
      const float M = fmaxf(m, x);
      l = l * __expf(m - M) + __expf(x - M);
      m = M;
    }
  }
  dino_block_ml(m, l, red);
  const float inv = 1.f / l;
  for (int k0 = threadIdx.x * 8; k0 < K; k0 += 256 * 8) {
    const Vec8<T> v = load8<T>(row + k0);
    Vec8<float> o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o.set(e, __expf((v.get(e) - center[k0 + e]) * inv_temp - m) * inv);
    store8<float>(out + k0, o);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void dino_student_kernel(const T* __restrict__ s, const float* __restrict__ q,
                                                          T* __restrict__ ds, float* __restrict__ loss_rows, int K, int B,
                                                          float inv_temp, float gcoef) {
  __shared__ float red[8];
  const int rowi = blockIdx.x;
  const int v = rowi / B, b = rowi - v * B;
  const T* row = s + (int64_t)rowi * K;
  float m = -INFINITY, l = 0.f;
  for (int k0 = threadIdx.x * 8; k0 < K; k0 += 256 * 8) {
    const Vec8<T> x8 = load8<T>(row + k0);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float x = x8.get(e) * inv_temp;
      const float M = fmaxf(m, x);
      l = l * __expf(m - M) + __expf(x - M);
      m = M;
    }
  }
  dino_block_ml(m, l, red);
  const float lz = m + __logf(l);                    // log-sum-exp
  const float* q0 = q + (int64_t)b * K;              // teacher crop 0, sample b
  const float* q1 = q + ((int64_t)B + b) * K;        // teacher crop 1
  const bool use0 = v != 0, use1 = v != 1;           // pairs (iq, v) with iq != v  (loss.py:133-137)
  const float cnt = (use0 ? 1.f : 0.f) + (use1 ? 1.f : 0.f);
  float acc = 0.f;
  for (int k0 = threadIdx.x * 8; k0 < K; k0 += 256 * 8) {
    const Vec8<T> x8 = load8<T>(row + k0);
    Vec8<float> a = vec8_zero<float>(), c = vec8_zero<float>();
    if (use0) a = load8<float>(q0 + k0);
    if (use1) c = load8<float>(q1 + k0);
    Vec8<T> g;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float logp = x8.get(e) * inv_temp - lz;
      const float qs = a.get(e) + c.get(e);
      acc -= qs * logp;
      g.set(e, gcoef * (cnt * __expf(logp) - qs));
    }
    store8<T>(ds + (int64_t)rowi * K + k0, g);
  }
  acc = group_sum<64>(acc);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) loss_rows[rowi] = red[0] + red[1] + red[2] + red[3];
}

template <typename T>
__global__ void dino_colsum_kernel(const T* __restrict__ t, float* __restrict__ out, int rows, int K) {
  const int k0 = (blockIdx.x * blockDim.x + threadIdx.x) * 8;
  if (k0 >= K) return;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int r = 0; r < rows; ++r) {
    const Vec8<T> v = load8<T>(t + (int64_t)r * K + k0);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += v.get(e);
  }
  Vec8<float> o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o.set(e, acc[e]);
  store8<float>(out + k0, o);
}

extern "C" {

size_t vtx_dino_loss_workspace(int B, int K) { return (size_t)2 * B * K * sizeof(float); }

/* DINOLoss.forward + its gradient (reference loss.py:122-144).
 *   student [n_crop*B, K], teacher [2*B, K] (dtype), center [K] fp32 (the module's buffer, read only);
 *   workspace: vtx_dino_loss_workspace() bytes (the teacher probabilities, fp32);
 *   loss_rows [n_crop*B] fp32: loss = sum(loss_rows) / ((2 n_crop - 2) B);
 *   dstudent [n_crop*B, K] (dtype) = gscale * d loss / d student;
 *   batch_center [K] fp32 = sum over the 2B teacher rows (input of update_center, loss.py:146-152).
 * K % 8 == 0. */
int vtx_dino_loss(const void* student, const void* teacher, const float* center, void* workspace, size_t ws_bytes,
                  float* loss_rows, void* dstudent, float* batch_center, int n_crop, int B, int K, float student_temp,
                  float teacher_temp, float gscale, int dtype, void* stream) {
  if (!student || !teacher || !center || !workspace || !loss_rows || !dstudent || !batch_center) return VTX_ERR_NULL;
  if (n_crop < 2 || B <= 0 || K <= 0 || student_temp <= 0.f || teacher_temp <= 0.f) return VTX_ERR_SHAPE;
  if (K & 7) return VTX_ERR_ALIGN;
  if (ws_bytes < vtx_dino_loss_workspace(B, K)) return VTX_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  float* q = (float*)workspace;
  const float nterms = 2.f * n_crop - 2.f;
  const float gcoef = gscale / (nterms * (float)B * student_temp);
  const int cb = (K / 8 + 255) / 256;
  if (dtype == VTX_BF16) {
    hipLaunchKernelGGL((dino_teacher_kernel<bf16>), dim3(2 * B), dim3(256), 0, st, (const bf16*)teacher, center, q, K, 1.f / teacher_temp);
    hipLaunchKernelGGL((dino_student_kernel<bf16>), dim3(n_crop * B), dim3(256), 0, st, (const bf16*)student, (const float*)q, (bf16*)dstudent, loss_rows, K, B, 1.f / student_temp, gcoef);
    hipLaunchKernelGGL((dino_colsum_kernel<bf16>), dim3(cb), dim3(256), 0, st, (const bf16*)teacher, batch_center, 2 * B, K);
  } else if (dtype == VTX_F32) {
    hipLaunchKernelGGL((dino_teacher_kernel<float>), dim3(2 * B), dim3(256), 0, st, (const float*)teacher, center, q, K, 1.f / teacher_temp);
    hipLaunchKernelGGL((dino_student_kernel<float>), dim3(n_crop * B), dim3(256), 0, st, (const float*)student, (const float*)q, (float*)dstudent, loss_rows, K, B, 1.f / student_temp, gcoef);
    hipLaunchKernelGGL((dino_colsum_kernel<float>), dim3(cb), dim3(256), 0, st, (const float*)teacher, batch_center, 2 * B, K);
  } else return VTX_ERR_DTYPE;
  return vtx_check_launch();
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// MixLoss of the supervised train step (reference loss.py:53-86, train.py:274-276): label-smoothed KL divergence between
// log_softmax(logits) and the mixup / cutmix target  t = r * smooth(label1) + (1 - r) * smooth(label2),
// smooth(l)[k] = 1 - eps + eps / K at k == l, eps / K elsewhere; reduction "mean" = sum over everything / B.
//   loss_rows[b] = sum_k t_k (log t_k - logp_k)   (0 log 0 = 0)         d logits[b][k] = gscale * (p_k - t_k) / B
// One workgroup per row; value and gradient in one sweep (the loss is a leaf of the graph).
template <typename T>
__global__ __launch_bounds__(256) void mix_loss_kernel(const T* __restrict__ logits, const int64_t* __restrict__ l1,
                                                      const int64_t* __restrict__ l2, const float* __restrict__ ratio,
                                                      T* __restrict__ dlogits, float* __restrict__ loss_rows, int K,
                                                      float eps, float gcoef) {
  __shared__ float red[8];
  const int b = blockIdx.x;
  const T* row = logits + (int64_t)b * K;
  float m = -INFINITY, l = 0.f;
  for (int k = threadIdx.x; k < K; k += 256) {
    const float x = to_f32<T>(row[k]);
    const float M = fmaxf(m, x);
    l = l * __expf(m - M) + __expf(x - M);
    m = M;
  }
  dino_block_ml(m, l, red);
  const float lz = m + __logf(l);
  const int a1 = (int)l1[b], a2 = (int)l2[b];
  const float r = ratio[b];
  const float off = eps / (float)K, on = 1.f - eps + off;
  float acc = 0.f;
  for (int k = threadIdx.x; k < K; k += 256) {
    const float logp = to_f32<T>(row[k]) - lz;
    const float t = r * (k == a1 ? on : off) + (1.f - r) * (k == a2 ? on : off);
    if (t > 0.f) acc += t * (__logf(t) - logp);
    dlogits[(int64_t)b * K + k] = from_f32<T>(gcoef * (__expf(logp) - t));
  }
  acc = group_sum<64>(acc);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) loss_rows[b] = red[0] + red[1] + red[2] + red[3];
}

extern "C" {

/* logits [B, K] (dtype), label1 / label2 [B] int64, ratio [B] fp32; loss = sum(loss_rows) / B;
 * dlogits [B, K] (dtype) = gscale * d loss / d logits. */
int vtx_mix_loss(const void* logits, const int64_t* label1, const int64_t* label2, const float* ratio, void* dlogits,
                 float* loss_rows, int B, int K, float eps, float gscale, int dtype, void* stream) {
  if (!logits || !label1 || !label2 || !ratio || !dlogits || !loss_rows) return VTX_ERR_NULL;
  if (B <= 0 || K <= 0 || eps < 0.f) return VTX_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const float gcoef = gscale / (float)B;
  if (dtype == VTX_BF16)
    hipLaunchKernelGGL((mix_loss_kernel<bf16>), dim3(B), dim3(256), 0, st, (const bf16*)logits, label1, label2, ratio,
                       (bf16*)dlogits, loss_rows, K, eps, gcoef);
  else if (dtype == VTX_F32)
    hipLaunchKernelGGL((mix_loss_kernel<float>), dim3(B), dim3(256), 0, st, (const float*)logits, label1, label2, ratio,
                       (float*)dlogits, loss_rows, K, eps, gcoef);
  else return VTX_ERR_DTYPE;
  return vtx_check_launch();
}

}  // extern "C"
