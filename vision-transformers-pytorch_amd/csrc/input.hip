// Device-side input pipeline: the per-sample mixup / cutmix of reference
// mix_dataset.py:36-90, the tensor normalisation and the constant-mode RandomErasing of reference
// transforms.py:321-418, applied to a whole batch resident on the GPU in ONE pass:
//
//   v = mode 1 (mixup):  ratio * a + (1 - ratio) * b          a = image[n], b = image[partner[n]]
//       mode 2 (cutmix): b inside the box [y1, y2) x [x1, x2), a outside
//       mode 0:          a
//   v = (v - mean[c]) / std[c]
//   v = 0 inside any of the sample's erase rectangles (RandomErasing mode "const", applied after normalisation)
//
// All random decisions (partner, mode, ratio, box, rectangles) are drawn on the host in the reference's own order
// (vtx.input_pipeline.plan_batch) and arrive as a small per-sample plan table; the kernel is pure data movement:
// uint8 or fp32 NCHW in, fp32 NCHW out (what the models' patch gathers read), 16-byte stores.
#include "vtx_common.h"

#define IN_MAX_RECT 4

struct MixPlan {            // one per sample, 80 bytes
  int partner, mode;        // mode 0 none | 1 mixup | 2 cutmix
  float ratio;              // mixup weight of the sample itself
  int x1, y1, x2, y2;       // cutmix box: rows [y1, y2), columns [x1, x2)
  int nrect;
  int top[IN_MAX_RECT], left[IN_MAX_RECT];
  short eh[IN_MAX_RECT], ew[IN_MAX_RECT];
};

template <typename TI> __device__ __forceinline__ float in_px(const TI* p, float u8scale);
template <> __device__ __forceinline__ float in_px<float>(const float* p, float) { return *p; }
template <> __device__ __forceinline__ float in_px<uint8_t>(const uint8_t* p, float u8scale) { return (float)(*p) * u8scale; }

template <typename TI>
__global__ void mix_normalize_erase_kernel(const TI* __restrict__ x, const MixPlan* __restrict__ plan,
                                           const float* __restrict__ mean, const float* __restrict__ stdv,
                                           float* __restrict__ out, int C, int H, int W, float u8scale, int64_t total4) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // one thread per 4 pixels of a row
  if (idx >= total4) return;
  const int w4 = W >> 2;
  const int xq = (int)(idx % w4);
  int64_t t = idx / w4;
  const int y = (int)(t % H); t /= H;
  const int c = (int)(t % C);
  const int n = (int)(t / C);
  const MixPlan pl = plan[n];
  const int64_t plane = (int64_t)H * W;
  const TI* a = x + ((int64_t)n * C + c) * plane + (int64_t)y * W + xq * 4;
  const TI* b = x + ((int64_t)pl.partner * C + c) * plane + (int64_t)y * W + xq * 4;
  const float m = mean[c], inv = 1.f / stdv[c];
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int xx = xq * 4 + e;
    float v = in_px<TI>(a + e, u8scale);
    if (pl.mode == 1) v = v * pl.ratio + in_px<TI>(b + e, u8scale) * (1.f - pl.ratio);
    else if (pl.mode == 2 && y >= pl.y1 && y < pl.y2 && xx >= pl.x1 && xx < pl.x2) v = in_px<TI>(b + e, u8scale);
    v = (v - m) * inv;
    for (int r = 0; r < pl.nrect; ++r)
      if (y >= pl.top[r] && y < pl.top[r] + pl.eh[r] && xx >= pl.left[r] && xx < pl.left[r] + pl.ew[r]) v = 0.f;
    o[e] = v;
  }
  *reinterpret_cast<f32x4*>(out + ((int64_t)n * C + c) * plane + (int64_t)y * W + xq * 4) = o;
}

extern "C" {

size_t vtx_mix_plan_bytes(void) { return sizeof(MixPlan); }
int vtx_mix_max_rects(void) { return IN_MAX_RECT; }

/* x: [N, C, H, W] uint8 (in_u8 != 0; scaled by 1/255 like ToTensor) or fp32; plan: device array of N records
 * {int partner, mode; float ratio; int x1, y1, x2, y2, nrect, top[4], left[4]; short eh[4], ew[4]}
 * (vtx_mix_plan_bytes() each); mean / std: [C] fp32; out: [N, C, H, W] fp32 (must not alias x).  W % 4 == 0. */
int vtx_mix_normalize_erase(const void* x, int in_u8, const void* plan, const float* mean, const float* stdv, float* out,
                            int N, int C, int H, int W, void* stream) {
  if (!x || !plan || !mean || !stdv || !out) return VTX_ERR_NULL;
  if (N <= 0 || C <= 0 || H <= 0 || W <= 0) return VTX_ERR_SHAPE;
  if (W & 3) return VTX_ERR_ALIGN;
  const int64_t total4 = (int64_t)N * C * H * (W >> 2);
  dim3 grid((unsigned)((total4 + 255) / 256));
  hipStream_t st = (hipStream_t)stream;
  if (in_u8)
    hipLaunchKernelGGL((mix_normalize_erase_kernel<uint8_t>), grid, dim3(256), 0, st, (const uint8_t*)x, (const MixPlan*)plan,
                       mean, stdv, out, C, H, W, 1.f / 255.f, total4);
  else
    hipLaunchKernelGGL((mix_normalize_erase_kernel<float>), grid, dim3(256), 0, st, (const float*)x, (const MixPlan*)plan,
                       mean, stdv, out, C, H, W, 1.f, total4);
  return vtx_check_launch();
}

}  // extern "C"
