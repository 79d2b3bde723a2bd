// Device-side input pipeline: the per-sample mixup / cutmix of reference
// mix_dataset.py:36-90, the tensor normalisation and the constant-mode RandomErasing of reference
// transforms.py:321-418, applied to a whole batch resident on the GPU in ONE pass:
//
//   v = mode 1 (mixup):  ratio * a + (1 - ratio) * b          a = image[n], b = image[partner[n]]
//       mode 2 (cutmix): b inside the box [y1, y2) x [x1, x2), a outside
//       mode 0:          a
//   v = (v - mean[c]) / std[c]
//   v = erase colour inside any of the sample's erase rectangles (RandomErasing after normalisation): 0 (mode "const"),
//       a per-channel normal draw ("rand") or a per-pixel one ("pixel" -- what factory.py:177-181 configures); the draws
//       are made on the HOST with torch's generator in the reference's order and shipped as a flat fp32 table
//
// All random decisions (partner, mode, ratio, box, rectangles) are drawn on the host in the reference's own order
// (vtx.input_pipeline.plan_batch) and arrive as a small per-sample plan table; the kernel is pure data movement:
// uint8 or fp32 NCHW in; fp32 NCHW out (the reference's model input contract), or bf16 NHWC out -- the layout the
// patch-embedding gather consumes directly (half the bytes, a patch row = one contiguous run).
#include "vtx_common.h"

#define IN_MAX_RECT 4

struct MixPlan {            // one per sample, 100 bytes
  int partner, mode;        // mode 0 none | 1 mixup | 2 cutmix
  float ratio;              // mixup weight of the sample itself
  int x1, y1, x2, y2;       // cutmix box: rows [y1, y2), columns [x1, x2)
  int nrect;
  int top[IN_MAX_RECT], left[IN_MAX_RECT];
  short eh[IN_MAX_RECT], ew[IN_MAX_RECT];
  int fmode;                // RandomErasing colour mode: 0 'const' (zeros) | 1 'rand' (one normal draw per channel and
                            // rectangle) | 2 'pixel' (one per erased pixel) -- transforms.py:309-318, 372-378
  int foff[IN_MAX_RECT];    // float offset of rectangle r's draws in `fills`: [C] (rand) or [C][eh][ew] (pixel)
};

template <typename TI> __device__ __forceinline__ float in_px(const TI* p, float u8scale);
template <> __device__ __forceinline__ float in_px<float>(const float* p, float) { return *p; }
template <> __device__ __forceinline__ float in_px<uint8_t>(const uint8_t* p, float u8scale) { return (float)(*p) * u8scale; }

// value of output pixel (n, c, y, xx): mix -> normalise -> erase (later rectangles overwrite earlier ones, like the
// reference's sequential slice assignments)
template <typename TI>
__device__ __forceinline__ float mix_pixel(const TI* a, const TI* b, const MixPlan& pl, const float* __restrict__ fills,
                                           int c, int y, int xx, float m, float inv, float u8scale) {
  float v = in_px<TI>(a, u8scale);
  if (pl.mode == 1) v = v * pl.ratio + in_px<TI>(b, u8scale) * (1.f - pl.ratio);
  else if (pl.mode == 2 && y >= pl.y1 && y < pl.y2 && xx >= pl.x1 && xx < pl.x2) v = in_px<TI>(b, u8scale);
  v = (v - m) * inv;
  for (int r = 0; r < pl.nrect; ++r) {
    const int dy = y - pl.top[r], dx = xx - pl.left[r];
    if (dy >= 0 && dy < pl.eh[r] && dx >= 0 && dx < pl.ew[r]) {
      if (pl.fmode == 0) v = 0.f;
      else if (pl.fmode == 1) v = fills[pl.foff[r] + c];
      else v = fills[pl.foff[r] + (c * pl.eh[r] + dy) * pl.ew[r] + dx];
    }
  }
  return v;
}

// fp32 NCHW out (what the reference's model(input) receives): one thread per 4 pixels of a row of one channel
template <typename TI>
__global__ void mix_normalize_erase_kernel(const TI* __restrict__ x, const MixPlan* __restrict__ plan,
                                           const float* __restrict__ mean, const float* __restrict__ stdv,
                                           const float* __restrict__ fills, float* __restrict__ out, int C, int H, int W,
                                           float u8scale, int64_t total4) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
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
  for (int e = 0; e < 4; ++e) o[e] = mix_pixel<TI>(a + e, b + e, pl, fills, c, y, xq * 4 + e, m, inv, u8scale);
  *reinterpret_cast<f32x4*>(out + ((int64_t)n * C + c) * plane + (int64_t)y * W + xq * 4) = o;
}

// bf16 NHWC out [N, H, W, C] -- the layout the patch-embedding gather wants (vtx_patch_gather_nhwc: a patch row is one
// contiguous run of p * C elements), half the bytes of fp32 NCHW and no second pass: one thread per 4 pixels of a row,
// all C <= 4 channels; the 4 C values go out as C 8-byte stores.
template <typename TI, int CC>
__global__ void mix_normalize_erase_nhwc_kernel(const TI* __restrict__ x, const MixPlan* __restrict__ plan,
                                                const float* __restrict__ mean, const float* __restrict__ stdv,
                                                const float* __restrict__ fills, bf16* __restrict__ out, int H, int W,
                                                float u8scale, int64_t total4) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total4) return;
  const int w4 = W >> 2;
  const int xq = (int)(idx % w4);
  int64_t t = idx / w4;
  const int y = (int)(t % H);
  const int n = (int)(t / H);
  const MixPlan pl = plan[n];
  const int64_t plane = (int64_t)H * W;
  bf16 v[4 * CC];
#pragma unroll
  for (int c = 0; c < CC; ++c) {
    const TI* a = x + ((int64_t)n * CC + c) * plane + (int64_t)y * W + xq * 4;
    const TI* b = x + ((int64_t)pl.partner * CC + c) * plane + (int64_t)y * W + xq * 4;
    const float m = mean[c], inv = 1.f / stdv[c];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e * CC + c] = (bf16)mix_pixel<TI>(a + e, b + e, pl, fills, c, y, xq * 4 + e, m, inv, u8scale);
  }
  bf16* dst = out + (((int64_t)n * H + y) * W + xq * 4) * CC;       // 8 CC bytes per thread: 8-byte aligned
#pragma unroll
  for (int k = 0; k < CC; ++k) {
    bf16x4 o = {v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]};
    *reinterpret_cast<bf16x4*>(dst + 4 * k) = o;
  }
}

extern "C" {

size_t vtx_mix_plan_bytes(void) { return sizeof(MixPlan); }
int vtx_mix_max_rects(void) { return IN_MAX_RECT; }

/* x: [N, C, H, W] uint8 (in_u8 != 0; scaled by 1/255 like ToTensor) or fp32; plan: device array of N records (MixPlan
 * above, vtx_mix_plan_bytes() each); mean / std: [C] fp32; fills: the host-drawn normal values of RandomErasing's
 * 'rand' / 'pixel' modes (may be NULL when every record has fmode 0).
 * out_nhwc_bf16 == 0: out [N, C, H, W] fp32;  != 0: out [N, H, W, C] bf16 (C <= 4).  out must not alias x.  W % 4 == 0. */
int vtx_mix_normalize_erase(const void* x, int in_u8, const void* plan, const float* mean, const float* stdv,
                            const float* fills, void* out, int out_nhwc_bf16, int N, int C, int H, int W, void* stream) {
  if (!x || !plan || !mean || !stdv || !out) return VTX_ERR_NULL;
  if (N <= 0 || C <= 0 || H <= 0 || W <= 0) return VTX_ERR_SHAPE;
  if (W & 3) return VTX_ERR_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  const MixPlan* pl = (const MixPlan*)plan;
  if (out_nhwc_bf16) {
    if (C != 3 && C != 1 && C != 4) return VTX_ERR_SHAPE;
    const int64_t total4 = (int64_t)N * H * (W >> 2);
    dim3 grid((unsigned)((total4 + 255) / 256));
#define VTX_NHWC(TI, CC, SC)                                                                                          \
  hipLaunchKernelGGL((mix_normalize_erase_nhwc_kernel<TI, CC>), grid, dim3(256), 0, st, (const TI*)x, pl, mean, stdv, \
                     fills, (bf16*)out, H, W, SC, total4)
    if (in_u8) { if (C == 3) VTX_NHWC(uint8_t, 3, 1.f / 255.f); else if (C == 1) VTX_NHWC(uint8_t, 1, 1.f / 255.f); else VTX_NHWC(uint8_t, 4, 1.f / 255.f); }
    else { if (C == 3) VTX_NHWC(float, 3, 1.f); else if (C == 1) VTX_NHWC(float, 1, 1.f); else VTX_NHWC(float, 4, 1.f); }
#undef VTX_NHWC
    return vtx_check_launch();
  }
  const int64_t total4 = (int64_t)N * C * H * (W >> 2);
  dim3 grid((unsigned)((total4 + 255) / 256));
  if (in_u8)
    hipLaunchKernelGGL((mix_normalize_erase_kernel<uint8_t>), grid, dim3(256), 0, st, (const uint8_t*)x, pl, mean, stdv,
                       fills, (float*)out, C, H, W, 1.f / 255.f, total4);
  else
    hipLaunchKernelGGL((mix_normalize_erase_kernel<float>), grid, dim3(256), 0, st, (const float*)x, pl, mean, stdv,
                       fills, (float*)out, C, H, W, 1.f, total4);
  return vtx_check_launch();
}

}  // extern "C"
