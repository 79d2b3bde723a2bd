// Data-movement kernels around the GEMMs (all HBM-bound): patch gathers of the two patch
// embeddings, token mean-pool of the Swin classifier, cls/pos assembly of ViT.
#include "vtx_common.h"

// ---------------------------------------------------------------------------------------------
// Patch gather from an NCHW fp32 image into a row-major patch matrix [B*gh*gw, Kp] of T.
//   order 0 (Swin, reference models/swin_transformer.py:15-22,208-213,371: permute to NHWC then
//            patchify): column = (py*p + px)*Cin + c
//   order 1 (ViT,  reference models/vit.py:73,76: Conv2d(k=p, s=p) == GEMM over the conv weight's own
//            flattening): column = (c*p + py)*p + px
// Columns K..Kp-1 are zero (Swin: K = 48 padded to the GEMM's 8-element vector granularity / k-tile).
// One block per (image, patch row): the Cin*p contiguous image rows are loaded coalesced
// (float4) into LDS once, then written out as full 16-byte vectors of T.
template <typename T>
__global__ __launch_bounds__(256) void patch_gather_kernel(const float* __restrict__ x, T* __restrict__ out, int Cin,
                                                          int H, int W, int p, int K, int Kp, int order) {
  extern __shared__ __attribute__((aligned(16))) float pg_smem[];   // [Cin][p][W]
  const int gh = H / p, gw = W / p;
  const int b = blockIdx.x / gh, i = blockIdx.x % gh;
  const int rowlen = W;                     // floats per image row
  const int nrows = Cin * p;
  const int vec_per_row = rowlen >> 2;
  for (int idx = threadIdx.x; idx < nrows * vec_per_row; idx += blockDim.x) {
    const int r = idx / vec_per_row, v = idx - r * vec_per_row;
    const int c = r / p, py = r - c * p;
    const float* src = x + (((int64_t)b * Cin + c) * H + (int64_t)i * p + py) * W + v * 4;
    *reinterpret_cast<f32x4*>(pg_smem + r * rowlen + v * 4) = *reinterpret_cast<const f32x4*>(src);
  }
  __syncthreads();
  const int kvec = Kp >> 3;
  T* obase = out + ((int64_t)b * gh + i) * gw * (int64_t)Kp;
  for (int idx = threadIdx.x; idx < gw * kvec; idx += blockDim.x) {
    const int j = idx / kvec, kv = idx - j * kvec;
    Vec8<T> o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int col = kv * 8 + e;
      float val = 0.f;
      if (col < K) {
        int c, py, px;
        if (order == 0) { c = col % Cin; const int t = col / Cin; px = t % p; py = t / p; }
        else { px = col % p; const int t = col / p; py = t % p; c = t / p; }
        val = pg_smem[(c * p + py) * rowlen + j * p + px];
      }
      o.set(e, val);
    }
    store8<T>(obase + (int64_t)j * Kp + kv * 8, o);
  }
}

// Same gather from a bf16 NHWC image [B, H, W, Cin] (the device input pipeline's output, csrc/input.hip): the p image
// rows of a patch row are p contiguous runs of W * Cin elements -- coalesced 16-byte loads into LDS, 16-byte stores.
template <typename T>
__global__ __launch_bounds__(256) void patch_gather_nhwc_kernel(const bf16* __restrict__ x, T* __restrict__ out, int Cin,
                                                               int H, int W, int p, int K, int Kp, int order) {
  extern __shared__ __attribute__((aligned(16))) unsigned char pgn_smem_raw[];   // [p][W][Cin] bf16
  bf16* sm = reinterpret_cast<bf16*>(pgn_smem_raw);
  const int gh = H / p, gw = W / p;
  const int b = blockIdx.x / gh, i = blockIdx.x % gh;
  const int rowlen = W * Cin;                                  // elements per image row (a multiple of 8: W % 8 == 0)
  const bf16* src = x + ((int64_t)b * H + (int64_t)i * p) * rowlen;   // p consecutive rows = one contiguous run
  for (int v = threadIdx.x; v < (p * rowlen) >> 3; v += blockDim.x)
    *reinterpret_cast<bf16x8*>(sm + v * 8) = *reinterpret_cast<const bf16x8*>(src + (int64_t)v * 8);
  __syncthreads();
  const int kvec = Kp >> 3;
  T* obase = out + ((int64_t)b * gh + i) * gw * (int64_t)Kp;
  for (int idx = threadIdx.x; idx < gw * kvec; idx += blockDim.x) {
    const int j = idx / kvec, kv = idx - j * kvec;
    Vec8<T> o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int col = kv * 8 + e;
      float val = 0.f;
      if (col < K) {
        int c, py, px;
        if (order == 0) { c = col % Cin; const int t = col / Cin; px = t % p; py = t / p; }
        else { px = col % p; const int t = col / p; py = t % p; c = t / p; }
        val = (float)sm[(py * W + j * p + px) * Cin + c];
      }
      o.set(e, val);
    }
    store8<T>(obase + (int64_t)j * Kp + kv * 8, o);
  }
}

// ---------------------------------------------------------------------------------------------
// Token mean: y[b][c] = mean_t x[b][t][c]   (AdaptiveAvgPool2d(1)+Flatten, swin_transformer.py:281)
template <typename T>
__global__ void token_mean_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int Tn, int C) {
  const int b = blockIdx.y;
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= (C >> 3)) return;
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const T* p = x + ((int64_t)b * Tn) * C + v * 8;
  for (int t = 0; t < Tn; ++t) {
    Vec8<T> a = load8<T>(p + (int64_t)t * C);
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] += a.get(e);
  }
  Vec8<T> o;
  const float inv = 1.f / (float)Tn;
#pragma unroll
  for (int e = 0; e < 8; ++e) o.set(e, s[e] * inv);
  store8<T>(y + (int64_t)b * C + v * 8, o);
}

template <typename T>
__global__ void token_mean_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int Tn, int C) {
  const int b = blockIdx.y;
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= (C >> 3)) return;
  Vec8<T> a = load8<T>(dy + (int64_t)b * C + v * 8), o;
  const float inv = 1.f / (float)Tn;
#pragma unroll
  for (int e = 0; e < 8; ++e) o.set(e, a.get(e) * inv);
  T* p = dx + ((int64_t)b * Tn) * C + v * 8;
  for (int t = 0; t < Tn; ++t) store8<T>(p + (int64_t)t * C, o);
}

// ---------------------------------------------------------------------------------------------
// ViT token assembly (vit.py:140-143): out[b][0] = cls + pos[0]; out[b][1+t] = patches[b][t] + pos[1+t]
template <typename T>
__global__ void vit_assemble_fwd_kernel(const T* __restrict__ patches, const float* __restrict__ cls,
                                        const float* __restrict__ pos, T* __restrict__ out, int B, int L, int C) {
  const int cv = C >> 3;
  const int64_t total = (int64_t)B * L * cv;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int v = (int)(idx % cv);
    const int64_t bt = idx / cv;
    const int t = (int)(bt % L);
    const int64_t b = bt / L;
    Vec8<T> o;
    if (t == 0) {
#pragma unroll
      for (int e = 0; e < 8; ++e) o.set(e, cls[v * 8 + e] + pos[v * 8 + e]);
    } else {
      Vec8<T> a = load8<T>(patches + ((b * (L - 1)) + (t - 1)) * C + v * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) o.set(e, a.get(e) + pos[(int64_t)t * C + v * 8 + e]);
    }
    store8<T>(out + bt * C + v * 8, o);
  }
}

// backward: dpatches[b][t] = dx[b][1+t]; dpos[t] = sum_b dx[b][t]; dcls = sum_b dx[b][0]  (fixed order).
// 256 threads = 16 column vectors x 16 batch lanes: lane j sums images j, j+16, ... (and copies their patch rows), the 16
// partial sums are added in lane order through LDS -- one thread per column vector walking the whole batch left the
// chip with ~150 waves and a 256-deep serial load chain (128 us for ViT-S/16 at B = 256; 38.7 MB = 7 us of traffic).
template <typename T>
__global__ __launch_bounds__(256) void vit_assemble_bwd_kernel(const T* __restrict__ dx, T* __restrict__ dpatches,
                                                              float* __restrict__ dcls, float* __restrict__ dpos, int B,
                                                              int L, int C) {
  __shared__ float red[16][16][9];
  const int cv = C >> 3;
  const int vl = threadIdx.x & 15, bl = threadIdx.x >> 4;
  const int idx = blockIdx.x * 16 + vl;
  const bool live = idx < L * cv;
  const int t = live ? idx / cv : 0, v = live ? idx - t * cv : 0;
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (live)
    for (int b = bl; b < B; b += 16) {
      Vec8<T> a = load8<T>(dx + ((int64_t)b * L + t) * C + v * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += a.get(e);
      if (t > 0) store8<T>(dpatches + ((int64_t)b * (L - 1) + (t - 1)) * C + v * 8, a);
    }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[bl][vl][e] = s[e];
  __syncthreads();
  if (bl == 0 && live) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float acc = 0.f;
      for (int j = 0; j < 16; ++j) acc += red[j][vl][e];
      dpos[(int64_t)t * C + v * 8 + e] = acc;
      if (t == 0) dcls[v * 8 + e] = acc;
    }
  }
}

#define MISC_BY_DTYPE(CALL_BF16, CALL_F32)            \
  do {                                                \
    if (dtype == VTX_BF16) { CALL_BF16; }             \
    else if (dtype == VTX_F32) { CALL_F32; }          \
    else return VTX_ERR_DTYPE;                        \
    return vtx_check_launch();                        \
  } while (0)

extern "C" {

int vtx_patch_gather(const float* x, void* out, int B, int Cin, int H, int W, int p, int Kp, int order, int dtype,
                     void* stream) {
  if (!x || !out) return VTX_ERR_NULL;
  const int K = Cin * p * p;
  if (p <= 0 || H % p || W % p || (W & 3) || (Kp & 7) || Kp < K) return VTX_ERR_SHAPE;
  const size_t smem = (size_t)Cin * p * W * sizeof(float);
  if (smem > 160 * 1024) return VTX_ERR_SHAPE;
  if (smem > 64 * 1024) {                                 // e.g. ViT-S/16 at 384 x 384: 3 x 16 rows x 384 floats = 72 KB
    if (hipFuncSetAttribute((const void*)patch_gather_kernel<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess ||
        hipFuncSetAttribute((const void*)patch_gather_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return VTX_ERR_LAUNCH;
  }
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(B * (H / p));
  MISC_BY_DTYPE(
      hipLaunchKernelGGL((patch_gather_kernel<bf16>), grid, dim3(256), smem, st, x, (bf16*)out, Cin, H, W, p, K, Kp, order),
      hipLaunchKernelGGL((patch_gather_kernel<float>), grid, dim3(256), smem, st, x, (float*)out, Cin, H, W, p, K, Kp, order));
}

int vtx_patch_gather_nhwc(const void* x, void* out, int B, int Cin, int H, int W, int p, int Kp, int order, int dtype,
                          void* stream) {
  if (!x || !out) return VTX_ERR_NULL;
  const int K = Cin * p * p;
  if (p <= 0 || H % p || W % p || (W & 7) || (Kp & 7) || Kp < K) return VTX_ERR_SHAPE;
  const size_t smem = (size_t)Cin * p * W * sizeof(bf16);
  if (smem > 64 * 1024) return VTX_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(B * (H / p));
  MISC_BY_DTYPE(
      hipLaunchKernelGGL((patch_gather_nhwc_kernel<bf16>), grid, dim3(256), smem, st, (const bf16*)x, (bf16*)out, Cin, H, W, p, K, Kp, order),
      hipLaunchKernelGGL((patch_gather_nhwc_kernel<float>), grid, dim3(256), smem, st, (const bf16*)x, (float*)out, Cin, H, W, p, K, Kp, order));
}

int vtx_token_mean_fwd(const void* x, void* y, int B, int Tn, int C, int dtype, void* stream) {
  if (!x || !y) return VTX_ERR_NULL;
  if (C & 7) return VTX_ERR_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(((C >> 3) + 63) / 64, B);
  MISC_BY_DTYPE(hipLaunchKernelGGL((token_mean_fwd_kernel<bf16>), grid, dim3(64), 0, st, (const bf16*)x, (bf16*)y, Tn, C),
                hipLaunchKernelGGL((token_mean_fwd_kernel<float>), grid, dim3(64), 0, st, (const float*)x, (float*)y, Tn, C));
}

int vtx_token_mean_bwd(const void* dy, void* dx, int B, int Tn, int C, int dtype, void* stream) {
  if (!dy || !dx) return VTX_ERR_NULL;
  if (C & 7) return VTX_ERR_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(((C >> 3) + 63) / 64, B);
  MISC_BY_DTYPE(hipLaunchKernelGGL((token_mean_bwd_kernel<bf16>), grid, dim3(64), 0, st, (const bf16*)dy, (bf16*)dx, Tn, C),
                hipLaunchKernelGGL((token_mean_bwd_kernel<float>), grid, dim3(64), 0, st, (const float*)dy, (float*)dx, Tn, C));
}

int vtx_vit_assemble_fwd(const void* patches, const float* cls, const float* pos, void* out, int B, int L, int C,
                         int dtype, void* stream) {
  if (!patches || !cls || !pos || !out) return VTX_ERR_NULL;
  if (C & 7) return VTX_ERR_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  int64_t total = (int64_t)B * L * (C >> 3);
  int nb = (int)((total + 255) / 256);
  if (nb > 4096) nb = 4096;
  MISC_BY_DTYPE(hipLaunchKernelGGL((vit_assemble_fwd_kernel<bf16>), dim3(nb), dim3(256), 0, st, (const bf16*)patches, cls, pos, (bf16*)out, B, L, C),
                hipLaunchKernelGGL((vit_assemble_fwd_kernel<float>), dim3(nb), dim3(256), 0, st, (const float*)patches, cls, pos, (float*)out, B, L, C));
}

int vtx_vit_assemble_bwd(const void* dx, void* dpatches, float* dcls, float* dpos, int B, int L, int C, int dtype,
                         void* stream) {
  if (!dx || !dpatches || !dcls || !dpos) return VTX_ERR_NULL;
  if (C & 7) return VTX_ERR_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  const int n = L * (C >> 3);
  dim3 grid((n + 15) / 16);
  MISC_BY_DTYPE(hipLaunchKernelGGL((vit_assemble_bwd_kernel<bf16>), grid, dim3(256), 0, st, (const bf16*)dx, (bf16*)dpatches, dcls, dpos, B, L, C),
                hipLaunchKernelGGL((vit_assemble_bwd_kernel<float>), grid, dim3(256), 0, st, (const float*)dx, (float*)dpatches, dcls, dpos, B, L, C));
}

}  // extern "C"
