// Twins-SVT helpers for gfx950.  (1) Positional-encoding generator (reference models/twins.py:25-37):
//   y = x + DepthwiseConv3x3(x)      x, y [B, H, W, C] channels-last, weight [C, 1, 3, 3] fp32, padding 1, no bias.
// The reference permutes to NCHW, convolves and permutes back (three passes over the tensor); here the nine taps are read
// in place from the channels-last tensor (one HBM read + one write: the neighbours of a pixel come out of L2) with the
// residual add folded in.  HBM-bound; fp32 accumulation, one rounding at the store.
//   * dwconv3_kernel<T, FLIP>: forward (FLIP = false) and the input gradient (FLIP = true: the adjoint correlation, the
//     taps mirrored -- dx = dy + sum_k w[c][8 - k] dy[pixel + offset(k)]).
//   * dwconv3_wgrad_kernel: dw[c][k] = sum over pixels of dy[pixel][c] * x[pixel + offset(k)][c]: per-workgroup partial
//     sums over a fixed run of pixels, added in fixed order (LDS, then a second kernel over the workgroups): deterministic.
// (2) twins_subsample_kernel: the operand gather of the global attention's sub-sampling convolution (further down).
#include "vtx_common.h"

namespace {

constexpr int DW_THREADS = 256;

// One thread per (pixel, 8-channel vector); the weights of the workgroup's channel range sit in LDS as [tap][channel].
template <typename T, bool FLIP>
__global__ __launch_bounds__(DW_THREADS) void dwconv3_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                            T* __restrict__ y, int H, int W, int C, int64_t total) {
  extern __shared__ __attribute__((aligned(16))) float dw_ws[];            // [9][C]
  for (int i = threadIdx.x; i < 9 * C; i += DW_THREADS) {
    const int c = i / 9, k = i - 9 * c;
    dw_ws[(FLIP ? 8 - k : k) * C + c] = w[i];
  }
  __syncthreads();
  const int64_t idx = (int64_t)blockIdx.x * DW_THREADS + threadIdx.x;
  if (idx >= total) return;
  const int cv = C >> 3;
  const int v = (int)(idx % cv);
  int64_t t = idx / cv;
  const int px = (int)(t % W); t /= W;
  const int py = (int)(t % H);
  const int64_t b = t / H;
  const T* xb = x + b * (int64_t)H * W * C + v * 8;
  float acc[8];
  {
    Vec8<T> c0 = load8<T>(xb + ((int64_t)py * W + px) * C);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = c0.get(e);                         // the residual term
  }
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int yy = py + ky - 1;
    if (yy < 0 || yy >= H) continue;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int xx = px + kx - 1;
      if (xx < 0 || xx >= W) continue;
      Vec8<T> a = load8<T>(xb + ((int64_t)yy * W + xx) * C);
      const float* wk = dw_ws + (ky * 3 + kx) * C + v * 8;
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(wk), w1 = *reinterpret_cast<const f32x4*>(wk + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { acc[e] += w0[e] * a.get(e); acc[4 + e] += w1[e] * a.get(4 + e); }
    }
  }
  Vec8<T> o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o.set(e, acc[e]);
  store8<T>(y + idx * 8, o);
}

// Workgroup g owns the pixels [g * run, (g + 1) * run) of the flattened (b, y, x) range; thread = (pixel lane pl, vector v);
// 72 fp32 accumulators per thread (9 taps x 8 channels), pixel lanes added in lane order through LDS.
// part [gridDim.x][9][C] fp32.
template <typename T>
__global__ __launch_bounds__(DW_THREADS) void dwconv3_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                                  float* __restrict__ part, int H, int W, int C,
                                                                  int64_t npix, int run) {
  extern __shared__ __attribute__((aligned(16))) float dw_red[];           // [lanes][9][C]
  const int cv = C >> 3, lanes = DW_THREADS / cv;
  const int v = threadIdx.x % cv, pl = threadIdx.x / cv;
  float acc[9][8];
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[k][e] = 0.f;
  const int64_t p0 = (int64_t)blockIdx.x * run;
  const int64_t p1 = p0 + run < npix ? p0 + run : npix;
  if (pl < lanes)
    for (int64_t p = p0 + pl; p < p1; p += lanes) {
      const int px = (int)(p % W);
      const int py = (int)((p / W) % H);
      const int64_t b = p / ((int64_t)W * H);
      Vec8<T> g = load8<T>(dy + p * C + v * 8);
      const T* xb = x + b * (int64_t)H * W * C + v * 8;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int yy = py + ky - 1;
        if (yy < 0 || yy >= H) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int xx = px + kx - 1;
          if (xx < 0 || xx >= W) continue;
          Vec8<T> a = load8<T>(xb + ((int64_t)yy * W + xx) * C);
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[ky * 3 + kx][e] += g.get(e) * a.get(e);
        }
      }
    }
  if (pl < lanes) {
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
      for (int e = 0; e < 8; ++e) dw_red[((int64_t)pl * 9 + k) * C + v * 8 + e] = acc[k][e];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 9 * C; i += DW_THREADS) {
    float s = 0.f;
    for (int l = 0; l < lanes; ++l) s += dw_red[(int64_t)l * 9 * C + i];
    part[(int64_t)blockIdx.x * 9 * C + i] = s;
  }
}

// dw[c][k] = sum over workgroups (fixed order) of part[g][k][c]
__global__ void dwconv3_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int C, int ng) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;                     // over [9][C]
  if (i >= 9 * C) return;
  float s = 0.f;
  for (int g = 0; g < ng; ++g) s += part[(int64_t)g * 9 * C + i];
  const int k = i / C, c = i - k * C;
  dw[c * 9 + k] = s;
}

// ---------------------------------------------------------------------------------------------
// Operand of the sub-sampling convolution of Twins-SVT's global attention, exactly as the reference builds it
// (models/twins.py:69-70): the module input is 4-D (B, H, W, C), so ``input.transpose(1, 2).reshape(B, C, H, W)`` is NOT
// the NCHW view of the feature map -- it reinterprets the (B, W, H, C)-ordered elements as an NCHW "image" Z:
//   Z[b][c'][y][x] = Tflat[b][f],  f = c' H W + y W + x,   Tflat = the transposed map in (w, h, c) order:
//   w = f / (H C), h = (f / C) % H, c = f % C,   value = X[b][h][w][c].
// Conv2d(C, C, r, stride r) on Z = a GEMM on the patch matrix P[b Lk + i (W/r) + j][(py r + px) C + c'] =
// Z[b][c'][i r + py][j r + px] (column order (py, px, c') like patchify_kernel; the host permutes the weight to match).
// A permutation of the elements of X: forward gathers it, backward scatters the patch gradient back (optionally adding).
// One thread per element, indexed in PATCH order (coalesced on the patch side; the X side is a 2-byte gather through L2).
template <typename T, bool BWD, bool ACC>
__global__ void twins_subsample_kernel(const T* __restrict__ src, T* __restrict__ dst, int H, int W, int C, int r,
                                       int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int K = r * r * C, Wr = W / r, Lk = (H / r) * Wr;
  const int col = (int)(idx % K);
  const int64_t row = idx / K;
  const int t = (int)(row % Lk);
  const int64_t b = row / Lk;
  const int cp = col % C, pp = col / C, py = pp / r, px = pp - py * r;
  const int i = t / Wr, j = t - i * Wr;
  const int64_t f = (int64_t)cp * H * W + (int64_t)(i * r + py) * W + (j * r + px);
  const int w = (int)(f / ((int64_t)H * C));
  const int rem = (int)(f - (int64_t)w * H * C);
  const int h = rem / C, c = rem - h * C;
  const int64_t xi = ((b * H + h) * W + w) * C + c;
  if (!BWD) {
    dst[idx] = src[xi];
  } else if (ACC) {
    dst[xi] = from_f32<T>(to_f32<T>(dst[xi]) + to_f32<T>(src[idx]));
  } else {
    dst[xi] = src[idx];
  }
}

int dw_groups(int64_t npix) {                      // workgroups of the weight-gradient pass: ~4 per CU, >= 64 pixels each
  int64_t g = (npix + 63) / 64;
  if (g > 1024) g = 1024;
  return (int)(g < 1 ? 1 : g);
}
bool dw_shape_ok(int B, int H, int W, int C) { return B > 0 && H > 0 && W > 0 && C >= 8 && C <= 1024 && (C & 7) == 0; }

}  // namespace

extern "C" {

/* y = x + depthwise 3x3 convolution of x (padding 1, no bias) on channels-last [B, H, W, C] features: the positional-encoding
 * generator of Twins-SVT (reference models/twins.py:25-37).  w = the Conv2d weight [C, 1, 3, 3] fp32.  adjoint != 0 applies the
 * mirrored taps: with x := dy this is the input gradient of the same module. */
int vtx_dwconv3_fwd(const void* x, const float* w, void* y, int B, int H, int W, int C, int adjoint, int dtype, void* stream) {
  if (!x || !w || !y) return VTX_ERR_NULL;
  if (!dw_shape_ok(B, H, W, C)) return VTX_ERR_SHAPE;
  const int64_t total = (int64_t)B * H * W * (C >> 3);
  const int blocks = (int)((total + DW_THREADS - 1) / DW_THREADS);
  const size_t smem = (size_t)9 * C * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VTX_BF16) {
    if (adjoint) hipLaunchKernelGGL((dwconv3_kernel<bf16, true>), dim3(blocks), dim3(DW_THREADS), smem, st, (const bf16*)x, w, (bf16*)y, H, W, C, total);
    else hipLaunchKernelGGL((dwconv3_kernel<bf16, false>), dim3(blocks), dim3(DW_THREADS), smem, st, (const bf16*)x, w, (bf16*)y, H, W, C, total);
  } else if (dtype == VTX_F32) {
    if (adjoint) hipLaunchKernelGGL((dwconv3_kernel<float, true>), dim3(blocks), dim3(DW_THREADS), smem, st, (const float*)x, w, (float*)y, H, W, C, total);
    else hipLaunchKernelGGL((dwconv3_kernel<float, false>), dim3(blocks), dim3(DW_THREADS), smem, st, (const float*)x, w, (float*)y, H, W, C, total);
  } else return VTX_ERR_DTYPE;
  return vtx_check_launch();
}

size_t vtx_dwconv3_wgrad_workspace(int B, int H, int W, int C) {
  if (!dw_shape_ok(B, H, W, C)) return 0;
  return (size_t)dw_groups((int64_t)B * H * W) * 9 * C * sizeof(float);
}

/* dw [C, 1, 3, 3] fp32 (written, not accumulated) from the module input x and the output gradient dy; deterministic. */
int vtx_dwconv3_wgrad(const void* x, const void* dy, float* dw, void* workspace, size_t ws_bytes, int B, int H, int W, int C,
                      int dtype, void* stream) {
  if (!x || !dy || !dw || !workspace) return VTX_ERR_NULL;
  if (!dw_shape_ok(B, H, W, C)) return VTX_ERR_SHAPE;
  if (ws_bytes < vtx_dwconv3_wgrad_workspace(B, H, W, C)) return VTX_ERR_WORKSPACE;
  const int64_t npix = (int64_t)B * H * W;
  const int ng = dw_groups(npix);
  const int run = (int)((npix + ng - 1) / ng);
  const int lanes = DW_THREADS / (C >> 3);                     // C <= 1024: at least two pixel lanes
  const size_t smem = (size_t)lanes * 9 * C * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)workspace;
  if (dtype == VTX_BF16) {
    auto kern = dwconv3_wgrad_kernel<bf16>;
    if (smem > 64 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return VTX_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(ng), dim3(DW_THREADS), smem, st, (const bf16*)x, (const bf16*)dy, part, H, W, C, npix, run);
  } else if (dtype == VTX_F32) {
    auto kern = dwconv3_wgrad_kernel<float>;
    if (smem > 64 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return VTX_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(ng), dim3(DW_THREADS), smem, st, (const float*)x, (const float*)dy, part, H, W, C, npix, run);
  } else return VTX_ERR_DTYPE;
  int rc = vtx_check_launch();
  if (rc) return rc;
  hipLaunchKernelGGL(dwconv3_wgrad_reduce_kernel, dim3((9 * C + 255) / 256), dim3(256), 0, st, (const float*)part, dw, C, ng);
  return vtx_check_launch();
}

/* Patch matrix of the sub-sampling convolution of Twins-SVT's global attention (reference models/twins.py:69-71), with the
 * reference's reshape of the 4-D input kept as written (see csrc/twins_misc.hip): x [B, H, W, C] -> out [B*(H/r)*(W/r), r*r*C],
 * column order (py, px, c').  _bwd scatters the patch gradient back into dx [B, H, W, C] (a permutation; accumulate != 0 adds
 * to what dx holds -- the gradient that also arrives through the query projection). */
int vtx_twins_subsample_fwd(const void* x, void* out, int B, int H, int W, int C, int r, int dtype, void* stream) {
  if (!x || !out) return VTX_ERR_NULL;
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || r <= 0 || H % r || W % r) return VTX_ERR_SHAPE;
  const int64_t total = (int64_t)B * H * W * C;
  const int blocks = (int)((total + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VTX_BF16) hipLaunchKernelGGL((twins_subsample_kernel<bf16, false, false>), dim3(blocks), dim3(256), 0, st, (const bf16*)x, (bf16*)out, H, W, C, r, total);
  else if (dtype == VTX_F32) hipLaunchKernelGGL((twins_subsample_kernel<float, false, false>), dim3(blocks), dim3(256), 0, st, (const float*)x, (float*)out, H, W, C, r, total);
  else return VTX_ERR_DTYPE;
  return vtx_check_launch();
}

int vtx_twins_subsample_bwd(const void* dout, void* dx, int B, int H, int W, int C, int r, int accumulate, int dtype,
                            void* stream) {
  if (!dout || !dx) return VTX_ERR_NULL;
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || r <= 0 || H % r || W % r) return VTX_ERR_SHAPE;
  const int64_t total = (int64_t)B * H * W * C;
  const int blocks = (int)((total + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VTX_BF16) {
    if (accumulate) hipLaunchKernelGGL((twins_subsample_kernel<bf16, true, true>), dim3(blocks), dim3(256), 0, st, (const bf16*)dout, (bf16*)dx, H, W, C, r, total);
    else hipLaunchKernelGGL((twins_subsample_kernel<bf16, true, false>), dim3(blocks), dim3(256), 0, st, (const bf16*)dout, (bf16*)dx, H, W, C, r, total);
  } else if (dtype == VTX_F32) {
    if (accumulate) hipLaunchKernelGGL((twins_subsample_kernel<float, true, true>), dim3(blocks), dim3(256), 0, st, (const float*)dout, (float*)dx, H, W, C, r, total);
    else hipLaunchKernelGGL((twins_subsample_kernel<float, true, false>), dim3(blocks), dim3(256), 0, st, (const float*)dout, (float*)dx, H, W, C, r, total);
  } else return VTX_ERR_DTYPE;
  return vtx_check_launch();
}

}  // extern "C"
