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
#include "options.h"
#include "vtx_common.h"

constexpr int DW_THREADS = 256;

// One workgroup per image row (b, y): its threads walk the row's (pixel, 8-channel vector) elements, 32-bit index
// arithmetic only; the weights sit in LDS as [tap][channel].  The rows above / below come out of L2 (every input element
// is requested by three rows of workgroups).
template <typename T, bool FLIP>
__global__ __launch_bounds__(DW_THREADS) void dwconv3_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                            T* __restrict__ y, int H, int W, int C) {
  extern __shared__ __attribute__((aligned(16))) float dw_ws[];            // [9][C]
  for (int i = threadIdx.x; i < 9 * C; i += DW_THREADS) {
    const int c = i / 9, k = i - 9 * c;
    dw_ws[(FLIP ? 8 - k : k) * C + c] = w[i];
  }
  __syncthreads();
  const int row = blockIdx.x;                        // b * H + py
  const int py = row % H;
  const int cv = C >> 3, n = W * cv;
  const T* xr = x + (int64_t)row * W * C;            // this row; the neighbours are +- W * C elements away
  T* yr = y + (int64_t)row * W * C;
  const bool up = py > 0, down = py + 1 < H;
  for (int e = threadIdx.x; e < n; e += DW_THREADS) {
    const int px = e / cv, v = e - px * cv;
    const int off = px * C + v * 8;
    float acc[8];
    {
      Vec8<T> c0 = load8<T>(xr + off);
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[q] = c0.get(q);                       // the residual term
    }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      if ((ky == 0 && !up) || (ky == 2 && !down)) continue;
      const T* xk = xr + (ky - 1) * W * C + off;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int xx = px + kx - 1;
        if (xx < 0 || xx >= W) continue;
        Vec8<T> a = load8<T>(xk + (kx - 1) * C);
        const float* wk = dw_ws + (ky * 3 + kx) * C + v * 8;
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(wk), w1 = *reinterpret_cast<const f32x4*>(wk + 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) { acc[q] += w0[q] * a.get(q); acc[4 + q] += w1[q] * a.get(4 + q); }
      }
    }
    Vec8<T> o;
#pragma unroll
    for (int q = 0; q < 8; ++q) o.set(q, acc[q]);
    store8<T>(yr + off, o);
  }
}

// Workgroup g owns the image rows [g * run, (g + 1) * run) of the flattened (b, y) range; thread = (pixel lane pl, vector v)
// walks the pixels pl, pl + lanes, ... of each row; 72 fp32 accumulators per thread (9 taps x 8 channels), pixel lanes added
// in lane order through LDS.  part [gridDim.x][9][C] fp32.
template <typename T>
__global__ __launch_bounds__(DW_THREADS) void dwconv3_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                                  float* __restrict__ part, int H, int W, int C,
                                                                  int nrows, int run) {
  extern __shared__ __attribute__((aligned(16))) float dw_red[];           // [lanes][9][C]
  const int cv = C >> 3, lanes = DW_THREADS / cv;
  const int v = threadIdx.x % cv, pl = threadIdx.x / cv;
  float acc[9][8];
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[k][e] = 0.f;
  const int r0 = blockIdx.x * run;
  const int r1 = r0 + run < nrows ? r0 + run : nrows;
  if (pl < lanes)
    for (int row = r0; row < r1; ++row) {
      const int py = row % H;
      const bool up = py > 0, down = py + 1 < H;
      const T* xr = x + (int64_t)row * W * C + v * 8;
      const T* gr = dy + (int64_t)row * W * C + v * 8;
      for (int px = pl; px < W; px += lanes) {
        Vec8<T> g = load8<T>(gr + px * C);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          if ((ky == 0 && !up) || (ky == 2 && !down)) continue;
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const int xx = px + kx - 1;
            if (xx < 0 || xx >= W) continue;
            Vec8<T> a = load8<T>(xr + ((ky - 1) * W + xx) * C);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[ky * 3 + kx][e] += g.get(e) * a.get(e);
          }
        }
      }
    }
  if (pl < lanes) {
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
      for (int e = 0; e < 8; ++e) dw_red[(pl * 9 + k) * C + v * 8 + e] = acc[k][e];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 9 * C; i += DW_THREADS) {
    float s = 0.f;
    for (int l = 0; l < lanes; ++l) s += dw_red[l * 9 * C + i];
    part[(int64_t)blockIdx.x * 9 * C + i] = s;
  }
}

// dw[c][k] = sum over workgroups of part[g][k][c], in a fixed order: thread (column i, slice s) adds the groups s, s + 16, ...
// in sequence, the 16 slices are then added in slice order through LDS.  16 columns x 16 slices per workgroup.
__global__ __launch_bounds__(256) void dwconv3_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int C,
                                                                  int ng) {
  __shared__ float red[16][17];
  const int il = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + il;                                      // over [9][C]
  float s = 0.f;
  if (i < 9 * C) {
#pragma unroll 4
    for (int g = sl; g < ng; g += 16) s += part[(int64_t)g * 9 * C + i];
  }
  red[sl][il] = s;
  __syncthreads();
  if (sl == 0 && i < 9 * C) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red[q][il];
    const int k = i / C, c = i - k * C;
    dw[c * 9 + k] = t;
  }
}

// ---------------------------------------------------------------------------------------------
// Operand of the sub-sampling convolution of Twins-SVT's global attention, exactly as the reference builds it
// (models/twins.py:69-70): the module input is 4-D (B, H, W, C), so ``input.transpose(1, 2).reshape(B, C, H, W)`` is NOT
// the NCHW view of the feature map -- it reinterprets the (B, W, H, C)-ordered elements as an NCHW "image" Z:
//   Z[b][c'][y][x] = Tflat[b][f],  f = c' H W + y W + x,   Tflat = the transposed map in (w, h, c) order:
//   w = f / (H C), h = (f / C) % H, c = f % C,   value = X[b][h][w][c].
// Conv2d(C, C, r, stride r) on Z = a GEMM on the patch matrix P[b Lk + i (W/r) + j][c' r r + py r + px] =
// Z[b][c'][i r + py][j r + px] -- the column order of the Conv2d weight's own memory layout [out][c'][py][px], so the weight
// (and its gradient) need no permutation.  A permutation of the elements of X: forward gathers it, backward scatters the
// patch gradient back (optionally adding).  One thread per element in PATCH order (coalesced there; on the X side runs of r
// consecutive f = r consecutive channels, 2-byte accesses that L2 merges; a thread per run of r was 1.5-6x slower: 14-byte
// strides on both sides); grid.y = image, 32-bit arithmetic inside it.
// divisors of the index map with their reciprocals: n / d == __umulhi(n, ceil(2^32 / d)) while n * d < 2^32 (checked on the
// host; the plain-division instantiation serves anything larger) -- the six runtime divisions were the kernel's whole time
struct SubGeom { int H, W, C, r, K, rr, Wr, HC; unsigned mK, mrr, mr, mWr, mHC, mC; };
template <bool FAST> __device__ __forceinline__ int sub_div(int n, int d, unsigned magic) {
  return FAST ? (int)__umulhi((unsigned)n, magic) : n / d;
}
template <typename T, bool BWD, bool ACC, bool FAST>
__global__ void twins_subsample_kernel(const T* __restrict__ src, T* __restrict__ dst, T* __restrict__ dst_t, SubGeom g) {
  const int per_img = g.H * g.W * g.C;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;                      // element of this image's patch block
  if (e >= per_img) return;
  const int t = sub_div<FAST>(e, g.K, g.mK), col = e - t * g.K;
  const int cp = sub_div<FAST>(col, g.rr, g.mrr), pp = col - cp * g.rr;
  const int py = sub_div<FAST>(pp, g.r, g.mr), px = pp - py * g.r;
  const int i = sub_div<FAST>(t, g.Wr, g.mWr), j = t - i * g.Wr;
  const int f = cp * g.H * g.W + (i * g.r + py) * g.W + (j * g.r + px);
  const int w = sub_div<FAST>(f, g.HC, g.mHC), rem = f - w * g.HC;
  const int h = sub_div<FAST>(rem, g.C, g.mC), c = rem - h * g.C;
  const int64_t base = (int64_t)blockIdx.y * per_img;
  const int64_t xi = base + (h * g.W + w) * g.C + c, pi = base + e;
  if (!BWD) {
    const T v = src[xi];
    dst[pi] = v;
    // second copy, transposed [K][B Lk]: the operand of the split-K launch that serves the long-K / few-row reduction convs
    if (dst_t != nullptr) {
      const int Lk = per_img / g.K;
      dst_t[(int64_t)col * ((int64_t)gridDim.y * Lk) + (int64_t)blockIdx.y * Lk + t] = v;
    }
  } else if (ACC) {
    dst[xi] = from_f32<T>(to_f32<T>(dst[xi]) + to_f32<T>(src[pi]));
  } else {
    dst[xi] = src[pi];
  }
}

// ---- LDS-staged variant (round 4; VERDICT r3 #8: the element-wise kernel above reaches 0.155 of the HBM peak -- 2-byte accesses,
// 128 bytes per wave instruction).  One workgroup per (image, row of patches i): the r Z-rows it gathers are, per Z-channel c', ONE
// contiguous range of r W elements of Tflat (f = c' H W + i r W .. + r W), and Tflat's f-order visits X pixel by pixel (C contiguous
// elements each).  Phase 1 copies those ranges X -> LDS in Z layout [c'][r W] with the widest vector CH the alignments allow (16 bytes at
// 56 x 56, 8 at 28 x 28, 4 at 14 x 14: r W, H W and C must be multiples of CH); phase 2 walks the patch rows of the row block in output
// order -- 16-byte stores of 8 (bf16) / 4 (fp32) consecutive columns, their elements picked out of LDS (runs of r).  The backward is
// the same two phases the other way round (16-byte loads of the patch gradient scattered into LDS, CH-wide read-modify-write of dx).
template <typename T, int CH, bool BWD, bool ACC>
__global__ __launch_bounds__(256) void twins_subsample_lds_kernel(const T* __restrict__ src, T* __restrict__ dst, SubGeom g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sub_smem[];
  T* z = reinterpret_cast<T*>(sub_smem);                    // [C][r W]
  constexpr int V = 16 / (int)sizeof(T);                    // elements per 16-byte vector on the patch side
  const int i = blockIdx.x, b = blockIdx.y;
  const int rW = g.r * g.W, HW = g.H * g.W;
  const int64_t xbase = (int64_t)b * HW * g.C;              // this image in X (and in the patch matrix: same element count)
  const int cpr = rW / CH;                                  // chunks per Z-channel
  const int nchunk = g.C * cpr;
  const int Lrow = g.Wr * g.K;                              // elements of this row block of the patch matrix
  const int64_t pbase = xbase + (int64_t)i * Lrow;
  struct alignas(sizeof(T) * CH) Chunk { T v[CH]; };
  struct alignas(16) Vec { T v[V]; };
  auto x_of = [&](int q, int& zoff) {                       // chunk q of the Z image -> element offset in X
    const int cp = q / cpr, u = q - cp * cpr;
    zoff = cp * rW + u * CH;
    const int f = cp * HW + i * rW + u * CH;
    const int w = f / g.HC, rem = f - w * g.HC;
    const int h = rem / g.C, c = rem - h * g.C;
    return (h * g.W + w) * g.C + c;
  };
  auto z_of = [&](int el) {                                 // element el of the row block (patch order) -> LDS index; el % V == 0 walks a run
    const int jp = el / g.K, col = el - jp * g.K;
    const int cp = col / g.rr, pp = col - cp * g.rr;
    const int py = pp / g.r, px = pp - py * g.r;
    return int4{cp * rW + py * g.W + jp * g.r + px, px, py, 0};
  };
  if (!BWD) {
    for (int q = threadIdx.x; q < nchunk; q += 256) {
      int zoff;
      const int xo = x_of(q, zoff);
      *reinterpret_cast<Chunk*>(z + zoff) = *reinterpret_cast<const Chunk*>(src + xbase + xo);
    }
    __syncthreads();
    for (int el = threadIdx.x * V; el < Lrow; el += 256 * V) {
      int4 zi = z_of(el);
      int zidx = zi.x, px = zi.y, py = zi.z;
      Vec o;
#pragma unroll
      for (int k = 0; k < V; ++k) {
        o.v[k] = z[zidx];
        px += 1; zidx += 1;
        if (px == g.r) { px = 0; py += 1; zidx += g.W - g.r; if (py == g.r) { py = 0; zidx += rW - g.r * g.W; } }
      }
      *reinterpret_cast<Vec*>(dst + pbase + el) = o;
    }
  } else {
    for (int el = threadIdx.x * V; el < Lrow; el += 256 * V) {
      int4 zi = z_of(el);
      int zidx = zi.x, px = zi.y, py = zi.z;
      const Vec d = *reinterpret_cast<const Vec*>(src + pbase + el);
#pragma unroll
      for (int k = 0; k < V; ++k) {
        z[zidx] = d.v[k];
        px += 1; zidx += 1;
        if (px == g.r) { px = 0; py += 1; zidx += g.W - g.r; if (py == g.r) { py = 0; zidx += rW - g.r * g.W; } }
      }
    }
    __syncthreads();
    for (int q = threadIdx.x; q < nchunk; q += 256) {
      int zoff;
      const int xo = x_of(q, zoff);
      Chunk c = *reinterpret_cast<const Chunk*>(z + zoff);
      if (ACC) {
        const Chunk old = *reinterpret_cast<const Chunk*>(dst + xbase + xo);
#pragma unroll
        for (int k = 0; k < CH; ++k) c.v[k] = from_f32<T>(to_f32<T>(old.v[k]) + to_f32<T>(c.v[k]));
        if constexpr (sizeof(T) == 4) {                     // (packed fp32 adds feeding a store in the next issue slot: vtx_common.h, vmem_guard)
#pragma unroll
          for (int k = 0; k < CH; ++k) vmem_guard(c.v[k]);
        }
      }
      *reinterpret_cast<Chunk*>(dst + xbase + xo) = c;
    }
  }
}
// widest chunk (elements, <= 16 bytes) the geometry allows, 0: take the element-wise kernel
template <typename T> static int sub_lds_chunk(int H, int W, int C, int r) {
  const int V = 16 / (int)sizeof(T);
  const size_t lds = (size_t)C * r * W * sizeof(T);
  if (lds > 150 * 1024 || (C * r * r) % V != 0) return 0;
  int ch = V;
  while (ch > 1 && ((r * W) % ch || (H * W) % ch || C % ch)) ch >>= 1;
  return ch >= 2 ? ch : 0;
}

static unsigned sub_magic(int d) { return (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); }
// -> geometry + whether the reciprocal form is exact for every dividend the kernel forms (all < H W C)
static bool sub_geom(SubGeom& g, int H, int W, int C, int r) {
  g.H = H; g.W = W; g.C = C; g.r = r; g.rr = r * r; g.K = g.rr * C; g.Wr = W / r; g.HC = H * C;
  g.mK = sub_magic(g.K); g.mrr = sub_magic(g.rr); g.mr = sub_magic(r); g.mWr = sub_magic(g.Wr); g.mHC = sub_magic(g.HC);
  g.mC = sub_magic(C);
  const uint64_t n = (uint64_t)H * W * C;
  const uint64_t dmax = (uint64_t)(g.K > g.HC ? g.K : g.HC);
  return n * dmax < (1ull << 32);
}
template <typename T, bool BWD, bool ACC>
static void sub_launch(const void* src, void* dst, void* dst_t, int B, int H, int W, int C, int r, hipStream_t st) {
  SubGeom g;
  const bool fast = sub_geom(g, H, W, C, r);
  // measured per Twins-SVT-S stage (B = 128, r = 7, bf16; tools/r4/sub_bench.py): 56 x 56 (16-byte chunks) forward 52.5 -> 47.7 us, scatter + add
  // 99.0 -> 31.1; 28 x 28 (8-byte) 25.2 -> 31.7 / 35.3 -> 24.4; 14 x 14 (4-byte) 13.7 -> 21.7 / 20.0 -> 31.4: the gather needs 16-byte
  // chunks to win, the scatter 8-byte ones
  int ch = dst_t == nullptr && vtx_opt(VTX_OPT_TWINS_SUB_LDS) ? sub_lds_chunk<T>(H, W, C, r) : 0;
  if (ch * (int)sizeof(T) < (BWD ? 8 : 16)) ch = 0;
  if (ch > 0) {
    const size_t lds = (size_t)C * r * W * sizeof(T);
    dim3 grid2(H / r, B);
#define SUB_LDS_LAUNCH(CHV)                                                                                                         \
    do {                                                                                                                            \
      auto kern = twins_subsample_lds_kernel<T, CHV, BWD, ACC>;                                                                     \
      if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess) {             \
        hipLaunchKernelGGL(kern, grid2, dim3(256), lds, st, (const T*)src, (T*)dst, g);                                             \
        return;                                                                                                                     \
      }                                                                                                                             \
    } while (0)
    if constexpr (sizeof(T) == 2) { if (ch == 8) SUB_LDS_LAUNCH(8); }
    if (ch == 4) SUB_LDS_LAUNCH(4);
    if (ch == 2) SUB_LDS_LAUNCH(2);
#undef SUB_LDS_LAUNCH
  }
  dim3 grid((H * W * C + 255) / 256, B);
  if (fast) hipLaunchKernelGGL((twins_subsample_kernel<T, BWD, ACC, true>), grid, dim3(256), 0, st, (const T*)src, (T*)dst, (T*)dst_t, g);
  else hipLaunchKernelGGL((twins_subsample_kernel<T, BWD, ACC, false>), grid, dim3(256), 0, st, (const T*)src, (T*)dst, (T*)dst_t, g);
}

// out[row][c] = T(x[row][c] + bias[c]): the epilogue of the split-K path of the reduction conv (fp32 partial sums -> compute dtype)
template <typename T>
__global__ void bias_cast_kernel(const float* __restrict__ x, const float* __restrict__ bias, T* __restrict__ out, int C,
                                 int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // over 8-element vectors
  if (idx >= total) return;
  const int cv = C >> 3;
  const int v = (int)(idx % cv);
  const f32x4 a0 = *reinterpret_cast<const f32x4*>(x + idx * 8), a1 = *reinterpret_cast<const f32x4*>(x + idx * 8 + 4);
  Vec8<T> o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    o.set(e, a0[e] + (bias ? bias[v * 8 + e] : 0.f));
    o.set(4 + e, a1[e] + (bias ? bias[v * 8 + 4 + e] : 0.f));
  }
  store8<T>(out + idx * 8, o);
}

static int dw_groups(int nrows) {                  // workgroups of the weight-gradient pass: <= 1024 (4 per CU)
  return nrows < 1024 ? (nrows < 1 ? 1 : nrows) : 1024;
}
static bool dw_shape_ok(int B, int H, int W, int C) {
  return B > 0 && H > 0 && W > 0 && C >= 8 && C <= 1024 && (C & 7) == 0 && (int64_t)B * H < (1 << 30) && (int64_t)H * W * C < (1ll << 31);
}

extern "C" {

/* y = x + depthwise 3x3 convolution of x (padding 1, no bias) on channels-last [B, H, W, C] features: the positional-encoding
 * generator of Twins-SVT (reference models/twins.py:25-37).  w = the Conv2d weight [C, 1, 3, 3] fp32.  adjoint != 0 applies the
 * mirrored taps: with x := dy this is the input gradient of the same module. */
int vtx_dwconv3_fwd(const void* x, const float* w, void* y, int B, int H, int W, int C, int adjoint, int dtype, void* stream) {
  if (!x || !w || !y) return VTX_ERR_NULL;
  if (!dw_shape_ok(B, H, W, C)) return VTX_ERR_SHAPE;
  const int blocks = B * H;                                     // one workgroup per image row
  const size_t smem = (size_t)9 * C * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VTX_BF16) {
    if (adjoint) hipLaunchKernelGGL((dwconv3_kernel<bf16, true>), dim3(blocks), dim3(DW_THREADS), smem, st, (const bf16*)x, w, (bf16*)y, H, W, C);
    else hipLaunchKernelGGL((dwconv3_kernel<bf16, false>), dim3(blocks), dim3(DW_THREADS), smem, st, (const bf16*)x, w, (bf16*)y, H, W, C);
  } else if (dtype == VTX_F32) {
    if (adjoint) hipLaunchKernelGGL((dwconv3_kernel<float, true>), dim3(blocks), dim3(DW_THREADS), smem, st, (const float*)x, w, (float*)y, H, W, C);
    else hipLaunchKernelGGL((dwconv3_kernel<float, false>), dim3(blocks), dim3(DW_THREADS), smem, st, (const float*)x, w, (float*)y, H, W, C);
  } else return VTX_ERR_DTYPE;
  return vtx_check_launch();
}

size_t vtx_dwconv3_wgrad_workspace(int B, int H, int W, int C) {
  if (!dw_shape_ok(B, H, W, C)) return 0;
  return (size_t)dw_groups(B * H) * 9 * C * sizeof(float);
}

/* dw [C, 1, 3, 3] fp32 (written, not accumulated) from the module input x and the output gradient dy; deterministic. */
int vtx_dwconv3_wgrad(const void* x, const void* dy, float* dw, void* workspace, size_t ws_bytes, int B, int H, int W, int C,
                      int dtype, void* stream) {
  if (!x || !dy || !dw || !workspace) return VTX_ERR_NULL;
  if (!dw_shape_ok(B, H, W, C)) return VTX_ERR_SHAPE;
  if (ws_bytes < vtx_dwconv3_wgrad_workspace(B, H, W, C)) return VTX_ERR_WORKSPACE;
  const int nrows = B * H;
  const int ng = dw_groups(nrows);
  const int run = (nrows + ng - 1) / ng;
  const int lanes = DW_THREADS / (C >> 3);                     // C <= 1024: at least two pixel lanes
  const size_t smem = (size_t)lanes * 9 * C * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)workspace;
  if (dtype == VTX_BF16) {
    auto kern = dwconv3_wgrad_kernel<bf16>;
    if (smem > 64 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return VTX_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(ng), dim3(DW_THREADS), smem, st, (const bf16*)x, (const bf16*)dy, part, H, W, C, nrows, run);
  } else if (dtype == VTX_F32) {
    auto kern = dwconv3_wgrad_kernel<float>;
    if (smem > 64 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return VTX_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(ng), dim3(DW_THREADS), smem, st, (const float*)x, (const float*)dy, part, H, W, C, nrows, run);
  } else return VTX_ERR_DTYPE;
  int rc = vtx_check_launch();
  if (rc) return rc;
  hipLaunchKernelGGL(dwconv3_wgrad_reduce_kernel, dim3((9 * C + 15) / 16), dim3(256), 0, st, (const float*)part, dw, C, ng);
  return vtx_check_launch();
}

/* Patch matrix of the sub-sampling convolution of Twins-SVT's global attention (reference models/twins.py:69-71), with the
 * reference's reshape of the 4-D input kept as written (see csrc/twins_misc.hip): x [B, H, W, C] -> out [B*(H/r)*(W/r), C*r*r],
 * column order (c', py, px) = the Conv2d weight's own layout; out_t (may be null): the same matrix transposed [C*r*r, B*(H/r)*(W/r)].  _bwd scatters the patch gradient back into dx [B, H, W, C] (a permutation; accumulate != 0 adds
 * to what dx holds -- the gradient that also arrives through the query projection). */
int vtx_twins_subsample_fwd(const void* x, void* out, void* out_t, int B, int H, int W, int C, int r, int dtype, void* stream) {
  if (!x || !out) return VTX_ERR_NULL;
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || r <= 0 || H % r || W % r) return VTX_ERR_SHAPE;
  if ((int64_t)H * W * C >= (1ll << 31) || B > 65535) return VTX_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VTX_BF16) sub_launch<bf16, false, false>(x, out, out_t, B, H, W, C, r, st);
  else if (dtype == VTX_F32) sub_launch<float, false, false>(x, out, out_t, B, H, W, C, r, st);
  else return VTX_ERR_DTYPE;
  return vtx_check_launch();
}

int vtx_twins_subsample_bwd(const void* dout, void* dx, int B, int H, int W, int C, int r, int accumulate, int dtype,
                            void* stream) {
  if (!dout || !dx) return VTX_ERR_NULL;
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || r <= 0 || H % r || W % r) return VTX_ERR_SHAPE;
  if ((int64_t)H * W * C >= (1ll << 31) || B > 65535) return VTX_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VTX_BF16) {
    if (accumulate) sub_launch<bf16, true, true>(dout, dx, nullptr, B, H, W, C, r, st);
    else sub_launch<bf16, true, false>(dout, dx, nullptr, B, H, W, C, r, st);
  } else if (dtype == VTX_F32) {
    if (accumulate) sub_launch<float, true, true>(dout, dx, nullptr, B, H, W, C, r, st);
    else sub_launch<float, true, false>(dout, dx, nullptr, B, H, W, C, r, st);
  } else return VTX_ERR_DTYPE;
  return vtx_check_launch();
}

/* out [rows, C] (dtype) = x [rows, C] fp32 + bias [C] fp32 (bias may be null); C % 8 == 0. */
int vtx_bias_cast(const float* x, const float* bias, void* out, int64_t rows, int C, int dtype, void* stream) {
  if (!x || !out) return VTX_ERR_NULL;
  if (rows <= 0 || C <= 0 || (C & 7)) return VTX_ERR_SHAPE;
  const int64_t total = rows * (C >> 3);
  const int blocks = (int)((total + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VTX_BF16) hipLaunchKernelGGL(bias_cast_kernel<bf16>, dim3(blocks), dim3(256), 0, st, x, bias, (bf16*)out, C, total);
  else if (dtype == VTX_F32) hipLaunchKernelGGL(bias_cast_kernel<float>, dim3(blocks), dim3(256), 0, st, x, bias, (float*)out, C, total);
  else return VTX_ERR_DTYPE;
  return vtx_check_launch();
}

}  // extern "C"
