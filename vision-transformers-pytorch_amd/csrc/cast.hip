// Multi-tensor weight cast: every fp32 Linear / Conv weight of a model -> bf16, plain AND transposed, in ONE
// launch per forward pass.
//
// Under bf16 autocast the reference casts each fp32 weight to bf16 inside every linear / matmul call (GPU autocast
// policy; the casts are cached for the duration of one autocast region only).  Here the same per-forward semantics
// -- the bf16 operands always reflect the CURRENT fp32 parameters, however they were updated: fused optimizers do not
// bump tensor version counters, EMA updates go through .data -- cost one HBM-bound kernel (fp32 read once, two bf16
// writes) instead of ~200 small cast / transpose launches.  The transposed copy [in][out] feeds the LDS-DMA kernel
// for dgrad (gemm_glds.hip), the plain copy [out][in] the forward GEMM.
//
// Work unit: one 64 x 64 tile of one matrix.  `desc` lists the matrices (device array), tiles are numbered
// matrix-major; a block finds its matrix by binary search over the tile prefix sums.
#include "vtx_common.h"

struct CastDesc {
  const float* src;   // [rows][cols] fp32, row-major, contiguous
  int64_t off;        // element offset of this matrix in the flat bf16 outputs
  int rows, cols;
  int tile0;          // index of this matrix's first tile
  int tiles_c;        // tiles along cols
};

__global__ __launch_bounds__(256) void cast_weights_kernel(const CastDesc* __restrict__ desc, int nmat,
                                                          bf16* __restrict__ dst, bf16* __restrict__ dst_t) {
  __shared__ float tile[64][65];
  const int t = blockIdx.x;
  int lo = 0, hi = nmat - 1;
  while (lo < hi) {                                  // last matrix with tile0 <= t
    const int mid = (lo + hi + 1) >> 1;
    if (desc[mid].tile0 <= t) lo = mid; else hi = mid - 1;
  }
  const CastDesc d = desc[lo];
  const int lt = t - d.tile0;
  const int r0 = (lt / d.tiles_c) * 64, c0 = (lt % d.tiles_c) * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;      // 4 rows per pass
  bf16* out = dst + d.off;
  bf16* out_t = dst_t + d.off;
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const int r = r0 + ty + 4 * i, c = c0 + tx;
    float v = 0.f;
    if (r < d.rows && c < d.cols) {
      v = d.src[(int64_t)r * d.cols + c];
      out[(int64_t)r * d.cols + c] = from_f32<bf16>(v);
    }
    tile[ty + 4 * i][tx] = v;
  }
  __syncthreads();
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const int c = c0 + ty + 4 * i, r = r0 + tx;                 // transposed: [cols][rows]
    if (r < d.rows && c < d.cols) out_t[(int64_t)c * d.rows + r] = from_f32<bf16>(tile[tx][ty + 4 * i]);
  }
}

extern "C" {

size_t vtx_cast_desc_bytes(void) { return sizeof(CastDesc); }

/* desc: device array of nmat descriptors {src, off, rows, cols, tile0, tiles_c} (layout: 8, 8, 4, 4, 4, 4 bytes);
 * ntiles = sum over matrices of ceil(rows/64) * ceil(cols/64).  dst / dst_t: flat bf16 buffers; matrix i occupies
 * [off_i, off_i + rows_i * cols_i) in both (dst row-major [rows][cols], dst_t row-major [cols][rows]). */
int vtx_cast_weights(const void* desc, int nmat, int ntiles, void* dst, void* dst_t, void* stream) {
  if (!desc || !dst || !dst_t) return VTX_ERR_NULL;
  if (nmat <= 0 || ntiles <= 0) return VTX_OK;
  hipLaunchKernelGGL(cast_weights_kernel, dim3(ntiles), dim3(256), 0, (hipStream_t)stream, (const CastDesc*)desc, nmat,
                     (bf16*)dst, (bf16*)dst_t);
  return vtx_check_launch();
}

}  // extern "C"
