// What HBM write rate does the GEMM epilogue's store pattern reach, with nothing else in the kernel?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/store_pattern tools/probe/store_pattern.hip && /tmp/store_pattern
// Each workgroup (256 threads) owns a BM x 128 bf16 output tile of a row-major [M][N] matrix and writes it as the
// epilogue does: per instruction a wavefront stores 4 rows x 256 contiguous bytes.  Variants:
//   mode 0: stores only;  mode 1: + the two LDS staging passes and their barriers;  mode 2: linear (grid-stride) fill
//   lds > 0: dynamic LDS per workgroup (bytes), to limit the resident workgroups per CU as the GEMM's ring does
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int BM> __global__ __launch_bounds__(256) void tile_store(unsigned short* C, int M, int N, int mode) {
  extern __shared__ float cbuf[];
  const int ntn = N / 128;
  const int nblk = gridDim.x;
  const int did = blockIdx.x;
  const int xq = nblk >> 3, xr = nblk & 7, xcd = did & 7;
  const int lid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (did >> 3);
  const int tn = lid % ntn, tm = lid / ntn;
  const int m0 = tm * BM, n0 = tn * 128;
  f32x4 val = {1.f, 2.f, 3.f, 4.f};
  for (int pass = 0; pass < 2; ++pass) {
    if (mode == 1) {
      for (int e = 0; e < BM / 8; ++e) cbuf[(threadIdx.x >> 4) * 132 + (threadIdx.x & 15) + e * 16] = (float)e;
      __syncthreads();
    }
    for (int it = 0; it < BM / 32; ++it) {
      const int v = threadIdx.x + 256 * it;
      const int lr = v >> 4, cv = v & 15;
      const int row = m0 + pass * (BM / 2) + lr;
      if (mode == 1) val = *reinterpret_cast<const f32x4*>(cbuf + (lr % 16) * 132 + cv * 8);
      if (row < M) *reinterpret_cast<f32x4*>(C + (size_t)row * N + n0 + cv * 8) = val;
    }
    if (mode == 1 && pass == 0) __syncthreads();
  }
}

__global__ __launch_bounds__(256) void linear_fill(f32x4* C, size_t nvec) {
  f32x4 val = {1.f, 2.f, 3.f, 4.f};
  for (size_t i = blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) C[i] = val;
}

int main() {
  const int shapes[][2] = {{100352, 768}, {100352, 640}, {25088, 1536}, {25088, 1152}, {401408, 384}, {6272, 3072}};
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (auto& sh : shapes) {
    const int M = sh[0], N = sh[1];
    unsigned short* C;
    hipMalloc(&C, (size_t)M * N * 2);
    const double mb = (double)M * N * 2 / 1e6;
    for (int bm : {64, 128})
      for (int mode : {0, 1})
        for (int lds : {0, 48 * 1024, 64 * 1024}) {
          if (mode == 0 && lds == 64 * 1024) continue;
          const int grid = (M + bm - 1) / bm * (N / 128);
          const int smem = lds > 20000 ? lds : 20000;
          auto k64 = tile_store<64>; auto k128 = tile_store<128>;
          hipFuncSetAttribute((const void*)k64, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
          hipFuncSetAttribute((const void*)k128, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
          float ms = 0;
          for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            for (int i = 0; i < 20; ++i) {
              if (bm == 64) hipLaunchKernelGGL(k64, dim3(grid), dim3(256), smem, 0, C, M, N, mode);
              else hipLaunchKernelGGL(k128, dim3(grid), dim3(256), smem, 0, C, M, N, mode);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
          }
          const double us = ms * 1e3 / 20;
          printf("M=%6d N=%4d %6.1f MB  BM=%3d mode=%d lds=%5d  %7.1f us  %5.2f TB/s\n", M, N, mb, bm, mode, smem, us, mb / us);
        }
    float ms = 0;
    for (int g : {2048, 8192}) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(linear_fill, dim3(g), dim3(256), 0, 0, (f32x4*)C, (size_t)M * N / 8);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
      }
      printf("M=%6d N=%4d %6.1f MB  linear fill grid %d  %7.1f us  %5.2f TB/s\n", M, N, mb, g, ms * 1e3 / 20, mb / (ms * 1e3 / 20));
    }
    hipFree(C);
  }
  return 0;
}
