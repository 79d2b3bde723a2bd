// Which per-instruction store shape does a persistent GEMM epilogue want?  (round 4; decides the accumulator layout of gemm_astat.hip)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/store_shapes tools/r4/store_shapes.hip && /tmp/store_shapes
// One 512-thread workgroup per CU walks 128-row strips of a row-major bf16 [M][N] matrix and writes every 128 x 128 tile of
// its strip with nothing else in the kernel.  A wave owns a 32 x 64 piece of the tile (4 x 2 waves); the variants differ ONLY in which
// bytes one wave-wide store instruction covers:
//   0  16 rows x 64 B   dwordx4, lane = (row c_, 16-byte chunk g_)         -- the transposed-product layout (gemm_astat v1/v2)
//   1   4 rows x 128 B  dwordx2, lane = (row g_, 8-byte chunk c_)          -- plain product, weight rows permuted n = 4 c_ + j
//   2   8 rows x 128 B  dwordx4, lane = (row lane / 8, 16-byte chunk lane % 8)   -- after an LDS transposition (gemm_glds_pv)
//   3   4 rows x 64 B   dword,   lane = (row g_, 4-byte chunk c_)          -- plain product, weight rows permuted n = 2 c_ + j
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

template <int MODE> __global__ __launch_bounds__(512) void strip_store(unsigned short* C, int M, int N, int nstrips) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int c_ = lane & 15, g_ = lane >> 4;
  const int ntn = N / 128;
  for (int s = blockIdx.x; s < nstrips; s += gridDim.x) {
    const int m0 = s * 128 + wm * 32;
    for (int tn = 0; tn < ntn; ++tn) {
      const int n0 = tn * 128 + wn * 64;
      if (MODE == 0) {
        for (int i = 0; i < 2; ++i)
          for (int pr = 0; pr < 2; ++pr) {
            const int row = m0 + i * 16 + c_;
            u32x4 v = {1u + tn, 2u, 3u, 4u};
            if (row < M) *reinterpret_cast<u32x4*>(C + (size_t)row * N + n0 + pr * 32 + g_ * 8) = v;
          }
      } else if (MODE == 1) {
        for (int i = 0; i < 2; ++i)
          for (int r = 0; r < 4; ++r) {
            const int row = m0 + i * 16 + g_ * 4 + r;
            u32x2 v = {1u + tn, 2u};
            if (row < M) *reinterpret_cast<u32x2*>(C + (size_t)row * N + n0 + c_ * 4) = v;
          }
      } else if (MODE == 2) {
        for (int it = 0; it < 4; ++it) {
          const int row = m0 + it * 8 + (lane >> 3);
          u32x4 v = {1u + tn, 2u, 3u, 4u};
          if (row < M) *reinterpret_cast<u32x4*>(C + (size_t)row * N + n0 + (lane & 7) * 8) = v;
        }
      } else {
        for (int h = 0; h < 2; ++h)
          for (int i = 0; i < 2; ++i)
            for (int r = 0; r < 4; ++r) {
              const int row = m0 + i * 16 + g_ * 4 + r;
              if (row < M) *reinterpret_cast<unsigned*>(C + (size_t)row * N + n0 + h * 32 + c_ * 2) = 1u + tn;
            }
      }
    }
  }
}

template <int MODE> static void run(unsigned short* C, int M, int N) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int nstrips = (M + 127) / 128;
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(strip_store<MODE>, dim3(nstrips < 256 ? nstrips : 256), dim3(512), 0, 0, C, M, N, nstrips);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const double us = ms * 1e3 / 20, mb = (double)M * N * 2 / 1e6;
  printf("M=%6d N=%4d %6.1f MB  mode %d  %7.1f us  %5.2f TB/s\n", M, N, mb, MODE, us, mb / us);
}

int main() {
  const int shapes[][2] = {{25088, 1152}, {25088, 1536}, {25088, 384}, {50432, 1152}, {50432, 1536}, {32768, 1152}};
  for (auto& sh : shapes) {
    unsigned short* C;
    hipMalloc(&C, (size_t)sh[0] * sh[1] * 2);
    run<0>(C, sh[0], sh[1]); run<1>(C, sh[0], sh[1]); run<2>(C, sh[0], sh[1]); run<3>(C, sh[0], sh[1]);
    hipFree(C);
  }
  return 0;
}
