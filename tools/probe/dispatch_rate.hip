// How fast can the chip start workgroups?  Empty kernels at the GEMM's launch geometry (512 threads, 0 / 48 / 64 KB of
// dynamic LDS), 4728 workgroups (ViT-S/16 fc1 forward with 128 x 128 tiles).  hipcc --offload-arch=gfx950 -O3 dispatch_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
extern __shared__ unsigned char smem[];
template <int WORK> __global__ __launch_bounds__(512) void k_empty(float* out, int n) {
  if (WORK) { smem[threadIdx.x] = (unsigned char)threadIdx.x; __syncthreads(); }
  if (n < 0) out[blockIdx.x] = smem[0];
}
template <int NT> __global__ __launch_bounds__(NT) void k_empty_nt(float* out, int n) {
  if (n < 0) out[blockIdx.x] = 1.f;
}
int main() {
  float* d; hipMalloc(&d, 1 << 20);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grids[] = {512, 1176, 2352, 4728, 9456};
  const int ldss[] = {0, 16384, 49152, 65536};
  hipFuncSetAttribute((const void*)k_empty<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute((const void*)k_empty<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int lds : ldss) for (int g : grids) {
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k_empty<0>, dim3(g), dim3(512), lds, 0, d, 1);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_empty<0>, dim3(g), dim3(512), lds, 0, d, 1);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("512 thr, lds %5d B, grid %5d: %7.2f us per launch = %6.1f WG/us\n", lds, g, ms * 50.f, g / (ms * 50.f));
  }
  for (int g : grids) {
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_empty_nt<256>, dim3(g), dim3(256), 0, 0, d, 1);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("256 thr, lds     0 B, grid %5d: %7.2f us per launch = %6.1f WG/us\n", g, ms * 50.f, g / (ms * 50.f));
  }
  return 0;
}
