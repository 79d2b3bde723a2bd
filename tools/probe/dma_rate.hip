// Upper bound of the L2 -> LDS DMA path (global_load_lds_dwordx4) per CU: workgroups of 8 waves stream 128-byte rows
// (row stride 768 B, as a K = 384 bf16 operand) of an L2-resident panel into LDS with DEPTH "k-tiles" of 4 instructions
// per wave in flight (counted vmcnt), no consumer.  Prints KB/us per CU for depth x workgroups-per-CU.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/dma_rate tools/probe/dma_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;
extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
template <int DEPTH> __global__ __launch_bounds__(512) void dma_kernel(const unsigned char* src, size_t panel_bytes, int ktiles, int rowstride, float* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // 256 rows x 128 B per k-tile: wave w owns rows 32 w .. 32 w + 31 = 4 instructions of 8 rows
  const unsigned char* base = src + ((size_t)blockIdx.x * 37 % 64) * (panel_bytes / 64 / 256 * 256);   // spread over the panel
  const int lr = lane >> 3, slot = lane & 7;
  const unsigned char* p[4];
  for (int j = 0; j < 4; ++j) p[j] = base + (size_t)(wave * 32 + j * 8 + lr) * rowstride + slot * 16;
  for (int kt = 0; kt < ktiles; ++kt) {
    unsigned char* dst = smem + (kt % DEPTH) * 32768 + wave * 4096;
    const int koff = (kt % (rowstride / 128)) * 128;
#pragma unroll
    for (int j = 0; j < 4; ++j) __builtin_amdgcn_global_load_lds((gbl_void_t*)(p[j] + koff), (lds_void_t*)(dst + j * 1024), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(4 * (DEPTH - 1)) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (ktiles < 0) sink[blockIdx.x] = smem[threadIdx.x];
}
template <int DEPTH> void run(const unsigned char* src, size_t bytes, int wgs_per_cu, float* sink) {
  const int ktiles = 96, smem_bytes = wgs_per_cu == 1 ? 131072 : (wgs_per_cu == 2 ? 65536 : 49152);
  if (DEPTH * 32768 > smem_bytes) return;
  hipFuncSetAttribute((const void*)dma_kernel<DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * wgs_per_cu;
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(dma_kernel<DEPTH>, dim3(grid), dim3(512), smem_bytes, 0, src, bytes, ktiles, 768, sink);
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(dma_kernel<DEPTH>, dim3(grid), dim3(512), smem_bytes, 0, src, bytes, ktiles, 768, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 100.0, kb = (double)wgs_per_cu * ktiles * 32.0;
  printf("depth %d, %d workgroups per CU: %7.1f us, %6.1f KB/us per CU, %5.2f TB/s chip, %.2f us per k-tile round\n", DEPTH, wgs_per_cu, us, kb / us,
         kb * 256 / us / 1e6 * 1.024, us / ktiles);
}
int main() {
  const size_t bytes = 2u << 20;               // 2 MB panel: L2 resident on every XCD
  unsigned char* src; hipMalloc(&src, bytes + (1 << 20)); hipMemset(src, 1, bytes + (1 << 20));
  float* sink; hipMalloc(&sink, 1 << 16);
  for (int w = 1; w <= 3; ++w) { run<1>(src, bytes, w, sink); run<2>(src, bytes, w, sink); run<3>(src, bytes, w, sink); run<4>(src, bytes, w, sink); }
  return 0;
}
