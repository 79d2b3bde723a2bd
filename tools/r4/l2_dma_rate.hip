// How fast can every CU stream an L2-RESIDENT weight panel into LDS with global_load_lds (the W stream of gemm_astat.hip)?
//   hipcc --offload-arch=gfx950 -O3 -o tools/r4/bin/l2_dma_rate.bin tools/r4/l2_dma_rate.hip
// One workgroup per CU; LW loader waves issue 16-KB k-tiles (128 rows x 128 B, row stride 768 B as in a K = 384 weight) round-robin
// into a ring of NS stages, waiting until at most NS - 1 tiles are in flight; nothing reads the LDS.  Reported: aggregate TB/s and
// us per 16-KB tile and CU.  `span` = bytes of the source panel that is walked (884 KB = the stage-3 qkv weight; 14 MB: past the L2).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

template <int LW, int NS> __global__ __launch_bounds__(64 * LW) void stream(const unsigned char* W, int ntiles_src, int iters, int wg_skew) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int IPW = 16 / LW;                               // 1-KB instructions per wave and tile
  const int lr = lane >> 3, slot = lane & 7;
  // tile t of the source: rows [128 (t / 6), +128) of a [N][384] bf16 matrix, k-columns [64 (t % 6), +64)
  auto issue = [&](int t, int stage) {
    const unsigned char* src = W + (size_t)(t / 6) * 128 * 768 + (t % 6) * 128;
#pragma unroll
    for (int j = 0; j < IPW; ++j) {
      const int r = wave * (128 / LW) + j * 8 + lr;
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(src + (size_t)r * 768 + ((slot ^ (r & 7)) << 4)),
                                       (lds_void_t*)(smem + stage * 16384 + (wave * (128 / LW) + j * 8) * 128), 16, 0, 0);
    }
  };
  int t = (blockIdx.x * wg_skew) % ntiles_src, stage = 0;
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) { issue(t, stage); t = t + 1 == ntiles_src ? 0 : t + 1; stage = stage + 1 == NS ? 0 : stage + 1; }
  for (int i = 0; i < iters; ++i) {
    issue(t, stage); t = t + 1 == ntiles_src ? 0 : t + 1; stage = stage + 1 == NS ? 0 : stage + 1;
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"((NS - 1) * IPW) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int LW, int NS> static void run(const unsigned char* W, int ntiles_src, int skew) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 400;
  auto k = stream<LW, NS>;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(64 * LW), 160 * 1024, 0, W, ntiles_src, iters, skew);     // 160 KB: one workgroup per CU
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const double us = ms * 1e3, bytes = 256.0 * (iters + NS - 1) * 16384;
  printf("span %6.0f KB  skew %2d  loader waves %d  ring %d (in flight %d x 16 KB)  %7.1f us  %6.2f TB/s  %5.3f us per tile and CU\n",
         ntiles_src * 16.384, skew, LW, NS, NS - 1, us, bytes / us / 1e6, us / (iters + NS - 1));
}

int main() {
  unsigned char* W;
  hipMalloc(&W, 64 << 20);
  hipMemset(W, 1, 64 << 20);
  for (int span : {54, 864}) {                 // 54 tiles = 884 KB; 864 tiles = 14 MB
    for (int skew : {0, 7}) {
      run<2, 3>(W, span, skew); run<2, 4>(W, span, skew); run<2, 6>(W, span, skew);
      run<4, 3>(W, span, skew); run<4, 4>(W, span, skew); run<4, 6>(W, span, skew);
      run<8, 3>(W, span, skew); run<8, 4>(W, span, skew); run<8, 6>(W, span, skew);
    }
  }
  return 0;
}
