// Debug harness for wgrad_glds_kernel: dumps the first A / B fragments of block 0 wave 0.  GPU box only.
#define WG_DEBUG 1
#include "../../vision-transformers-pytorch_amd/csrc/gemm_wgrad_glds.hip"
#include <stdio.h>
#include <vector>
int main() {
  const int M = 64, N = 128, K = 128;
  std::vector<unsigned short> dy(M * N), x(M * K);
  auto bf = [](float f) { union { float f; unsigned u; } c; c.f = f; return (unsigned short)(c.u >> 16); };
  for (int t = 0; t < M; ++t) for (int n = 0; n < N; ++n) dy[t * N + n] = bf((float)(t));        // value = token
  for (int t = 0; t < M; ++t) for (int k = 0; k < K; ++k) x[t * K + k] = bf((float)(t));
  void *ddy, *dx; float* dC;
  hipMalloc(&ddy, dy.size() * 2); hipMalloc(&dx, x.size() * 2); hipMalloc(&dC, (128 * 128 + 2048) * 4);
  hipMemcpy(ddy, dy.data(), dy.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dx, x.data(), x.size() * 2, hipMemcpyHostToDevice);
  int rc = wgrad_glds_launch(ddy, dx, dC, nullptr, M, N, K, N, K, nullptr, 1, 0.f, 1, 128, 0);
  hipDeviceSynchronize();
  std::vector<float> h(128 * 128 + 2048);
  hipMemcpy(h.data(), dC, h.size() * 4, hipMemcpyDeviceToHost);
  printf("rc %d  dW[0][0] = %f (expect sum t^2 = %f)\n", rc, h[0], 85344.0);
  for (int l = 0; l < 64; ++l) { if (l % 16 > 1 && l % 16 < 15) continue;
    printf("lane %2d A:", l); for (int e = 0; e < 8; ++e) printf(" %2.0f", h[128*128 + l*8 + e]);
    printf("   B:"); for (int e = 0; e < 8; ++e) printf(" %2.0f", h[128*128 + 512 + l*8 + e]); printf("\n"); }
  printf("LDS row first elements:"); for (int r = 0; r < 64; ++r) printf(" %2.0f", h[128*128 + 1024 + r]); printf("\n");
  return 0;
}
