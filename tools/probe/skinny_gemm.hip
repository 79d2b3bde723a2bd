// Prototype (round 3): weight-resident streaming GEMM for the K = 96 GEMMs of Swin stage 1 (M = 401 408):
// C[M, N] = A[M, 96] . W[N, 96]^T + bias, bf16 in / out.  The whole weight sits in LDS once per persistent workgroup; a wave
// streams 32-row blocks of A straight from global memory into MFMA operand registers (no LDS for A), the product is taken
// TRANSPOSED (W rows as the MFMA A operand, permuted so that a lane ends up with 8 consecutive output columns of one row) and
// stored with 16-byte vectors.  Question: how close to the HBM floor (49 us for N = 288, 110 us for N = 384 with two outputs)?
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe/skinny_gemm.bin tools/probe/skinny_gemm.hip && tools/probe/skinny_gemm.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int K = 96, KS = 3, WSTR = 104;           // LDS row stride of W (elements): 208 B

template <int N, bool TWO_OUT, int NW, int VAR>
__global__ __launch_bounds__(64 * NW) void skinny_kernel(const bf16* __restrict__ A, const bf16* __restrict__ W,
                                                     const float* __restrict__ bias, bf16* __restrict__ C,
                                                     bf16* __restrict__ C2, int M, int Nrt, long ldrt) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16* ws = reinterpret_cast<bf16*>(smem);
  float* bs = reinterpret_cast<float*>(smem + N * WSTR * 2);
  for (int i = threadIdx.x; i < N * (K / 8); i += 64 * NW) {
    const int n = i / (K / 8), q = i % (K / 8);
    *reinterpret_cast<bf16x8*>(ws + n * WSTR + q * 8) = *reinterpret_cast<const bf16x8*>(W + n * K + q * 8);
  }
  for (int i = threadIdx.x; i < N; i += 64 * NW) bs[i] = bias[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const int NN = (VAR & 4) ? Nrt : N;
  const long LD = (VAR & 4) ? ldrt : N;
  const int nrb = (M + 31) / 32;
  const int stride = gridDim.x * NW;
  bf16x8 an[2][KS];
  auto load_a = [&](int rb) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int row = min(rb * 32 + mt * 16 + c, M - 1);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) an[mt][ks] = *reinterpret_cast<const bf16x8*>(A + (int64_t)row * K + ks * 32 + g * 8);
    }
  };
  int rb = blockIdx.x * NW + wave;
  if (rb < nrb) load_a(rb);
  for (; rb < nrb; rb += stride) {
    bf16x8 a[2][KS];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) a[mt][ks] = an[mt][ks];
    if (rb + stride < nrb) load_a(rb + stride);
#pragma unroll (VAR & 2 ? 1 : 3)
    for (int np = 0; np < NN / 32; ++np) {
      f32x4 acc[2][2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[mt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = np * 32 + 8 * (c >> 2) + 4 * j + (c & 3);       // the W row this lane supplies as MFMA A-operand row c
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const bf16x8 wf = *reinterpret_cast<const bf16x8*>(ws + n * WSTR + ks * 32 + g * 8);
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, a[mt][ks], acc[mt][j], 0, 0, 0);
        }
      }
      // acc[mt][j][r] = C[row = 32 rb + 16 mt + c][col = 32 np + 8 g + 4 j + r]
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(bs + np * 32 + 8 * g), b1 = *reinterpret_cast<const f32x4*>(bs + np * 32 + 8 * g + 4);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int row = rb * 32 + mt * 16 + c;
        if (row < M) {
          bf16x8 o, o2;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float z0 = acc[mt][0][r] + b0[r], z1 = acc[mt][1][r] + b1[r];
            o[r] = (bf16)z0; o[4 + r] = (bf16)z1;
            if (TWO_OUT) {
              const float zz0 = (float)o[r], zz1 = (float)o[4 + r];
              o2[r] = (bf16)(zz0 * __builtin_amdgcn_rcpf(1.f + __expf(-zz0)));
              o2[4 + r] = (bf16)(zz1 * __builtin_amdgcn_rcpf(1.f + __expf(-zz1)));
            }
          }
          if (VAR & 1) {
            __builtin_nontemporal_store(o, reinterpret_cast<bf16x8*>(C + (int64_t)row * LD + np * 32 + 8 * g));
            if (TWO_OUT) __builtin_nontemporal_store(o2, reinterpret_cast<bf16x8*>(C2 + (int64_t)row * LD + np * 32 + 8 * g));
          } else {
            *reinterpret_cast<bf16x8*>(C + (int64_t)row * LD + np * 32 + 8 * g) = o;
            if (TWO_OUT) *reinterpret_cast<bf16x8*>(C2 + (int64_t)row * LD + np * 32 + 8 * g) = o2;
          }
        }
      }
    }
  }
}

static float frand() { return (float)rand() / RAND_MAX - 0.5f; }

template <int N, bool TWO, int NW, int VAR> static void run(int M) {
  std::vector<bf16> ha((size_t)M * K), hw((size_t)N * K);
  std::vector<float> hb(N);
  for (auto& v : ha) v = (bf16)frand();
  for (auto& v : hw) v = (bf16)(frand() * 0.2f);
  for (auto& v : hb) v = frand();
  bf16 *a, *w, *c, *c2; float* b;
  hipMalloc(&a, ha.size() * 2); hipMalloc(&w, hw.size() * 2); hipMalloc(&b, N * 4);
  hipMalloc(&c, (size_t)M * N * 2 * 4); hipMalloc(&c2, (size_t)M * N * 2 * 4);      // 4 output buffers, used in rotation
  hipMemcpy(a, ha.data(), ha.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(b, hb.data(), N * 4, hipMemcpyHostToDevice);
  const int smem = N * WSTR * 2 + N * 4;
  auto kern = skinny_kernel<N, TWO, NW, VAR>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  for (int wgs : {256}) {
    if ((size_t)smem * (wgs / 256) > 160 * 1024) continue;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(wgs), dim3(64 * NW), smem, 0, a, w, b, c, c2, M, N, (long)N);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(wgs), dim3(64 * NW), smem, 0, a, w, b, c + (size_t)(i & 3) * M * N, c2 + (size_t)(i & 3) * M * N, M, N, (long)N);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)M * K * 2 + (double)M * N * 2 * (TWO ? 2 : 1);
    printf("N=%d two_out=%d waves=%d var=%d workgroups=%d: %.1f us  (%.0f GB/s algorithmic; floor %.1f us at 6.3 TB/s)\n", N, (int)TWO, NW, VAR, wgs,
           ms * 1e3 / 20, bytes / (ms / 20 * 1e-3) / 1e9, bytes / 6.3e12 * 1e6);
  }
  // spot check against the host
  std::vector<bf16> hc((size_t)64 * N);
  const int r0 = M - 64;
  hipMemcpy(hc.data(), c + (size_t)r0 * N, hc.size() * 2, hipMemcpyDeviceToHost);
  double maxerr = 0;
  for (int r = 0; r < 64; ++r)
    for (int n = 0; n < N; ++n) {
      double s = hb[n];
      for (int k = 0; k < K; ++k) s += (double)(float)ha[(size_t)(r0 + r) * K + k] * (double)(float)hw[(size_t)n * K + k];
      const double e = fabs(s - (double)(float)hc[(size_t)r * N + n]);
      if (e > maxerr) maxerr = e;
    }
  printf("   max abs error of the last 64 rows vs fp64: %.4f\n", maxerr);
  hipFree(a); hipFree(w); hipFree(b); hipFree(c); hipFree(c2);
}

int main() {
  const int M = 401408;
  run<288, false, 4, 0>(M);
  run<288, false, 4, 4>(M);
  run<384, true, 4, 0>(M);
  run<384, true, 4, 4>(M);
  return 0;
}
