// Prototype (round 3): weight-STATIONARY GEMM for K = 384 (Swin stage 3 / ViT-S/16): C[M, N] = A[M, 384] . W[N, 384]^T + bias.
// A column chunk of the weight (CN columns x 384, <= 100 KB) is resident in LDS per persistent workgroup (8 waves, one workgroup per
// CU); a wave streams 32-row blocks of A from global memory straight into MFMA operand registers (double-buffered: 2 x 96
// registers), multiplies them with every column pair of the chunk (transposed product, W rows permuted: 8 consecutive output
// columns per lane) and stores 16-byte vectors.  LDS traffic per FLOP is half of the 32 x 64 wave tiles of gemm_glds_pv_kernel
// and A never touches LDS.  Question: what TFLOP/s does this reach (tiled kernel: 450-650 in bench_gemm)?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/wstat.bin tools/probe/wstat_gemm.hip && /tmp/wstat.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int K = 384, KS = 12, WSTR = K;          // no padding: 16-byte chunk q of LDS row r sits at chunk (q & ~15) | ((q & 15) ^ (r & 15))

template <int CN, int NW, int DB>
__global__ __launch_bounds__(64 * NW) void wstat_kernel(const bf16* __restrict__ A, const bf16* __restrict__ W,
                                                        const float* __restrict__ bias, bf16* __restrict__ C, int M, int N,
                                                        int nchunk) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16* ws = reinterpret_cast<bf16*>(smem);
  float* bs = reinterpret_cast<float*>(smem + CN * WSTR * 2);
  // XCD-aware groups: workgroup b runs on XCD b & 7 (observed dispatch rule); the nchunk workgroups that need the SAME rows of A
  // (one per column chunk) sit on one XCD and walk the row blocks in the same order -- the first to ask brings a row block into
  // that XCD's L2, the others hit it (A from beyond L2 arrives at ~10 B/clk/CU: 12 re-reads of A were the whole run time)
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, gpx = (int)(gridDim.x >> 3) / nchunk;   // groups per XCD
  if (slot >= gpx * nchunk) return;
  const int chunk = slot % nchunk, wgc = xcd * gpx + slot / nchunk, nwgc = 8 * gpx;
  const int n0 = chunk * CN;
  for (int i = threadIdx.x; i < CN * (K / 8); i += 64 * NW) {
    const int r = i / (K / 8), q = i % (K / 8);         // LDS row r = 32 np + 16 j + c holds W row 32 np + 8 (c >> 2) + 4 j + (c & 3):
    const int np_ = r >> 5, j_ = (r >> 4) & 1, c_ = r & 15;   // the 16 rows one fragment read touches are CONSECUTIVE in LDS (conflict-free)
    const int n = np_ * 32 + 8 * (c_ >> 2) + 4 * j_ + (c_ & 3);
    *reinterpret_cast<bf16x8*>(ws + r * WSTR + (((q & ~15) | ((q & 15) ^ (r & 15))) << 3)) = *reinterpret_cast<const bf16x8*>(W + (size_t)(n0 + n) * K + q * 8);
  }
  for (int i = threadIdx.x; i < CN; i += 64 * NW) bs[i] = bias[n0 + i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const int nrb = (M + 31) / 32;
  const int stride = nwgc * NW;
  bf16x8 a[DB ? 2 : 1][2][KS];                                   // [buffer][m tile][k step]
  auto load_a = [&](int rb, int buf) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int row = min(rb * 32 + mt * 16 + c, M - 1);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) a[buf][mt][ks] = *reinterpret_cast<const bf16x8*>(A + (size_t)row * K + ks * 32 + g * 8);
    }
  };
  auto compute = [&](int rb, int buf) {
#pragma unroll 1
    for (int np = 0; np < CN / 32; ++np) {
      f32x4 acc[2][2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[mt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      bf16x8 wf[KS][2];                                 // every weight fragment of the pair requested before the first MFMA
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          wf[ks][j] = *reinterpret_cast<const bf16x8*>(ws + (np * 32 + 16 * j + c) * WSTR + ((((ks * 4 + g) & ~15) | (((ks * 4 + g) & 15) ^ c)) << 3));
        }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][j], a[buf][mt][ks], acc[mt][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(bs + np * 32 + 8 * g), b1 = *reinterpret_cast<const f32x4*>(bs + np * 32 + 8 * g + 4);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int row = rb * 32 + mt * 16 + c;
        if (row < M) {
          bf16x8 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) { o[r] = (bf16)(acc[mt][0][r] + b0[r]); o[4 + r] = (bf16)(acc[mt][1][r] + b1[r]); }
          *reinterpret_cast<bf16x8*>(C + (size_t)row * N + n0 + np * 32 + 8 * g) = o;
        }
      }
    }
  };
  int rb = wgc * NW + wave;
  if (DB) {
    if (rb < nrb) load_a(rb, 0);
    for (; rb < nrb; rb += 2 * stride) {                  // two row blocks per trip: static register buffers 0 / 1
      if (rb + stride < nrb) load_a(rb + stride, 1);
      compute(rb, 0);
      if (rb + stride < nrb) {
        if (rb + 2 * stride < nrb) load_a(rb + 2 * stride, 0);
        compute(rb + stride, 1);
      }
    }
  } else {
    for (; rb < nrb; rb += stride) { load_a(rb, 0); compute(rb, 0); }   // single buffer: the other waves of the SIMD cover the loads
  }
}

static float frand() { return (float)rand() / RAND_MAX - 0.5f; }

template <int CN, int NW, int DB> static void run(int M, int N) {
  std::vector<bf16> ha((size_t)M * K), hw((size_t)N * K);
  std::vector<float> hb(N);
  for (auto& v : ha) v = (bf16)frand();
  for (auto& v : hw) v = (bf16)(frand() * 0.1f);
  for (auto& v : hb) v = frand();
  bf16 *a, *w, *c; float* b;
  hipMalloc(&a, ha.size() * 2); hipMalloc(&w, hw.size() * 2); hipMalloc(&b, N * 4); hipMalloc(&c, (size_t)M * N * 2 * 3);
  hipMemcpy(a, ha.data(), ha.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(b, hb.data(), N * 4, hipMemcpyHostToDevice);
  const int smem = CN * WSTR * 2 + CN * 4;
  const int nchunk = N / CN;
  auto kern = wstat_kernel<CN, NW, DB>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  const int wgs = 256;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(wgs), dim3(64 * NW), smem, 0, a, w, b, c, M, N, nchunk);
  hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(wgs), dim3(64 * NW), smem, 0, a, w, b, c + (size_t)(i % 3) * M * N, M, N, nchunk);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double fl = 2.0 * M * N * K;
  printf("M=%d N=%d chunk=%d waves=%d db=%d workgroups=%d: %.1f us  %.0f TFLOP/s\n", M, N, CN, NW, DB, wgs, ms * 1e3 / 20, fl / (ms / 20 * 1e-3) / 1e12);
  std::vector<bf16> hc((size_t)32 * N);
  const int r0 = M - 32;
  hipMemcpy(hc.data(), c + (size_t)r0 * N, hc.size() * 2, hipMemcpyDeviceToHost);
  double maxerr = 0;
  for (int r = 0; r < 32; ++r)
    for (int n = 0; n < N; n += 7) {
      double s = hb[n];
      for (int k = 0; k < K; ++k) s += (double)(float)ha[(size_t)(r0 + r) * K + k] * (double)(float)hw[(size_t)n * K + k];
      const double e = fabs(s - (double)(float)hc[(size_t)r * N + n]);
      if (e > maxerr) maxerr = e;
    }
  printf("   max abs error (sampled) vs fp64: %.4f\n", maxerr);
  hipFree(a); hipFree(w); hipFree(b); hipFree(c);
}

int main(int argc, char** argv) {
  if (argc > 1) { run<128, 8, 0>(50432, 1536); return 0; }      // (one configuration: counter passes)
  run<128, 4, 1>(50432, 1536);
  run<128, 8, 0>(50432, 1536);
  run<128, 4, 1>(25088, 1152);
  run<128, 8, 0>(25088, 1152);
  run<128, 8, 0>(25088, 384);
  return 0;
}
