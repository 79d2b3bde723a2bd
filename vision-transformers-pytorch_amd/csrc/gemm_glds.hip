// LDS-DMA GEMM for gfx950: C[M,N] = epilogue(A[M,K] . B[N,K]^T), bf16 in, fp32 accumulate, K % 64 == 0.
//
// The register-staged kernel in gemm.hip is LDS-WRITE bound on these shapes: per 128x128x64 block-tile the
// 32 KB of operands cost ~400 LDS cycles as ds_write_b128 (~79 B/clk/CU) next to ~256 cycles of fragment
// reads and ~515 cycles of MFMA at peak.  Here the operand tiles go HBM/L2 -> LDS directly with
// global_load_lds_dwordx4 (no VGPR round trip, no ds_write), double-buffered so the loads of k-tile t+1 fly
// while tile t is multiplied, with ONE barrier per k-tile.
//
// LDS image: global_load_lds writes lane-linear (wave-uniform base + lane*16 B), so a tile is stored as
// plain [rows][64 bf16] = 128-byte rows; bank conflicts of the ds_read_b128 fragment reads are removed by an
// XOR swizzle applied on the SOURCE address: the 16-byte chunk q of row r is stored in slot q ^ (r & 7), and
// read back from there (conflict-free for all four 16-lane ds_read_b128 groups of the 16x16x32 fragments).
// Used for the forward GEMMs and -- with a transposed bf16 weight copy made once per step -- for dgrad.
#include <stdlib.h>

#include "gemm_common.h"

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

template <int BN>
__global__ __launch_bounds__(256) void gemm_glds_kernel(GemmArgs p) {
  constexpr int BM = 128, BK = 64, ROWB = 128;          // 64 bf16 per LDS row
  constexpr int WM = BM / 32, WN = BN / 32;
  constexpr int STAGE = (BM + BN) * ROWB;               // bytes per pipeline stage
  static_assert(2 * STAGE >= (BM / 2) * (BN + 4) * 4, "C staging must fit");
  extern __shared__ __attribute__((aligned(16))) unsigned char glds_smem[];   // [2][STAGE]

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int c_ = lane & 15, g_ = lane >> 4;

  const int ntn = gridDim.x, ntm = gridDim.y;
  const int nblk = ntn * ntm;
  const int did = blockIdx.y * ntn + blockIdx.x;
  const int xq = nblk >> 3, xr = nblk & 7, xcd = did & 7;
  const int lid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (did >> 3);
  const int tn = lid % ntn, tm = lid / ntn;
  const int m0 = tm * BM, n0 = tn * BN;
  const int nk = p.K / BK;

  const bf16* A = (const bf16*)p.A;
  const bf16* B = (const bf16*)p.B;

  // per-lane source pointers of this wave's DMA pieces (8 rows x 128 B per instruction)
  const int lr = lane >> 3, slot = lane & 7;
  constexpr int APW = BM / 32, BPW = BN / 32;            // pieces per wave for A / B
  const bf16* asrc[APW];
  const bf16* bsrc[BPW];
#pragma unroll
  for (int j = 0; j < APW; ++j) {
    const int r = wave * (BM / 4) + j * 8 + lr;
    const int row = min(m0 + r, p.M - 1);                // rows past M are never stored: any valid address will do
    asrc[j] = A + (int64_t)row * p.lda + ((slot ^ (r & 7)) << 3);
  }
#pragma unroll
  for (int j = 0; j < BPW; ++j) {
    const int r = wave * (BN / 4) + j * 8 + lr;
    const int row = min(n0 + r, p.N - 1);
    bsrc[j] = B + (int64_t)row * p.ldb + ((slot ^ (r & 7)) << 3);
  }

  auto issue = [&](int kt, int buf) {
    unsigned char* sa = glds_smem + buf * STAGE + wave * (BM / 4) * ROWB;
    unsigned char* sb = glds_smem + buf * STAGE + BM * ROWB + wave * (BN / 4) * ROWB;
#pragma unroll
    for (int j = 0; j < APW; ++j)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(asrc[j] + kt * BK), (lds_void_t*)(sa + j * 8 * ROWB), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < BPW; ++j)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(bsrc[j] + kt * BK), (lds_void_t*)(sb + j * 8 * ROWB), 16, 0, 0);
  };

  f32x4 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) issue(kt + 1, buf ^ 1);             // everyone left the barrier after finishing tile kt-1
    const unsigned char* la = glds_smem + buf * STAGE;
    const unsigned char* lb = la + BM * ROWB;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      Vec8<bf16> fa[WM], fb[WN];
#pragma unroll
      for (int i = 0; i < WM; ++i) {
        const int r = wm * (BM / 2) + i * 16 + c_;
        fa[i] = load8<bf16>(reinterpret_cast<const bf16*>(la + r * ROWB + (((ks * 4 + g_) ^ (r & 7)) << 4)));
      }
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int r = wn * (BN / 2) + j * 16 + c_;
        fb[j] = load8<bf16>(reinterpret_cast<const bf16*>(lb + r * ROWB + (((ks * 4 + g_) ^ (r & 7)) << 4)));
      }
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) mma16(fa[i], fb[j], acc[i][j]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // tile kt+1 has landed (this wave's pieces)
    __syncthreads();                                       // ... and everybody's; tile kt's buffer is free
  }

  gemm_epilogue<bf16, bf16, BM, BN>(p, acc, glds_smem, m0, n0, 0, wm, wn, c_, g_);
}

template <int BN> static int glds_launch_bn(const GemmArgs& a, hipStream_t st) {
  constexpr size_t smem = (size_t)2 * (128 + BN) * 128;
  dim3 grid((a.N + BN - 1) / BN, (a.M + 127) / 128, 1);
  hipLaunchKernelGGL((gemm_glds_kernel<BN>), grid, dim3(256), smem, st, a);
  return vtx_check_launch();
}

bool gemm_glds_enabled() {
  static int on = -1;
  if (on < 0) { const char* e = getenv("VTX_GEMM_GLDS"); on = e ? atoi(e) : 1; }
  return on != 0;
}

int gemm_glds_launch(const GemmArgs& a, hipStream_t st) {
  if (a.N % 128 == 0) return glds_launch_bn<128>(a, st);
  if (a.N % 96 == 0) return glds_launch_bn<96>(a, st);
  if (a.N <= 64) return glds_launch_bn<64>(a, st);
  return glds_launch_bn<128>(a, st);
}
