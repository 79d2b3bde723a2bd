// EXPERIMENTAL (option GEMM_WS, off by default): persistent, wave-specialised variant of the LDS-DMA GEMM
// C[M,N] = epilogue(A[M,K] . B[N,K]^T), bf16, K % 64 == 0, N % 128 == 0, 128 x 128 tiles.
//
// Why: the phase ablation of gemm_glds_kernel (profiles/round2_gemm_phase_ablation.txt) shows that the parts of a
// launch ADD -- first k-tile latency + main loop + epilogue math + the HBM drain of the stores -- instead of overlapping.
// On gfx950 loads and stores of a wave share ONE in-order-agnostic counter (vmcnt), so a wave that has just issued its
// epilogue stores cannot wait for the next tile's DMA without also waiting for those stores; and a workgroup that exits
// holds its LDS and wave slots until its stores have drained.  Here the two kinds of memory traffic live in DIFFERENT
// waves of a persistent workgroup:
//   * waves 8-11 are PRODUCERS: they only issue global_load_lds into an NS-stage ring and wait for it (their vmcnt
//     never sees a store), and run straight on into the next tile's k-tiles while the consumers are in their epilogue;
//   * waves 0-7 are CONSUMERS (2 x 4 waves of 64 x 32, the tile layout of gemm_glds_kernel<128,128,64,2,4>): ds_read
//     fragments, MFMA, fused epilogue through a staging buffer of its own, stores -- and never wait for a load of the
//     main loop, so the stores of tile i drain under the MFMAs of tile i + 1;
//   * ONE s_barrier per k-tile orders the ring (producers have made tile s + 1 visible, consumers have released tile s).
// Same summation order per output element as gemm_glds_kernel: bitwise identical results (tests/test_gpu_dispatch.py).
#include "gemm_common.h"
#include "options.h"

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

namespace {
constexpr int WS_BM = 128, WS_BN = 128, WS_BK = 64, WS_ROWB = WS_BK * 2, WS_STAGE = (WS_BM + WS_BN) * WS_ROWB;   // 32 KB
constexpr int WS_CONS = 8, WS_PROD = 4, WS_NT = 64 * (WS_CONS + WS_PROD);
constexpr int WS_CBYTES = (WS_BM / 2) * (WS_BN + 4) * 4;                                                          // 33 792 B
__device__ __forceinline__ int ws_swz(int r) { return r & 7; }
}  // namespace

// epilogue of the consumer waves: gemm_epilogue (gemm_common.h) with raw barriers -- __syncthreads() would put
// s_waitcnt vmcnt(0) in front of the barrier and drain the stores of the first pass before the second one starts
template <typename EO>
__device__ __forceinline__ void ws_epilogue(const GemmArgs& p, f32x4 (&acc)[4][2], float* cbuf, int m0, int n0, int wm,
                                            int wn, int c_, int g_, const EO& eo, bool consumer) {
  constexpr int BM = WS_BM, BN = WS_BN, NWN = 4, WM = 4, WN = 2, NT = 512, CSTR = BN + 4;
  constexpr int VROW = EO::VROW, NIT = EO::NIT;
  bf16* __restrict__ Cout = (bf16*)p.C;
  const bf16* __restrict__ resid = (const bf16*)p.resid;
  bf16* __restrict__ aux_out = (bf16*)p.aux_out;
  const bool act_fwd = p.act == 1 || p.act == 3, act_bwd = p.act == 2 || p.act == 4;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    if (consumer) {
#pragma unroll
      for (int ii = 0; ii < WM / 2; ++ii) {
        const int i = pass * (WM / 2) + ii;
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            cbuf[(wm * (BM / 4) + ii * 16 + g_ * 4 + r) * CSTR + wn * (BN / NWN) + j * 16 + c_] = acc[i][j][r] + eo.bcol[j];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (consumer) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int q = pass * NIT + it;
        int row, col;
        if (!EO::where(p, m0, n0, pass, it, row, col)) continue;
        const int64_t off = (int64_t)row * p.ldc + col;
        const int v = threadIdx.x + NT * it;
        const int lr = v / VROW, cv = v - lr * VROW;
        const float* cp = cbuf + lr * CSTR + cv * 8;
        f32x4 lo = *reinterpret_cast<const f32x4*>(cp), hi = *reinterpret_cast<const f32x4*>(cp + 4);
        float val[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        if (act_fwd) {
          Vec8<bf16> z;
#pragma unroll
          for (int e = 0; e < 8; ++e) z.set(e, val[e]);
          if (p.act == 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) val[e] = silu_f(z.get(e));
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) val[e] = gelu_f(z.get(e));
          }
          if (aux_out) store8<bf16>(aux_out + off, z);
        } else if (act_bwd) {
          if (p.act == 2) {
#pragma unroll
            for (int e = 0; e < 8; ++e) val[e] *= dsilu_f(eo.ein[q].get(e));
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) val[e] *= dgelu_f(eo.ein[q].get(e));
          }
        }
        Vec8<bf16> rv = eo.ein[q];
        if (act_bwd) rv = resid ? load8<bf16>(resid + off) : vec8_zero<bf16>();
        Vec8<bf16> o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o.set(e, val[e] * eo.rsc[q] + rv.get(e));
        store8<bf16>(Cout + off, o);
      }
    }
    if (pass == 0) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
}

template <int NS>
__global__ __launch_bounds__(WS_NT) void gemm_ws_kernel(GemmArgs p) {
  constexpr int BM = WS_BM, BN = WS_BN, BK = WS_BK, ROWB = WS_ROWB, STAGE = WS_STAGE, PR = 8;   // 8 rows per DMA instruction
  constexpr int IPT = (BM + BN) / (PR * WS_PROD);                                               // DMA instructions per producer wave and k-tile: 8
  extern __shared__ __attribute__((aligned(16))) unsigned char ws_smem[];                       // [NS][STAGE] ring | C staging
  float* cbuf = reinterpret_cast<float*>(ws_smem + NS * STAGE);

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool consumer = wave < WS_CONS;
  const int wm = (wave & 7) / 4, wn = wave & 3;
  const int c_ = lane & 15, g_ = lane >> 4;
  const int pw = wave - WS_CONS;                         // producer index 0..3 (meaningless for consumers)

  const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM;
  const int nblk = ntn * ntm;
  const int nk = p.K / BK;
  const int xq = nblk >> 3, xr = nblk & 7;
  const bf16* A = (const bf16*)p.A;
  const bf16* B = (const bf16*)p.B;
  const int lr = lane >> 3, slot = lane & 7;             // DMA lane geometry: 8 rows x 8 chunks of 16 B

  auto tile_of = [&](int did, int& m0, int& n0) {
    const int xcd = did & 7;
    const int lid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (did >> 3);
    n0 = (lid % ntn) * BN;
    m0 = (lid / ntn) * BM;
  };

  if ((int)blockIdx.x >= nblk) return;                   // (never: the grid is at most one workgroup per tile)

  if (!consumer) {
    // ------------------------------------------------------------------ producers: the DMA stream of all this workgroup's tiles
    // pieces 0-3 = A rows [32 pw, 32 pw + 32), pieces 4-7 = B rows [32 pw, 32 pw + 32) of the k-tile (8 rows x 128 B each)
    int did = blockIdx.x;
    int m0, n0;
    tile_of(did, m0, n0);
    const bf16* src[8];
    auto set_src = [&]() {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = pw * 32 + j * 8 + lr;
        src[j] = A + (int64_t)min(m0 + r, p.M - 1) * p.lda + ((slot ^ ws_swz(r)) << 3);
        src[4 + j] = B + (int64_t)min(n0 + r, p.N - 1) * p.ldb + ((slot ^ ws_swz(r)) << 3);
      }
    };
    set_src();
    auto issue = [&](int kt, int stage) __attribute__((always_inline)) {
      unsigned char* sa = ws_smem + stage * STAGE + pw * 32 * ROWB;
      unsigned char* sb = sa + BM * ROWB;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(src[j] + kt * BK), (lds_void_t*)(sa + j * PR * ROWB), 16, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(src[4 + j] + kt * BK), (lds_void_t*)(sb + j * PR * ROWB), 16, 0, 0);
    };
    // stream position of the NEXT k-tile to request: (tile `did`, k-tile `pk`), ring stage `ps`
    int pk = 0, ps = 0;
    bool more = true;
    auto advance = [&]() __attribute__((always_inline)) {
      if (++pk == nk) {
        pk = 0;
        did += gridDim.x;
        more = did < nblk;
        if (more) { tile_of(did, m0, n0); set_src(); }
      }
      ps = ps + 1 == NS ? 0 : ps + 1;
    };
    // prologue: NS - 1 k-tiles in flight, the first one landed
    int n_iss = 0;
#pragma unroll
    for (int i = 0; i < NS - 1; ++i)
      if (more) { issue(pk, ps); advance(); ++n_iss; }
    if (n_iss == NS - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"i"((NS - 2) * IPT) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // a stream shorter than the ring
    __builtin_amdgcn_s_barrier();
    // steady state: one step per k-tile the consumers multiply; the tile's 3 epilogue barriers are walked as well
    for (int cd = blockIdx.x; cd < nblk; cd += gridDim.x) {
      for (int kt = 0; kt < nk; ++kt) {
        // the k-tile the consumers read NEXT must have landed: everything but the youngest NS - 2 tiles -- as long as a
        // new tile went out in this step; once the stream has ended the youngest tile IS the next one: drain
        if (more) {
          issue(pk, ps);
          advance();
          asm volatile("s_waitcnt vmcnt(%0)" ::"i"((NS - 2) * IPT) : "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
      }
      __builtin_amdgcn_s_barrier();                    // epilogue pass 0: staging written
      __builtin_amdgcn_s_barrier();                    // epilogue pass 0: staging read
      __builtin_amdgcn_s_barrier();                    // epilogue pass 1: staging written
    }
    return;
  }

  // -------------------------------------------------------------------- consumers
  __builtin_amdgcn_s_barrier();                          // the producers' prologue
  int cs = 0;                                            // ring stage of the k-tile being multiplied
  for (int did = blockIdx.x; did < nblk; did += gridDim.x) {
    int m0, n0;
    tile_of(did, m0, n0);
    EpiOperands<bf16, BM, BN, 4> eo;                     // bias / residual / z / DropPath scale of THIS tile, requested now
    eo.load(p, m0, n0, wn, c_);
    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < nk; ++kt) {
      const unsigned char* la = ws_smem + cs * STAGE;
      const unsigned char* lb = la + BM * ROWB;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        Vec8<bf16> fa[4], fb[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = wm * (BM / 2) + i * 16 + c_;
          fa[i] = load8<bf16>(reinterpret_cast<const bf16*>(la + r * ROWB + (((ks * 4 + g_) ^ ws_swz(r)) << 4)));
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int r = wn * (BN / 4) + j * 16 + c_;
          fb[j] = load8<bf16>(reinterpret_cast<const bf16*>(lb + r * ROWB + (((ks * 4 + g_) ^ ws_swz(r)) << 4)));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) mma16(fa[i], fb[j], acc[i][j]);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                      // stage cs released; the next k-tile is visible
      cs = cs + 1 == NS ? 0 : cs + 1;
    }
    ws_epilogue(p, acc, cbuf, m0, n0, wm, wn, c_, g_, eo, true);
  }
}

bool gemm_ws_ok(const GemmArgs& a) {
  if (!vtx_opt(VTX_OPT_GEMM_WS)) return false;
  if ((a.K % 64) != 0 || (a.N % 128) != 0) return false;
  const long tiles = (long)(a.N / 128) * ((a.M + 127) / 128);
  return tiles >= 4 * 256;                               // several tiles per persistent workgroup, else the static walk is unbalanced
}

int gemm_ws_launch(const GemmArgs& a, hipStream_t st) {
  constexpr int NS = 3;
  constexpr int smem = NS * WS_STAGE + WS_CBYTES;        // 96 KB ring + 33 KB staging: one workgroup per CU
  auto kern = gemm_ws_kernel<NS>;
  if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) return VTX_ERR_LAUNCH;
  const long tiles = (long)(a.N / 128) * ((a.M + 127) / 128);
  long g = 256;
  if (g > tiles) g = (tiles + 7) / 8 * 8 > tiles ? tiles : (tiles + 7) / 8 * 8;
  hipLaunchKernelGGL(kern, dim3((unsigned)g), dim3(WS_NT), smem, st, a);
  return vtx_check_launch();
}
