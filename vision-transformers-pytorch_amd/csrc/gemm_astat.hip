// A-STATIONARY LDS-DMA GEMM for gfx950 (bf16): C[M,N] = epilogue(A[M,K] . W[N,K]^T) with a SHORT contraction K <= 384
// (Swin-S stage 3 / ViT-S/16: qkv, fc1, proj forward; proj and fc2 dgrad) and N a multiple of 128.
//
// What the counters say about the tiled kernel on these shapes (profiles/round4_pmc_sq_gemm_in_model.txt): no unit is busy
// (MFMA ~18 %, LDS ~18 %, TA ~37 %, HBM 0.3), the waves are parked -- a 128 x 128 tile with K = 384 is six k-tiles between a
// prologue that waits for the first operands and an epilogue, every one of its k-tiles waits for the ONE k-tile of LDS-DMA the
// 2-stage ring keeps in flight, and the A rows are fetched again for every column tile.  In the model a round of resident
// workgroups takes 10-14 us for 2.6 us of MFMA work.  Here
//   * ONE workgroup per CU (8 waves) owns a 128-row strip of A for ALL column tiles: the strip's K x 128 panel (<= 96 KB) is
//     copied into LDS once (every k-tile of it requested up front: one latency) and stays there;
//   * the weight streams through a ring of NSW 16-KB stages as ONE continuous sequence of k-tiles over all column tiles:
//     no pipeline drain between column tiles, no per-tile prologue; counted s_waitcnt vmcnt + one raw s_barrier per k-tile;
//   * the product is taken TRANSPOSED (W rows as the MFMA A operand, in the row order of gemm_skinny.hip, applied on the DMA
//     SOURCE address so that the LDS image keeps the conflict-free fragment layout): after two 16 x 16 tiles a lane holds 8
//     CONSECUTIVE output columns of one row -- the epilogue works on 16-byte vectors straight from the accumulators: no LDS
//     staging, no epilogue barrier, bias from an LDS table, residual / z vectors by inline-asm loads (hipcc would drain the
//     whole DMA ring with vmcnt(0) in front of the first use of an ordinary load);
//   * the stores of column tile j drain while the k-tiles of tile j + 1 are multiplied (they retire in order behind the ring's
//     requests: the waits in front of the next tiles' operands count them in).
// Same element-wise epilogue expressions and the same k order inside the MFMA as gemm_glds.hip: bit-identical outputs.
// Stochastic-depth compaction (GemmArgs::perm) as in gemm_glds_pv_kernel: the strip's rows go through the tile's four sample
// scalars; strips of dropped samples are copy-only.
#include "gemm_common.h"
#include "options.h"

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

namespace {

constexpr int AS_BM = 128, AS_BN = 128, AS_BK = 64;
constexpr int AS_KT_BYTES = AS_BM * AS_BK * 2;          // 16 KB: one k-tile image of the A strip = one stage of the W ring

// logical row -> row of the operands for the rows of ONE 128-row strip (GemmArgs::perm; see TileRowMap in gemm_glds.hip)
template <bool MAPPED> struct StripRowMap {
  int s0, sp0, sp1, sp2, sp3;
  __device__ __forceinline__ void init(const GemmArgs& p, int m0) {
    if constexpr (MAPPED) {
      s0 = (int)__umulhi((unsigned)m0, p.map_magic);
      const int last = p.M / p.map_T - 1;
      sp0 = __builtin_amdgcn_readfirstlane(p.perm[min(s0, last)]);
      sp1 = __builtin_amdgcn_readfirstlane(p.perm[min(s0 + 1, last)]);
      sp2 = __builtin_amdgcn_readfirstlane(p.perm[min(s0 + 2, last)]);
      sp3 = __builtin_amdgcn_readfirstlane(p.perm[min(s0 + 3, last)]);
    }
  }
  __device__ __forceinline__ int orow(const GemmArgs& p, int row, int* smp) const {
    if constexpr (!MAPPED) {
      if (smp) *smp = row / p.rows_per_scale;
      return row;
    } else {
      const int s = (int)__umulhi((unsigned)row, p.map_magic), d = s - s0;
      const int lo = d <= 0 ? sp0 : sp1, hi = d == 2 ? sp2 : sp3;
      const int sm = d <= 1 ? lo : hi;
      if (smp) *smp = sm;
      return sm * p.map_T + (row - s * p.map_T);
    }
  }
};

}  // namespace

// NKT = K / 64 k-tiles of the strip resident in LDS; NSW ring stages for the weight stream
// VEC: the epilogue reads one 16-byte vector per output vector (the residual, or z for act'); AUX: it writes z (act forward)
template <int NKT, int NSW, bool MAPPED, bool VEC, bool AUX>
__global__ __launch_bounds__(512) void gemm_astat_kernel(GemmArgs p) {
  constexpr int ROWB = 128, L = 2;                     // bytes per LDS row; DMA instructions per wave and 16-KB k-tile
  constexpr int NST = 4 * (AUX ? 2 : 1);               // 16-byte store instructions per wave and column tile
  extern __shared__ __attribute__((aligned(16))) unsigned char as_smem[];
  unsigned char* const sa = as_smem;                                  // [NKT][128 rows][128 B]
  unsigned char* const sw = as_smem + NKT * AS_KT_BYTES;              // [NSW][128 rows][128 B]
  float* const sbias = reinterpret_cast<float*>(as_smem + (NKT + NSW) * AS_KT_BYTES);   // [N]

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;             // 4 x 2 waves of 32 x 64
  const int c_ = lane & 15, g_ = lane >> 4;
  const int m0 = blockIdx.x * AS_BM;
  const int live_rows = MAPPED ? p.Mk : p.M;
  const int ntn = p.N / AS_BN;

  StripRowMap<MAPPED> rmap;
  rmap.init(p, m0);
  bf16* __restrict__ Cout = (bf16*)p.C;
  const bf16* __restrict__ resid = (const bf16*)p.resid;

  if (MAPPED && m0 >= live_rows) {
    // copy-only strip of a mapped launch: the rows of DROPPED samples: C = resid, no operands touched
    if (resid != nullptr) {
      const int vrow = p.N >> 3;
      for (int v = threadIdx.x; v < AS_BM * vrow; v += 512) {
        const int lr = v / vrow, cv = v - lr * vrow;
        if (m0 + lr < p.M) {
          const int64_t off = (int64_t)rmap.orow(p, m0 + lr, nullptr) * p.ldc + cv * 8;
          store8<bf16>(Cout + off, load8<bf16>(resid + off));
        }
      }
    }
    return;
  }

  // ---- rows this lane finishes: row(i) = m0 + 32 wm + 16 i + c_ (both 16-row tiles of the wave), columns 8 g_ .. 8 g_ + 7 of each
  // 32-column pair.  DropPath scale per row, requested before any DMA (ordinary loads: nothing is in flight yet)
  int orow_[2];
  float rsc[2];
  bool rok[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int lrow = m0 + wm * 32 + i * 16 + c_;
    rok[i] = lrow < p.M;
    int smp;
    orow_[i] = rmap.orow(p, rok[i] ? lrow : m0, &smp);
    rsc[i] = (rok[i] && p.rowscale) ? p.rowscale[smp] : 1.f;
  }
  for (int i = threadIdx.x; i < p.N; i += 512) sbias[i] = p.bias ? p.bias[i] : 0.f;
  // every ordinary load above must have RETURNED before the first DMA request: consumed inside the k-tile loop, hipcc would
  // otherwise put s_waitcnt vmcnt(0) -- the whole ring -- in front of that use in every epilogue
  asm volatile("s_waitcnt lgkmcnt(0)" ::"v"(rsc[0]), "v"(rsc[1]), "v"(orow_[0]), "v"(orow_[1]) : "memory");
  const bool full_strip = m0 + AS_BM <= p.M;            // every lane issues every store: the counted waits may rely on NST

  // ---- DMA source pointers.  A: LDS row r <- strip row r (row map applied; rows past the computed ones read the last one).
  // W: LDS row R of a 128-column tile <- weight row 32 (R >> 5) + 8 ((R & 15) >> 2) + 4 ((R >> 4) & 1) + (R & 3): the operand-row
  // order of the transposed product (see the header); XOR swizzle of the 16-byte chunk by the LDS row on both.
  const int lr = lane >> 3, slot = lane & 7;
  const bf16* asrc[2];
  const bf16* wsrc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = wave * 16 + j * 8 + lr;
    asrc[j] = (const bf16*)p.A + (int64_t)rmap.orow(p, min(m0 + r, live_rows - 1), nullptr) * p.lda + ((slot ^ (r & 7)) << 3);
    const int n = 32 * (r >> 5) + 8 * ((r & 15) >> 2) + 4 * ((r >> 4) & 1) + (r & 3);
    wsrc[j] = (const bf16*)p.B + (int64_t)n * p.ldb + ((slot ^ (r & 7)) << 3);
  }
  const int64_t wtile = (int64_t)AS_BN * p.ldb;         // elements between column tiles of W
  auto issue_a = [&](int kt) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(asrc[j] + kt * AS_BK), (lds_void_t*)(sa + kt * AS_KT_BYTES + (wave * 16 + j * 8) * ROWB), 16, 0, 0);
  };
  auto issue_w = [&](int tn, int kt, int stage) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(wsrc[j] + tn * wtile + kt * AS_BK), (lds_void_t*)(sw + stage * AS_KT_BYTES + (wave * 16 + j * 8) * ROWB), 16, 0, 0);
  };

  // ---- prologue: the whole A strip and the first NSW - 1 weight k-tiles, in the order they are needed
  const int T = ntn * NKT;                              // k-tiles of the weight stream
  issue_a(0);
  issue_w(0, 0, 0);
#pragma unroll
  for (int kt = 1; kt < NKT; ++kt) {
    issue_a(kt);
    if (kt < NSW - 1) issue_w(kt / NKT, kt % NKT, kt);
  }
#pragma unroll
  for (int t = NKT; t < NSW - 1; ++t)                   // (NSW - 1 > NKT only for very short contractions)
    if (t < T) issue_w(t / NKT, t % NKT, t);
  // Requests in flight now, oldest first: A0 W0 A1 [W1] A2 [W2] ... ; k-tile t of the loop below needs A(t) (t < NKT) and W(t)

  f32x4 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  int tn = 0, kt = 0, stage = 0;
  for (int t = 0; t < T; ++t) {
    // ---- wait for the operands of k-tile t.  vmcnt retires in order; what may stay in flight is exactly what was requested
    // AFTER them.  First pass (t < NKT): the A k-tiles and weight k-tiles requested behind A(t) / W(t) in the prologue.
    // Steady state: weight k-tiles t + 1 .. t + NSW - 2 (L requests each) and, in the NSW - 1 iterations after an epilogue, that
    // epilogue's NST stores.  A partial last strip (some lanes store nothing) waits for everything instead.
    static_assert(NSW == 3 && NKT >= 2, "the counted waits below are written out for a 3-stage ring");
    if (t == 0) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"i"(L * (NKT - 1) + L) : "memory");          // behind W(0): A(1..NKT-1) and W(1)
    } else if (t == T - 1) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                  // nothing was requested behind the last k-tile
    } else if (full_strip && t >= NKT && kt < NSW - 1) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"i"(L + NST) : "memory");                    // W(t + 1) and the last epilogue's stores
    } else {
      asm volatile("s_waitcnt vmcnt(%0)" ::"i"(L) : "memory");                          // W(t + 1) (1 <= t < NKT: the A k-tiles too, once)
    }
    __builtin_amdgcn_s_barrier();
    // the stage that k-tile t - 1 occupied is free (every wave is past its reads): request k-tile t + NSW - 1 into it
    if (t + NSW - 1 < T) {
      int tnq = tn, ktq = kt + NSW - 1;
      if (ktq >= NKT) { ktq -= NKT; tnq += 1; }
      issue_w(tnq, ktq, stage == 0 ? NSW - 1 : stage - 1);
    }
    const unsigned char* la = sa + kt * AS_KT_BYTES;
    const unsigned char* lb = sw + stage * AS_KT_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      Vec8<bf16> fa[2], fb[4];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = wm * 32 + i * 16 + c_;
        fa[i] = load8<bf16>(reinterpret_cast<const bf16*>(la + r * ROWB + (((ks * 4 + g_) ^ (r & 7)) << 4)));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = wn * 64 + j * 16 + c_;
        fb[j] = load8<bf16>(reinterpret_cast<const bf16*>(lb + r * ROWB + (((ks * 4 + g_) ^ (r & 7)) << 4)));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) mma16(fb[j], fa[i], acc[i][j]);          // TRANSPOSED: W rows x A rows
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    if (kt == NKT - 1) {
      // ---- epilogue of column tile tn, straight from the accumulators:
      //   acc[i][2 pr + h][r] = (A . W^T)[row(i)][n0 + 64 wn + 32 pr + 8 g_ + 4 h + r]
      const int n0 = tn * AS_BN;
      bf16x8 ev[2][2];
      if constexpr (VEC) {
        const bf16* __restrict__ vsrc = (p.act == 2 || p.act == 4) ? (const bf16*)p.aux_in : resid;
        const bf16* q00 = vsrc + (int64_t)orow_[0] * p.ldc + n0 + wn * 64 + g_ * 8;
        const bf16* q10 = vsrc + (int64_t)orow_[1] * p.ldc + n0 + wn * 64 + g_ * 8;
        // four 16-byte loads hipcc does not see (it would wait vmcnt(0) for an ordinary load AND not know about it in the counted
        // waits above) + their wait, in ONE asm block: the results are valid when it ends.  (v1: vmcnt(0) also waits for the
        // ring's youngest request, one k-tile old.)
        asm volatile(
            "global_load_dwordx4 %0, %4, off\n\t"
            "global_load_dwordx4 %1, %4, off offset:64\n\t"
            "global_load_dwordx4 %2, %5, off\n\t"
            "global_load_dwordx4 %3, %5, off offset:64\n\t"
            "s_waitcnt vmcnt(0)"
            : "=&v"(ev[0][0]), "=&v"(ev[0][1]), "=&v"(ev[1][0]), "=&v"(ev[1][1])
            : "v"(q00), "v"(q10)
            : "memory");
      }
      const bool act_fwd = p.act == 1 || p.act == 3, act_bwd = p.act == 2 || p.act == 4;
      bf16* __restrict__ aux_out = (bf16*)p.aux_out;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          const int col = n0 + wn * 64 + pr * 32 + g_ * 8;
          const f32x4 b0 = *reinterpret_cast<const f32x4*>(sbias + col), b1 = *reinterpret_cast<const f32x4*>(sbias + col + 4);
          float val[8];
#pragma unroll
          for (int r = 0; r < 4; ++r) { val[r] = acc[i][2 * pr][r] + b0[r]; val[4 + r] = acc[i][2 * pr + 1][r] + b1[r]; }
          const int64_t off = (int64_t)orow_[i] * p.ldc + col;
          Vec8<bf16> evv;
          if constexpr (VEC) evv.v = ev[i][pr]; else evv = vec8_zero<bf16>();
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
            if constexpr (AUX) { if (rok[i]) store8<bf16>(aux_out + off, z); }
          } else if (act_bwd) {
            if (p.act == 2) {
#pragma unroll
              for (int e = 0; e < 8; ++e) val[e] *= dsilu_f(evv.get(e));
            } else {
#pragma unroll
              for (int e = 0; e < 8; ++e) val[e] *= dgelu_f(evv.get(e));
            }
          }
          Vec8<bf16> rv = evv;
          if (act_bwd) rv = vec8_zero<bf16>();             // (a residual next to act' is not routed here: astat_ok)
          Vec8<bf16> o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o.set(e, val[e] * rsc[i] + rv.get(e));
          if (rok[i]) store8<bf16>(Cout + off, o);
          acc[i][2 * pr] = f32x4{0.f, 0.f, 0.f, 0.f};
          acc[i][2 * pr + 1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
      kt = 0;
      tn += 1;
    } else {
      kt += 1;
    }
    stage = stage + 1 == NSW ? 0 : stage + 1;
  }
}

// ------------------------------------------------------------------------------------------------------------------ host side
static constexpr int AS_NSW = 3;
static size_t astat_smem(int N, int K) { return (size_t)(K / 64 + AS_NSW) * AS_KT_BYTES + (size_t)N * 4; }

bool gemm_astat_ok(const GemmArgs& a) {
  const int mode = vtx_opt(VTX_OPT_GEMM_ASTAT);
  if (mode == 0) return false;
  if (a.K % 64 != 0 || a.K < 128 || a.K > 384 || a.N % 128 != 0 || a.N > 4096) return false;
  if ((a.lda % 8) || (a.ldb % 8) || (a.ldc % 8)) return false;
  if (a.kscale != nullptr || a.ksum_out != nullptr) return false;
  if ((a.act == 2 || a.act == 4) && a.resid != nullptr) return false;     // (no hot-path launch has both)
  if ((a.act == 1 || a.act == 3) && a.resid != nullptr) return false;
  const long rows = a.perm != nullptr ? a.Mk : a.M;
  if (rows < 128 * 96 && mode != 2) return false;                          // fewer strips than ~3/8 of the CUs: the tiled kernels' job
  if (a.perm != nullptr && (3 * a.map_T < 126 || (a.rowscale != nullptr && a.rows_per_scale != a.map_T))) return false;
  if (a.perm != nullptr && a.Mk < a.M && a.resid == nullptr) return false;
  return astat_smem(a.N, a.K) <= 160 * 1024;
}

template <int NKT, bool MAPPED, bool VEC, bool AUX> static int astat_launch_k(const GemmArgs& a, hipStream_t st) {
  const size_t smem = astat_smem(a.N, a.K);
  auto kern = gemm_astat_kernel<NKT, AS_NSW, MAPPED, VEC, AUX>;
  if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return VTX_ERR_LAUNCH;
  hipLaunchKernelGGL(kern, dim3((a.M + AS_BM - 1) / AS_BM), dim3(512), smem, st, a);
  return vtx_check_launch();
}
template <int NKT, bool MAPPED> static int astat_launch_m(const GemmArgs& a, hipStream_t st) {
  const bool vec = a.resid != nullptr || a.act == 2 || a.act == 4;
  const bool aux = (a.act == 1 || a.act == 3) && a.aux_out != nullptr;
  if (vec) return astat_launch_k<NKT, MAPPED, true, false>(a, st);
  if (aux) return astat_launch_k<NKT, MAPPED, false, true>(a, st);
  return astat_launch_k<NKT, MAPPED, false, false>(a, st);
}
template <int NKT> static int astat_launch_n(const GemmArgs& a, hipStream_t st) {
  return a.perm != nullptr ? astat_launch_m<NKT, true>(a, st) : astat_launch_m<NKT, false>(a, st);
}

int gemm_astat_launch(const GemmArgs& a, hipStream_t st) {
  switch (a.K / 64) {
    case 2: return astat_launch_n<2>(a, st);
    case 3: return astat_launch_n<3>(a, st);
    case 4: return astat_launch_n<4>(a, st);
    case 5: return astat_launch_n<5>(a, st);
    case 6: return astat_launch_n<6>(a, st);
    default: return VTX_ERR_SHAPE;
  }
}
