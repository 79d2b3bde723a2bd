// Two-group ("ping-pong") GEMM for gfx950: C[M, N] = epilogue(A[M, K] . B[N, K]^T), bf16 in, fp32 accumulate, N % 192 == 0,
// K % 64 == 0 -- every linear layer of ViT-S/16 and of Swin-S stages 2-4 (N = 192, 384, 576, 768, 1152, 1536, 2304, 3072).
// Reference shapes: models/layer.py:186-196 (fc1 / fc2), models/vit.py:23-25,43 and models/swin_transformer.py:128,155 (qkv / proj)
// and their dgrads.
//
// What round 4 left (profiles/round4_astat_gemm.md) and what round 5 measured (profiles/round5_strip_gemm_ablation.md):
//   * the tiled kernels walk barrier -> fragment reads -> MFMAs -> barrier in lockstep: ~45 % MFMA busy inside the loop at best;
//   * the global -> LDS path of a CU retires one REQUEST (a 64- or 128-byte piece of a line) per ~2.5-2.9 cycles whatever its size,
//     requests of different streams add up, and an HBM miss in front of L2 hits of the same wave holds those back (loads return
//     in order per wave): a 128 x 128 x 64 tile is 256 requests per 512 MFMA cycles -- request-bound at 2/3 of the MFMA peak, and
//     a 32-deep ring of 64-byte rows (the first round-5 build, 32 KB per k-step) needs twice the requests per byte.
// This kernel:
//   * tiles of BM x 192 with BM = 32 WMF (WMF = 4 .. 7: 128 .. 224 rows), ONE workgroup of 8 waves per CU: (BM + 192) full
//     128-byte lines per 64-deep k-tile = 0.31 requests per MFMA cycle at WMF = 7 (128 x 128: 0.5);
//   * waves 2 (M) x 4 (N), each 16 WMF rows x 48 columns (WMF x 3 accumulator tiles: 21 MFMAs per 10 fragment reads);
//     waves 0-3 (upper rows) and 4-7 (lower rows) are the two GROUPS -- one wave of each per SIMD -- that run half a k-step apart:
//     while one group multiplies k-step u out of registers, the other reads its fragments of k-step u from LDS, so the matrix pipe
//     of every SIMD always has one wave in its MFMA segment; one s_barrier per phase (two per 32-deep k-step);
//   * operands by LDS-DMA (global_load_lds_dwordx4, 8 rows x 128 B per instruction) into a ring of three 64-deep k-tiles
//     ([rows][128 B], 16-byte chunk q of row r in slot q ^ (r & 7)), requested three k-tiles ahead from inside the MFMA segments;
//     group 0 requests only A (HBM misses), group 1 only B (L2 hits): neither stream queues behind the other's latency; counted
//     vmcnt waits; the fragment reads are inline asm, so hipcc never sees an LDS read next to an outstanding DMA;
//   * the column tiles of one row panel run back to back on ONE XCD (its L2 serves the panel's second .. N/192-th read).
// Element values: the same products in the same k order and the same epilogue expression per element as gemm_glds_pv_kernel /
// gemm_astat_kernel -- bitwise interchangeable (tests/test_gpu_dispatch.py).
#include <stdlib.h>
#include <utility>

#include "gemm_common.h"
#include "options.h"

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

#ifndef PP_AUX_A
#define PP_AUX_A 0          // cache-policy bits of the A requests (group 0): 2 = nt -- A/B: profiles/round6_nt_load_screen.txt
#endif

namespace {

constexpr int PP_NT = 512;                 // 8 waves
constexpr int PP_BN = 192, PP_NF = 3;      // columns per tile, 16-column accumulator tiles per wave
constexpr int PP_NS = 3;                   // ring: 64-deep k-tiles
constexpr int PP_IP = 4;                   // 16-row accumulator tiles per wave staged per epilogue pass

template <int OFF> __device__ __forceinline__ void pp_ds_read16(bf16x8& d, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "i"(OFF) : "memory");
}
template <int N_, int... I>
__device__ __forceinline__ void pp_read_frags(bf16x8 (&f)[N_], unsigned addr, std::integer_sequence<int, I...>) {
  (pp_ds_read16<I * 2048>(f[I], addr), ...);
}
__device__ __forceinline__ void pp_pin1(bf16x8& x) { asm volatile("" : "+v"(x)); }
template <int N_, int... I> __device__ __forceinline__ void pp_pin(bf16x8 (&f)[N_], std::integer_sequence<int, I...>) {
  (pp_pin1(f[I]), ...);
}

template <int WMF, int NF = PP_NF> constexpr int pp_slot_bytes() { return (32 * WMF + 64 * NF) * 128; }
template <int WMF, int NF = PP_NF> constexpr int pp_smem_bytes() { return PP_NS * pp_slot_bytes<WMF, NF>(); }

// logical row -> row of the row-indexed operands (stochastic-depth compaction: GemmArgs::perm)
template <bool MAPPED> __device__ __forceinline__ int pp_orow(const GemmArgs& p, int row, int* smp) {
  if constexpr (!MAPPED) {
    if (smp) *smp = row / p.rows_per_scale;
    return row;
  } else {
    const int s = (int)__umulhi((unsigned)row, p.map_magic), sm = p.perm[s];
    if (smp) *smp = sm;
    return sm * p.map_T + (row - s * p.map_T);
  }
}

struct PpGrid { int ntm, ntn, skew; };   // skew: start delay per column tile, in units of ~1024 cycles (see pp_skew)

// ABL: phase ablation for timing probes (tools/r5/pp_ablate.py; results are garbage, durations are what is measured):
// 1 no MFMAs | 2 no DMA requests inside the loop | 4 no fragment reads | 8 no A requests | 16 no B requests | 32 no barriers.  0 in the product.
// NF: 16-column accumulator tiles per wave = tile width / 64 (3: the 192-column tile everything above describes; round 5 also 2 and 4 --
// N = 128 / 256 / 512 layers of Twins-SVT and PVT that are not multiples of 192 -- as gemm_ppn_kernel<WMF, MAPPED, NF>; 128 registers per
// wave bound WMF x NF: NF = 4 runs with WMF <= 5, and its ring of three (32 WMF + 256)-row k-tiles fills the LDS at WMF = 5)
template <int WMF, bool MAPPED, int ABL, int NF>
__device__ __forceinline__ void pp_body(const GemmArgs& p, const PpGrid& gr) {
  constexpr int BM = 32 * WMF, BN = 64 * NF, WC = 16 * NF, NS = PP_NS;
  constexpr int A_BYTES = BM * 128, SLOT = pp_slot_bytes<WMF, NF>();
  constexpr int NIA = WMF, NIB = BN / 32;                              // DMA instructions per wave and k-tile: A waves | B waves
  constexpr int LPT = NIA > NIB ? NIA : NIB;                           // both kinds issue this many (the shorter list repeats its last piece)
  constexpr int H0 = (LPT + 1) / 2;                                    // requested in the first MFMA segment of a k-tile, the rest in the second
  static_assert(NS * SLOT <= 160 * 1024, "ring must fit the LDS");
  extern __shared__ __attribute__((aligned(16))) unsigned char pp_smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wave >> 2, wn = wave & 3;                            // waves w and w + 4 share a SIMD: the two groups
  const int c_ = lane & 15, g_ = lane >> 4;
  const int rows = MAPPED ? p.Mk : p.M;                                // rows this launch computes
  // block -> tile: XCD x = block % 8 walks its row panels x, x + 8, ... one after the other, all column tiles of a panel back to back
  const int xq = blockIdx.x >> 3, xcd = blockIdx.x & 7;
  const int tm = xcd + 8 * (xq / gr.ntn), tn = xq % gr.ntn;
  if (tm >= gr.ntm) return;
  const int m0 = tm * BM, n0 = tn * BN;
  const int NT = p.K >> 6;                                             // 64-deep k-tiles
  // The column tiles of a row panel read the same A rows; started together they miss the L2 together and the panel crosses the
  // fabric once per column tile.  Column tile tn starts tn * skew * ~0.43 us late, so that it finds its lines in the XCD's L2.
  for (int d = 0; d < tn * gr.skew; ++d) __builtin_amdgcn_s_sleep(16);

  if (MAPPED && m0 >= rows) {
    // copy-only tile of a mapped launch: rows of DROPPED samples (DropPath scale 0): C = resid, no operands touched
    const bf16* __restrict__ rs = (const bf16*)p.resid;
    bf16* __restrict__ cd = (bf16*)p.C;
    if (rs != nullptr)
      for (int v = threadIdx.x; v < BM * (BN / 8); v += PP_NT) {
        const int lrow = m0 + v / (BN / 8), col = n0 + (v % (BN / 8)) * 8;
        if (lrow < p.M) {
          const int64_t off = (int64_t)pp_orow<MAPPED>(p, lrow, nullptr) * p.ldc + col;
          store8<bf16>(cd + off, load8<bf16>(rs + off));
        }
      }
    return;
  }

  // ---- DMA sources: one instruction = 8 rows x 128 B (lane: row lane / 8, chunk (lane % 8) ^ row, i.e. swizzled on the source side)
  // group 0 (waves 0-3): A rows wave * 8 WMF + 8 j + lane / 8;  group 1 (waves 4-7): B rows (wave - 4) * 48 + 8 j + lane / 8
  const int lr = lane >> 3, sw = ((lane & 7) ^ lr) << 3;
  const bf16* src[LPT];
  int dst0;                                                            // LDS offset of this wave's first piece inside a slot
  if (grp == 0) {
#pragma unroll
    for (int j = 0; j < LPT; ++j) {
      const int jj = j < NIA ? j : NIA - 1;
      // rows past the computed ones are never stored: any valid, finite row will do -- the last computed one
      const int r = min(m0 + wave * (8 * WMF) + jj * 8 + lr, rows - 1);
      src[j] = (const bf16*)p.A + (int64_t)pp_orow<MAPPED>(p, r, nullptr) * p.lda + sw;
    }
    dst0 = wave * (8 * WMF) * 128;
  } else {
#pragma unroll
    for (int j = 0; j < LPT; ++j) {
      const int jj = j < NIB ? j : NIB - 1;
      src[j] = (const bf16*)p.B + (int64_t)(n0 + wn * WC + jj * 8 + lr) * p.ldb + sw;
    }
    dst0 = A_BYTES + wn * WC * 128;
  }
  constexpr int NIMIN = NIA < NIB ? NIA : NIB;
  const int ni = grp == 0 ? NIA : NIB;

  // pieces [j0, j1) of this wave's next k-tile -> ring slot sl (the pointers advance once per k-tile, with the last piece)
  auto issue = [&](int sl, int j0, int j1) {
    unsigned char* base = pp_smem + sl * SLOT + dst0;
#pragma unroll
    for (int j = 0; j < LPT; ++j) {
      if (j < j0 || j >= j1) continue;
      const bool mine = (ABL & 8) ? grp != 0 : ((ABL & 16) ? grp == 0 : true);
      const int jj = j < NIMIN ? j : (j < ni ? j : ni - 1);            // (the repeated piece lands where its original does)
      if (!(ABL & 24) || mine) {
        if (PP_AUX_A != 0 && grp == 0) __builtin_amdgcn_global_load_lds((gbl_void_t*)src[j], (lds_void_t*)(base + jj * 1024), 16, 0, PP_AUX_A);
        else __builtin_amdgcn_global_load_lds((gbl_void_t*)src[j], (lds_void_t*)(base + jj * 1024), 16, 0, 0);
      }
      src[j] += 64;
    }
  };

  // ---- fragment addresses: row r of a slot at r * 128, chunk (4 ks + g) ^ (r & 7); + 2048 per 16-row tile as an immediate
  const unsigned lds0 = (unsigned)(size_t)pp_smem;
  const unsigned fro = c_ * 128 + ((g_ ^ (c_ & 7)) << 4);                // k-step 0 of a k-tile; k-step 1: chunk ^ 4, i.e. byte offset ^ 64
  const unsigned fragA[2] = {lds0 + grp * (16 * WMF) * 128 + fro, lds0 + grp * (16 * WMF) * 128 + (fro ^ 64)};
  const unsigned fragB[2] = {lds0 + A_BYTES + wn * WC * 128 + fro, lds0 + A_BYTES + wn * WC * 128 + (fro ^ 64)};

  f32x4 acc[WMF][NF];
#pragma unroll
  for (int i = 0; i < WMF; ++i)
#pragma unroll
    for (int j = 0; j < NF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- prologue: all three slots are free: k-tiles 0, 1, 2
#pragma unroll
  for (int t = 0; t < NS; ++t)
    if (t < NT) issue(t, 0, LPT);
  if (NT >= NS) asm volatile("s_waitcnt vmcnt(%0)" ::"i"((NS - 1) * LPT) : "memory");
  else if (NT == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(LPT) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                                        // k-tile 0 is in LDS
  if (grp == 1) __builtin_amdgcn_s_barrier();                          // group 1 sits out phase 0

  // Phase p: group 0 reads k-step u in phase 2u and multiplies it in phase 2u + 1; group 1 one phase later.  K-tile t (steps 2t,
  // 2t + 1) is last read in phase 4t + 3, so its slot takes k-tile t + 3 from phase 4t + 4 on: group 1 requests it in its MFMA
  // segments of steps 2t + 1 and 2t + 2, group 0 in those of steps 2t + 2 and 2t + 3.  K-tile t + 1 must be visible before phase
  // 4t + 4: both groups wait for their pieces of it in phase 4t + 3, when exactly one younger k-tile (t + 2) of theirs is out.
  int rs = 0;                                                          // ring slot of the k-tile being read
  int is = 0;                                                          // ring slot of the k-tile being requested
  int nxt = NS;                                                        // next k-tile to request
  bf16x8 fa[WMF], fb[NF];
  if constexpr ((ABL & 4) != 0) {
#pragma unroll
    for (int i = 0; i < WMF; ++i) fa[i] = bf16x8{};
#pragma unroll
    for (int j = 0; j < NF; ++j) fb[j] = bf16x8{};
  }
  auto step = [&](auto ks_c, int t) {
    constexpr int ks = decltype(ks_c)::value;
    const int u = 2 * t + ks;
    // ---------------- read segment: fragments of k-step u (the partner group multiplies meanwhile)
    if constexpr (!(ABL & 4)) {
      const unsigned so = (unsigned)(rs * SLOT);
      pp_read_frags(fb, fragB[ks] + so, std::make_integer_sequence<int, NF>{});
      pp_read_frags(fa, fragA[ks] + so, std::make_integer_sequence<int, WMF>{});
    }
    const bool wait_here = ks == 1 && t + 1 < NT;                      // phase 4t + 3: this wave's pieces of k-tile t + 1 have landed
    const bool one_younger = t + 2 < NT;
    if (!(ABL & 2) && grp == 1 && wait_here) {
      if (one_younger) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(LPT) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    pp_pin(fb, std::make_integer_sequence<int, NF>{});
    pp_pin(fa, std::make_integer_sequence<int, WMF>{});
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!(ABL & 32)) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---------------- MFMA segment (the partner group reads its fragments meanwhile)
    if (!(ABL & 2) && nxt < NT) {
      // group 1: first half at odd steps >= 1, second half at even steps >= 2; group 0: first half at even steps >= 2, second at odd >= 3
      const bool first = grp == 1 ? (ks == 1) : (ks == 0 && u >= 2);
      const bool second = grp == 1 ? (ks == 0 && u >= 2) : (ks == 1 && u >= 3);
      if (first) issue(is, 0, H0);
      if (second) {
        issue(is, H0, LPT);
        ++nxt;
        is = is + 1 == NS ? 0 : is + 1;
      }
    }
    __builtin_amdgcn_s_setprio(1);
    if constexpr (!(ABL & 1)) {
#pragma unroll
      for (int i = 0; i < WMF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    if (!(ABL & 2) && grp == 0 && wait_here) {
      if (one_younger) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(LPT) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!(ABL & 32)) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int t = 0; t < NT; ++t) {
    step(std::integral_constant<int, 0>{}, t);
    step(std::integral_constant<int, 1>{}, t);
    rs = rs + 1 == NS ? 0 : rs + 1;
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();                          // group 0 sits out the last phase
  // (no DMA is outstanding and nobody reads the ring any more: it becomes the staging buffer of the epilogue)

  // ---------------- epilogue: PP_IP x 16 rows per wave and pass through LDS as fp32 (acc + bias), stored as whole 384-byte rows
  // acc[i][j][r] = C[m0 + grp 16 WMF + 16 i + 4 g + r][n0 + wn 48 + 16 j + c]
  constexpr int CSTR = BN + 4;                                         // floats per staged row
  constexpr int VROW = BN / 8;                                         // 8-element vectors per row
  static_assert(2 * PP_IP * 16 * CSTR * 4 <= NS * SLOT, "epilogue staging must fit the ring");
  float* cbuf = reinterpret_cast<float*>(pp_smem);
  bf16* __restrict__ Cout = (bf16*)p.C;
  const bf16* __restrict__ resid = (const bf16*)p.resid;
  const bf16* __restrict__ aux_in = (const bf16*)p.aux_in;
  bf16* __restrict__ aux_out = (bf16*)p.aux_out;
  const bool act_fwd = p.act == 1 || p.act == 3, act_bwd = p.act == 2 || p.act == 4;
  float bcol[NF];
#pragma unroll
  for (int j = 0; j < NF; ++j) bcol[j] = p.bias ? p.bias[n0 + wn * WC + j * 16 + c_] : 0.f;

#pragma unroll
  for (int i0 = 0; i0 < WMF; i0 += PP_IP) {
    const int ni2 = WMF - i0 < PP_IP ? WMF - i0 : PP_IP;               // accumulator tiles per wave in this pass
    if (i0) __syncthreads();                                           // the previous pass has been read
#pragma unroll
    for (int ii = 0; ii < PP_IP; ++ii) {
      if (i0 + ii < WMF) {
#pragma unroll
        for (int j = 0; j < NF; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            cbuf[((grp * PP_IP + ii) * 16 + g_ * 4 + r) * CSTR + wn * WC + j * 16 + c_] = acc[i0 + ii < WMF ? i0 + ii : 0][j][r] + bcol[j];
      }
    }
    __syncthreads();
    const int nvec = 2 * PP_IP * 16 * VROW;
    for (int v = threadIdx.x; v < nvec; v += PP_NT) {
      const int sr = v / VROW, cv = v - sr * VROW;                     // staged row, vector of the row
      const int hg = sr / (PP_IP * 16), ii = (sr >> 4) % PP_IP;
      if (ii >= ni2) continue;
      const int lrow = m0 + hg * (16 * WMF) + (i0 + ii) * 16 + (sr & 15);
      if (lrow >= p.M) continue;
      int srow;
      const int64_t off = (int64_t)pp_orow<MAPPED>(p, lrow, &srow) * p.ldc + n0 + cv * 8;
      if (MAPPED && lrow >= rows) {                                    // a dropped sample's row inside the last computed tile
        if (resid) store8<bf16>(Cout + off, load8<bf16>(resid + off));
        continue;
      }
      const float* cp = cbuf + sr * CSTR + cv * 8;
      f32x4 lo = *reinterpret_cast<const f32x4*>(cp), hi = *reinterpret_cast<const f32x4*>(cp + 4);
      float val[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      if (act_fwd) {
        Vec8<bf16> z;
#pragma unroll
        for (int e = 0; e < 8; ++e) z.set(e, val[e]);                 // activation of the ROUNDED pre-activation (what the backward sees)
        if (p.act == 1) {
#pragma unroll
          for (int e = 0; e < 8; ++e) val[e] = silu_f(z.get(e));
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) val[e] = gelu_f(z.get(e));
        }
        if (aux_out) store8<bf16>(aux_out + off, z);
      } else if (act_bwd) {
        const Vec8<bf16> zin = load8<bf16>(aux_in + off);
        if (p.act == 2) {
#pragma unroll
          for (int e = 0; e < 8; ++e) val[e] *= dsilu_f(zin.get(e));
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) val[e] *= dgelu_f(zin.get(e));
        }
      }
      const Vec8<bf16> rv = resid ? load8<bf16>(resid + off) : vec8_zero<bf16>();
      const float rsc = p.rowscale ? p.rowscale[srow] : 1.f;
      Vec8<bf16> o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o.set(e, val[e] * rsc + rv.get(e));
      store8<bf16>(Cout + off, o);
    }
  }
}

template <int WMF, bool MAPPED, int ABL = 0>
__global__ __launch_bounds__(PP_NT, 2) void gemm_pp_kernel(GemmArgs p, PpGrid gr) { pp_body<WMF, MAPPED, ABL, PP_NF>(p, gr); }
template <int WMF, bool MAPPED, int NF>
__global__ __launch_bounds__(PP_NT, 2) void gemm_ppn_kernel(GemmArgs p, PpGrid gr) { pp_body<WMF, MAPPED, 0, NF>(p, gr); }

int pp_skew() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("VTX_PP_SKEW"); v = e ? atoi(e) : 0; }
  return v;
}

template <int WMF, bool MAPPED, int ABL = 0, int NF = PP_NF> int pp_launch_k(const GemmArgs& a, hipStream_t st) {
  constexpr int smem = pp_smem_bytes<WMF, NF>();
  void (*kern)(GemmArgs, PpGrid);
  if constexpr (NF == PP_NF) kern = gemm_pp_kernel<WMF, MAPPED, ABL>;
  else kern = gemm_ppn_kernel<WMF, MAPPED, NF>;
  if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) return VTX_ERR_LAUNCH;
  // (a mapped launch with dropped samples and a residual: copy-only tiles behind the computed ones)
  const int rows = (MAPPED && (a.Mk == a.M || a.resid == nullptr)) ? a.Mk : a.M;
  PpGrid gr;
  gr.ntm = (rows + 32 * WMF - 1) / (32 * WMF);
  gr.ntn = a.N / (64 * NF);
  gr.skew = pp_skew();
  const int blocks = 8 * ((gr.ntm + 7) / 8) * gr.ntn;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(PP_NT), smem, st, a, gr);
  return vtx_check_launch();
}
template <int WMF, int NF = PP_NF> int pp_launch_m(const GemmArgs& a, hipStream_t st) {
  return a.perm != nullptr ? pp_launch_k<WMF, true, 0, NF>(a, st) : pp_launch_k<WMF, false, 0, NF>(a, st);
}

// Tile height: the fewest rounds of one-workgroup-per-CU tiles, then the smallest tiles that still fit that many rounds.
int pp_pick_wmf(long rows, int ntn, int wmax = 7) {
  const int cus = vtx_cu_count_cached();
  int best = wmax;
  long best_cost = 1L << 60;
  for (int w = wmax; w >= 4; --w) {
    const long tiles = (rows + 32 * w - 1) / (32 * w) * ntn;
    const long rounds = (tiles + cus - 1) / cus;
    const long cost = rounds * (w + 2);                                // (+2: prologue / epilogue of a tile in 32-row units)
    if (cost < best_cost) { best_cost = cost; best = w; }
  }
  return best;
}

}  // namespace

// accumulator tiles per wave for N columns: 3 (192-column tiles) where N allows, else 4 (256) or 2 (128); 0: not this kernel's shape
static int pp_nf(int N) { return N % 192 == 0 ? 3 : (N % 256 == 0 ? 4 : (N % 128 == 0 ? 2 : 0)); }
static int pp_wmax(int nf) { return nf == 4 ? 5 : 7; }

bool gemm_pp_ok(const GemmArgs& a) {
  const int mode = vtx_opt(VTX_OPT_GEMM_PP);
  if (mode == 0) return false;
  const int nf = pp_nf(a.N);
  if (nf == 0 || a.K % 64 != 0) return false;
  if (nf != PP_NF) {
    // 128- / 256-column tiles (round 5; tools/r5/tail_shapes.py, profiles/round5_tail_gemm_shapes.txt): only the N = 128, K = 1024 layers
    // of PVT-Small stage 2 gain (fc2 forward 64 -> 59 us, fc1 dgrad 63 -> 55 us); 256-column tiles lose everywhere they were tried
    // (25 088 x 256 x 1024: 157 tiles of 160 x 256 on 256 CUs, 26.6 vs 21.6 us) and stay behind GEMM_PP = 2
    if (mode < 2 && !(nf == 2 && a.K >= 1024)) return false;
    if ((a.lda % 8) || (a.ldb % 8) || (a.ldc % 8)) return false;
    if (a.kscale != nullptr || a.ksum_out != nullptr) return false;
    if ((a.act == 2 || a.act == 4) && a.aux_in == nullptr) return false;
    const long rows_ = a.perm != nullptr ? a.Mk : a.M;
    if (rows_ <= 0) return false;
    if (mode < 2 && (rows_ + 127) / 128 * (a.N / (64 * nf)) < (3L * vtx_cu_count_cached()) / 4) return false;
    if (a.perm != nullptr) {
      if (a.map_T <= 0 || a.M % a.map_T != 0) return false;
      if (a.rowscale != nullptr && a.rows_per_scale != a.map_T) return false;
      if (a.Mk < a.M && a.resid == nullptr) return false;
    }
    return true;
  }
  // where it wins (profiles/round5_gemm_pp_microbench.txt): long contractions -- K >= 1152, or K >= 768 under narrow outputs (N <= 384);
  // the K <= 384 layers stay on the A-stationary kernel, K = 768 with N >= 768 (Swin stage 4 qkv / fc1 / fc2-dgrad) on the tiled one;
  // one column tile (N = 192: Swin stage-2 qkv dgrad K = 576, patch merging 1 -> 2 K = 384) from K = 384 up (tools/r5/tail_shapes.py:
  // 45.9 vs 55.2 us, 36.8 vs 42.3 us; K = 192 loses, 26.4 vs 22.4 us)
  if (mode < 2 && !(a.K >= 1152 || (a.K >= 768 && a.N <= 384) || (a.K >= 384 && a.N == PP_BN))) return false;
  if ((a.lda % 8) || (a.ldb % 8) || (a.ldc % 8)) return false;
  if (a.kscale != nullptr || a.ksum_out != nullptr) return false;
  if ((a.act == 2 || a.act == 4) && a.aux_in == nullptr) return false;
  const long rows = a.perm != nullptr ? a.Mk : a.M;
  if (rows <= 0) return false;
  // under a CU-filling round of the smallest tiles: the tiled kernels' job
  if (mode < 2 && (rows + 127) / 128 * (a.N / PP_BN) < (3L * vtx_cu_count_cached()) / 4) return false;
  if (a.perm != nullptr) {
    if (a.map_T <= 0 || a.M % a.map_T != 0) return false;
    if (a.rowscale != nullptr && a.rows_per_scale != a.map_T) return false;
    if (a.Mk < a.M && a.resid == nullptr) return false;
  }
  return true;
}

int gemm_pp_launch(const GemmArgs& a, hipStream_t st) {
  const int mode = vtx_opt(VTX_OPT_GEMM_PP);
  const int nf = pp_nf(a.N);
  if (nf != PP_NF) {
    int w = pp_pick_wmf(a.perm != nullptr ? a.Mk : a.M, a.N / (64 * nf), pp_wmax(nf));
    if (mode >= 100 && mode < 1000) w = mode % 10 > pp_wmax(nf) ? pp_wmax(nf) : mode % 10;
    if (nf == 4) return w == 4 ? pp_launch_m<4, 4>(a, st) : pp_launch_m<5, 4>(a, st);
    switch (w) {
      case 4: return pp_launch_m<4, 2>(a, st);
      case 5: return pp_launch_m<5, 2>(a, st);
      case 6: return pp_launch_m<6, 2>(a, st);
      case 7: return pp_launch_m<7, 2>(a, st);
      default: return VTX_ERR_SHAPE;
    }
  }
  int wmf = pp_pick_wmf(a.perm != nullptr ? a.Mk : a.M, a.N / PP_BN);
#ifdef VTX_PP_ABLATE
  if (mode >= 1000 && a.perm == nullptr) {                             // timing probes: mode = 1000 * ablation bits (WMF 7)
    switch (mode / 1000) {
      case 1: return pp_launch_k<7, false, 1>(a, st);
      case 2: return pp_launch_k<7, false, 2>(a, st);
      case 3: return pp_launch_k<7, false, 3>(a, st);
      case 4: return pp_launch_k<7, false, 4>(a, st);
      case 5: return pp_launch_k<7, false, 5>(a, st);
      case 6: return pp_launch_k<7, false, 6>(a, st);
      case 7: return pp_launch_k<7, false, 7>(a, st);
      case 13: return pp_launch_k<7, false, 13>(a, st);
      case 21: return pp_launch_k<7, false, 21>(a, st);
      case 34: return pp_launch_k<7, false, 34>(a, st);
      default: return VTX_ERR_SHAPE;
    }
  }
#endif
  if (mode >= 100 && mode < 1000) wmf = mode % 10;                     // forced tile height (tools / tests): 10W -> 32 W rows
  switch (wmf) {
    case 4: return pp_launch_m<4>(a, st);
    case 5: return pp_launch_m<5>(a, st);
    case 6: return pp_launch_m<6>(a, st);
    case 7: return pp_launch_m<7>(a, st);
    default: return VTX_ERR_SHAPE;
  }
}
