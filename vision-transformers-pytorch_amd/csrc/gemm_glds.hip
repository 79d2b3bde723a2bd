// LDS-DMA GEMM for gfx950: C[M,N] = epilogue(A[M,K] . B[N,K]^T), bf16 in, fp32 accumulate, K % 64 == 0 (or K % 32 == 0 with N % 128 == 0).
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
#include "options.h"

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

// swizzle of the 16-byte chunk index of LDS row r (applied on the DMA source and on the fragment read)
template <int BK> __device__ __forceinline__ int glds_swz(int r) {
  if constexpr (BK == 64) return r & 7;                       // 8 chunks per 128-byte row
  else return (0x1230 >> (((r >> 2) & 3) << 2)) & 3;          // 4 chunks per 64-byte row: {0,3,2,1}[(r>>2)&3]
}

// NS-stage LDS ring: the DMA of k-tile kt+NS-1 is issued while tile kt is multiplied; a counted
// s_waitcnt vmcnt(N) + raw s_barrier (never __syncthreads, which would drain the DMA queue) lets NS-2
// tiles stay in flight across the barrier.
// NWN waves along N: 2 (2 x 2 waves, 256 threads) or 4 (2 x 4 waves of BM/2 x BN/4, 512 threads: half the DMA requests and
// MFMAs per wave and k-tile, twice the waves per SIMD to interleave them; 64 x 128 x 64 tiles only).
template <int BM, int BN, int BK, int NS, int NWN = 2>
__global__ __launch_bounds__(128 * NWN) void gemm_glds_kernel(GemmArgs p) {
  constexpr int ROWB = BK * 2;                           // bytes per LDS row
  constexpr int CPR = BK / 8;                            // 16-byte chunks per row
  constexpr int PR = 1024 / ROWB;                        // rows per DMA instruction (1 KB per wave-instruction)
  constexpr int KS = BK / 32;                            // mma16 k-steps per tile
  constexpr int NWV = 2 * NWN;                           // waves
  constexpr int WM = BM / 32, WN = BN / (16 * NWN);      // 16 x 16 tiles per wave along M / N
  constexpr int STAGE = (BM + BN) * ROWB;                // bytes per pipeline stage
  static_assert(NS * STAGE >= (BM / 2) * (BN + 4) * 4, "C staging must fit");
  extern __shared__ __attribute__((aligned(16))) unsigned char glds_smem[];   // [NS][STAGE]

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / NWN, wn = wave % NWN;
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

  VTX_TRACE(0);
  EpiOperands<bf16, BM, BN, NWN> eo;               // bias / residual / z / DropPath scale: requested before the first DMA
  eo.load(p, m0, n0, wn, c_);

  // per-lane source pointers of this wave's DMA pieces (PR rows x ROWB bytes per instruction)
  const int lr = lane / CPR, slot = lane % CPR;
  constexpr int APW = BM / (NWV * PR), BPW = BN / (NWV * PR);   // pieces per wave for A / B
  static_assert(APW * NWV * PR == BM && BPW * NWV * PR == BN, "whole DMA pieces per wave");
  constexpr int LPT = APW + BPW;                               // DMA instructions per wave per k-tile
  const bf16* asrc[APW];
  const bf16* bsrc[BPW];
#pragma unroll
  for (int j = 0; j < APW; ++j) {
    const int r = wave * (BM / NWV) + j * PR + lr;
    const int row = min(m0 + r, p.M - 1);                // rows past M are never stored: any valid address will do
    asrc[j] = A + (int64_t)row * p.lda + ((slot ^ glds_swz<BK>(r)) << 3);
  }
#pragma unroll
  for (int j = 0; j < BPW; ++j) {
    const int r = wave * (BN / NWV) + j * PR + lr;
    const int row = min(n0 + r, p.N - 1);
    bsrc[j] = B + (int64_t)row * p.ldb + ((slot ^ glds_swz<BK>(r)) << 3);
  }

  auto issue = [&](int kt, int buf) {
    unsigned char* sa = glds_smem + buf * STAGE + wave * (BM / NWV) * ROWB;
    unsigned char* sb = glds_smem + buf * STAGE + BM * ROWB + wave * (BN / NWV) * ROWB;
#pragma unroll
    for (int j = 0; j < APW; ++j)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(asrc[j] + kt * BK), (lds_void_t*)(sa + j * PR * ROWB), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < BPW; ++j)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(bsrc[j] + kt * BK), (lds_void_t*)(sb + j * PR * ROWB), 16, 0, 0);
  };

  f32x4 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ALL NS stages are free at entry: k-tiles 0 .. NS-1 are requested back to back (their latencies overlap; the loop
  // itself can only refill a stage after the barrier that frees it, so iteration 0 has nothing to request).
#pragma unroll
  for (int s2 = 0; s2 < NS; ++s2)
    if (s2 < nk) issue(s2, s2);
  if (nk >= NS) asm volatile("s_waitcnt vmcnt(%0)" ::"i"((NS - 1) * LPT) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  VTX_TRACE(1);

  int buf = 0;
  for (int kt = 0; kt < ((GLDS_ABLATE & 1) ? 0 : nk); ++kt) {
    const bool refill = kt + NS - 1 < nk;
    if (refill && kt >= 1) issue(kt + NS - 1, buf == 0 ? NS - 1 : buf - 1);   // == (kt + NS - 1) % NS: freed by the last barrier
    const unsigned char* la = glds_smem + buf * STAGE;
    const unsigned char* lb = la + BM * ROWB;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      Vec8<bf16> fa[WM], fb[WN];
#pragma unroll
      for (int i = 0; i < WM; ++i) {
        const int r = wm * (BM / 2) + i * 16 + c_;
        fa[i] = load8<bf16>(reinterpret_cast<const bf16*>(la + r * ROWB + (((ks * 4 + g_) ^ glds_swz<BK>(r)) << 4)));
      }
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int r = wn * (BN / NWN) + j * 16 + c_;
        fb[j] = load8<bf16>(reinterpret_cast<const bf16*>(lb + r * ROWB + (((ks * 4 + g_) ^ glds_swz<BK>(r)) << 4)));
      }
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) mma16(fa[i], fb[j], acc[i][j]);
    }
    // tile kt+1 must have landed; up to NS-2 younger tiles may stay in flight (vmcnt retires in order)
    if (refill) asm volatile("s_waitcnt vmcnt(%0)" ::"i"((NS - 2) * LPT) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    buf = buf + 1 == NS ? 0 : buf + 1;
  }

  VTX_TRACE(2);
  gemm_epilogue<bf16, bf16, BM, BN, NWN>(p, acc, glds_smem, m0, n0, 0, wm, wn, c_, g_, eo);
  VTX_TRACE(7);
}


// ------------------------------------------------------------------------------------------------------------------
// Variant with a WAVE-PRIVATE epilogue (option GLDS_EPI = 1, the default): 128-column tiles, 64-deep k-tiles, 2 stages, 8 waves laid out
// (8 / NWN) x NWN with 32-row wave tiles -- 4 x 2 waves of 32 x 64 for BM = 128 (every wave owns whole 128-byte output
// lines), 2 x 4 waves of 32 x 32 for BM = 64.  Same main loop.  After the main loop's last barrier each wave transposes
// ITS OWN accumulators through a private 16-row LDS region (two passes) and stores them: no workgroup barrier in the
// epilogue -- the phase trace (tools/probe/gemm_trace.hip) shows 1.4 us of a 10.8 us workgroup lifetime spent in the shared
// staging passes, mostly waiting at their three barriers for the slowest wave.  Element values are those of
// gemm_epilogue (same expression per element), so the two variants are bitwise interchangeable.
// logical row -> row of the operands (GemmArgs::perm: stochastic-depth compaction; identity without a map)
template <bool MAPPED> __device__ __forceinline__ int glds_orow(const GemmArgs& p, int row) {
  if constexpr (!MAPPED) return row;
  const int s = (int)__umulhi((unsigned)row, p.map_magic);
  return p.perm[s] * p.map_T + (row - s * p.map_T);
}
// The same for the rows of ONE tile (rows m0 .. m0 + BM - 1): they span at most four samples when 3 T >= BM - 2 (a condition
// of the mapped launch; every compacted geometry of these models has T = 196 / 49 tokens per sample), whose perm entries are
// fetched with four SCALAR loads at kernel entry -- a per-lane perm[s] is a dependent vector load in front of the first DMA
// request and again in front of the epilogue's stores (in-model the mapped launches ran ~10 % over what their row count
// predicts).
template <bool MAPPED, int BM> struct TileRowMap {
  int s0, sp0, sp1, sp2, sp3;             // (four scalars, not an array: indexed by a lane value an array goes to scratch)
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
  // sample (perm applied) of the s-th sample in compacted order, s0 <= s <= s0 + 3
  __device__ __forceinline__ int sample_of(int s) const {
    const int d = s - s0;
    const int lo = d <= 0 ? sp0 : sp1, hi = d == 2 ? sp2 : sp3;
    return d <= 1 ? lo : hi;
  }
  __device__ __forceinline__ int orow(const GemmArgs& p, int row, int* smp = nullptr) const {
    if constexpr (!MAPPED) {
      if (smp) *smp = row / p.rows_per_scale;
      return row;
    } else {
      const int s = (int)__umulhi((unsigned)row, p.map_magic), sm = sample_of(s);
      if (smp) *smp = sm;
      return sm * p.map_T + (row - s * p.map_T);
    }
  }
};

template <int BM, int NWN, bool MAPPED = false> struct PvEpiOperands {
  static constexpr int WN = 128 / (16 * NWN);          // 16-column tiles per wave: 4 | 2
  static constexpr int VROW = 2 * WN;                  // 8-element vectors per staged row of the wave tile
  static constexpr int NIT = 16 * VROW / 64;           // store iterations per 16-row pass: 2 | 1
  float bcol[WN];
  float rsc[2 * NIT];
  Vec8<bf16> ein[2 * NIT];
  // (row, col) of the 8-vector this lane stores in iteration it of pass i; clamped in range, `ok` says whether it exists
  // `row` is the row of the operands (mapped: GemmArgs::perm); `srow` the sample whose DropPath scale applies
  using Map = TileRowMap<MAPPED, BM>;
  static __device__ __forceinline__ bool where(const GemmArgs& p, const Map& tmap, int m0, int n0, int wm, int wn, int lane,
                                               int i, int it, int& row, int& col, int* srow = nullptr) {
    const int v = lane + 64 * it, lr = v / VROW, cv = v - lr * VROW;
    const int lrow = m0 + wm * 32 + i * 16 + lr;
    col = n0 + wn * (16 * WN) + cv * 8;
    const bool ok = lrow < p.M && col < p.N;
    row = tmap.orow(p, (MAPPED && !ok) ? m0 : lrow, srow);
    return ok;
  }
  // bias per accumulator column and the DropPath scale per stored row: small, L2-resident; requested before the first DMA
  __device__ __forceinline__ void load_small(const GemmArgs& p, const Map& tmap, int m0, int n0, int wm, int wn, int lane) {
    const int c_ = lane & 15;
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int col = n0 + wn * (16 * WN) + j * 16 + c_;
      bcol[j] = (p.bias && col < p.N) ? p.bias[col] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 2 * NIT; ++q) {
      int row, col, srow;
      const bool ok = where(p, tmap, m0, n0, wm, wn, lane, q / NIT, q % NIT, row, col, &srow);
      ein[q] = vec8_zero<bf16>();
      rsc[q] = (ok && p.rowscale) ? p.rowscale[srow] : 1.f;
    }
  }
  // the residual / z vectors: HBM misses.  Loads return in order, so requested before the DMA pieces they hold back the
  // first k-tile (phase trace: 1.5 -> 2.6 us to the first k-tile) and requested in between they hold back the next one;
  // the kernel requests them right after its LAST DMA issue, where nothing younger is ever waited for.  EXACTLY NVEC load
  // instructions per wave when there is a source (no lane predicate: lanes without an element read element 0), so that
  // the counted vmcnt of the k-loop stays exact.
  static constexpr int NVEC = 2 * NIT;
  static __device__ __forceinline__ const bf16* vec_src(const GemmArgs& p) {
    return (p.act == 2 || p.act == 4) ? (const bf16*)p.aux_in : (const bf16*)p.resid;
  }
  __device__ __forceinline__ void load_vec(const GemmArgs& p, const Map& tmap, int m0, int n0, int wm, int wn, int lane) {
    const bf16* __restrict__ esrc = vec_src(p);
    if (!esrc || (GLDS_ABLATE & 8)) return;
#pragma unroll
    for (int q = 0; q < 2 * NIT; ++q) {
      int row, col;
      const bool ok = where(p, tmap, m0, n0, wm, wn, lane, q / NIT, q % NIT, row, col);
      ein[q] = load8<bf16>(esrc + (ok ? (int64_t)row * p.ldc + col : (int64_t)0));
    }
  }
};

__device__ __forceinline__ void glds_wave_sync() {      // orders this wave's own LDS traffic (DS operations issue in order)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int BM, int NWN, bool MAPPED = false>
__global__ __launch_bounds__(512) void gemm_glds_pv_kernel(GemmArgs p) {
  constexpr int BN = 128, BK = 64, NS = 2;
  constexpr int ROWB = BK * 2, CPR = BK / 8, PR = 1024 / ROWB, KS = BK / 32, NWV = 8;
  constexpr int NWM = NWV / NWN;
  static_assert(BM == 32 * NWM, "32-row wave tiles");
  constexpr int WM = 2, WN = BN / (16 * NWN);
  constexpr int STAGE = (BM + BN) * ROWB;
  constexpr int CSTR = 16 * WN + 4;                      // fp32 row stride of a wave's private staging region
  constexpr int PVB = 16 * CSTR * 4;                     // bytes per wave: 4 352 | 2 304
  static_assert(NWV * PVB <= NS * STAGE, "private staging must fit the ring");
  extern __shared__ __attribute__((aligned(16))) unsigned char glds_smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / NWN, wn = wave % NWN;
  const int c_ = lane & 15, g_ = lane >> 4;

  const int live_rows = MAPPED ? p.Mk : p.M;
  const int ntn = gridDim.x, ntm = gridDim.y;
  // Dispatch order -> tile: workgroup `did` lands on XCD did & 7; each XCD gets a contiguous chunk of the tile order (its
  // L2 then holds a band of A rows).  In a mapped launch only the tiles of the kept rows COMPUTE and the chunking runs over
  // those alone: chunked over all tiles, the copy-only tiles (last in tile order) would all fall to the last XCDs and the
  // others would hold ntiles / 8 compute tiles each -- 73 on 64 resident slots for the 588-tile N = 384 launches, a second
  // round for a launch whose compute tiles (~480) fit one (measured in the model: 42 us where the row count predicts 25).
  const int nblk = ntn * (MAPPED ? (live_rows + BM - 1) / BM : ntm);
  const int did = blockIdx.y * ntn + blockIdx.x;
  const int xq = nblk >> 3, xr = nblk & 7, xcd = did & 7;
  const int lid = (MAPPED && did >= nblk) ? did
                                          : (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (did >> 3);
  const int tn = lid % ntn, tm = lid / ntn;
  const int m0 = tm * BM, n0 = tn * BN;
  const int nk = p.K / BK;

  const bf16* A = (const bf16*)p.A;
  const bf16* B = (const bf16*)p.B;

  TileRowMap<MAPPED, BM> tmap;
  tmap.init(p, m0);
  if (MAPPED && m0 >= live_rows) {
    // copy-only tile of a mapped launch: the rows of DROPPED samples (DropPath scale 0): C = resid, no operands touched
    const bf16* __restrict__ rs = (const bf16*)p.resid;
    bf16* __restrict__ cd = (bf16*)p.C;
    if (rs != nullptr)
      for (int v = threadIdx.x; v < BM * 16; v += 512) {
        const int lrow = m0 + (v >> 4), col = n0 + (v & 15) * 8;
        if (lrow < p.M && col < p.N) {
          const int64_t off = (int64_t)tmap.orow(p, lrow) * p.ldc + col;
          store8<bf16>(cd + off, load8<bf16>(rs + off));
        }
      }
    return;
  }

  VTX_TRACE(0);
  PvEpiOperands<BM, NWN, MAPPED> eo;
  eo.load_small(p, tmap, m0, n0, wm, wn, lane);
  const bool has_vec = PvEpiOperands<BM, NWN, MAPPED>::vec_src(p) != nullptr && !(GLDS_ABLATE & 8);
  constexpr int NVEC = PvEpiOperands<BM, NWN, MAPPED>::NVEC;

  const int lr = lane / CPR, slot = lane % CPR;
  constexpr int APW = BM / (NWV * PR), BPW = BN / (NWV * PR);
  constexpr int LPT = APW + BPW;
  const bf16* asrc[APW];
  const bf16* bsrc[BPW];
#pragma unroll
  for (int j = 0; j < APW; ++j) {
    const int r = wave * (BM / NWV) + j * PR + lr;
    // rows past the computed ones (past M; past Mk in a tile that straddles the kept / dropped boundary of a mapped
    // launch) are never stored or are scaled by an exact 0: any valid, FINITE row will do -- the last computed one
    asrc[j] = A + (int64_t)tmap.orow(p, min(m0 + r, live_rows - 1)) * p.lda + ((slot ^ glds_swz<BK>(r)) << 3);
  }
#pragma unroll
  for (int j = 0; j < BPW; ++j) {
    const int r = wave * (BN / NWV) + j * PR + lr;
    bsrc[j] = B + (int64_t)min(n0 + r, p.N - 1) * p.ldb + ((slot ^ glds_swz<BK>(r)) << 3);
  }
  auto issue = [&](int kt, int buf) {
    unsigned char* sa = glds_smem + buf * STAGE + wave * (BM / NWV) * ROWB;
    unsigned char* sb = glds_smem + buf * STAGE + BM * ROWB + wave * (BN / NWV) * ROWB;
#pragma unroll
    for (int j = 0; j < APW; ++j)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(asrc[j] + kt * BK), (lds_void_t*)(sa + j * PR * ROWB), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < BPW; ++j)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(bsrc[j] + kt * BK), (lds_void_t*)(sb + j * PR * ROWB), 16, 0, 0);
  };

  f32x4 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // BOTH stages are free at entry: k-tiles 0 and 1 are requested back to back, so their latencies overlap (the loop
  // itself can only request k-tile kt + 1 after the barrier that frees its stage).  vmcnt retires in order: waiting for
  // k-tile 0 leaves exactly the younger requests outstanding (k-tile 1's LPT pieces, the NVEC epilogue vectors when
  // the last DMA request is already out).
  issue(0, 0);
  if (nk >= 2) issue(1, 1);
  const bool vec_early = nk <= 2;
  if (vec_early) eo.load_vec(p, tmap, m0, n0, wm, wn, lane);
  if (nk >= 2) {
    if (vec_early && has_vec) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(LPT + NVEC) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"i"(LPT) : "memory");
  } else {
    if (has_vec) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NVEC) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  VTX_TRACE(1);

  int buf = 0;
  bool vec_pending = vec_early;                            // the epilogue vectors are the youngest requests
  for (int kt = 0; kt < ((GLDS_ABLATE & 1) ? 0 : nk); ++kt) {
    const bool refill = kt + 1 < nk;
    if (refill && kt >= 1) issue(kt + 1, buf ^ 1);         // (k-tile 1 is already on its way)
    if (kt >= 1 && kt + 2 == nk) {                        // k-tile nk - 1 was just requested: the vectors go behind it
      eo.load_vec(p, tmap, m0, n0, wm, wn, lane);
      vec_pending = true;
    }
    const unsigned char* la = glds_smem + buf * STAGE;
    const unsigned char* lb = la + BM * ROWB;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      Vec8<bf16> fa[WM], fb[WN];
#pragma unroll
      for (int i = 0; i < WM; ++i) {
        const int r = wm * 32 + i * 16 + c_;
        fa[i] = load8<bf16>(reinterpret_cast<const bf16*>(la + r * ROWB + (((ks * 4 + g_) ^ glds_swz<BK>(r)) << 4)));
        if constexpr ((GLDS_ABLATE & 16) != 0) {          // timing probe (round 3): SiLU on the A fragments after the LDS read
#pragma unroll
          for (int e = 0; e < 8; ++e) fa[i].set(e, silu_f(fa[i].get(e)));
        }
      }
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int r = wn * (16 * WN) + j * 16 + c_;
        fb[j] = load8<bf16>(reinterpret_cast<const bf16*>(lb + r * ROWB + (((ks * 4 + g_) ^ glds_swz<BK>(r)) << 4)));
      }
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) mma16(fa[i], fb[j], acc[i][j]);
    }
    // k-tile kt + 1 must have landed.  After the last request only the epilogue vectors are younger (left in flight);
    // in the last iteration no DMA is outstanding any more.
    if (refill) {
      if (vec_pending && has_vec) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NVEC) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    buf ^= 1;
  }
  VTX_TRACE(2);

  // ---------------- wave-private epilogue: acc[i][j][r] = C[m0 + 32 wm + 16 i + 4 g + r][n0 + 16 WN wn + 16 j + c]
  using EO = PvEpiOperands<BM, NWN, MAPPED>;
  constexpr int VROW = EO::VROW, NIT = EO::NIT;
  float* cbuf = reinterpret_cast<float*>(glds_smem + wave * PVB);
  bf16* __restrict__ Cout = (bf16*)p.C;
  const bf16* __restrict__ resid = (const bf16*)p.resid;
  bf16* __restrict__ aux_out = (bf16*)p.aux_out;
  const bool act_fwd = p.act == 1 || p.act == 3, act_bwd = p.act == 2 || p.act == 4;
#pragma unroll
  for (int i = 0; i < WM; ++i) {
    if (i) glds_wave_sync();                              // the previous pass's reads are issued before these writes
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) cbuf[(g_ * 4 + r) * CSTR + j * 16 + c_] = acc[i][j][r] + eo.bcol[j];
    glds_wave_sync();
    VTX_TRACE(3 + 2 * i);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int q = i * NIT + it;
      int row, col;
      if (!EO::where(p, tmap, m0, n0, wm, wn, lane, i, it, row, col)) continue;
      const int64_t off = (int64_t)row * p.ldc + col;
      const int v = lane + 64 * it, lrow = v / VROW, cv = v - lrow * VROW;
      const float* cp = cbuf + lrow * CSTR + cv * 8;
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
    VTX_TRACE(4 + 2 * i);
  }
  VTX_TRACE(7);
}

template <int BM, int NWN, bool MAPPED> static int glds_launch_pv_m(const GemmArgs& a, hipStream_t st) {
  constexpr size_t smem = (size_t)2 * (BM + 128) * 128;
  auto kern = gemm_glds_pv_kernel<BM, NWN, MAPPED>;
  if (smem > 64 * 1024 &&
      hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
    return VTX_ERR_LAUNCH;
  dim3 grid((a.N + 127) / 128, (a.M + BM - 1) / BM, 1);
  hipLaunchKernelGGL(kern, grid, dim3(512), smem, st, a);
  return vtx_check_launch();
}
// (the row map of stochastic-depth compaction is a compile-time variant: the unmapped kernels carry none of its code)
template <int BM, int NWN> static int glds_launch_pv(const GemmArgs& a, hipStream_t st) {
  return a.perm != nullptr ? glds_launch_pv_m<BM, NWN, true>(a, st) : glds_launch_pv_m<BM, NWN, false>(a, st);
}

template <int BM, int BN, int BK, int NS, int NWN = 2> static int glds_launch_cfg(const GemmArgs& a, hipStream_t st) {
  constexpr size_t smem = (size_t)NS * (BM + BN) * BK * 2;
  auto kern = gemm_glds_kernel<BM, BN, BK, NS, NWN>;
  if (smem > 64 * 1024 &&
      hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
    return VTX_ERR_LAUNCH;
  dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM, 1);
  hipLaunchKernelGGL(kern, grid, dim3(128 * NWN), smem, st, a);
  return vtx_check_launch();
}

// Pipeline configuration measured on MI355X (tools/bench_gemm.py, stage 3/4 forward shapes, per-step total):
// (BK 64, 2 stages) 3.93 ms | (64, 3) 5.23 ms | (32, 3) 3.90 ms | (32, 4) 4.30 ms -- deeper rings cost
// occupancy (LDS) and do not pay: these GEMMs (K = 384..3072) are bound by their epilogue's HBM writes and by
// per-block prologue/epilogue, not by DMA latency.  The shipped configuration is (64, 2).
// K % 64 != 0 but K % 32 == 0 (stage-1 K = 96): 32-deep k-tiles, 3 stages (the epilogue's C staging needs the room);
// only with 128-column tiles -- a 32-deep tile row is 64 B, one DMA instruction covers 16 rows, and 96 / 4 waves does
// not split into whole instructions.  Measured (stage 1, K = 96, N = 384): fc1 forward 256 -> 215 us, fc2 dgrad
// 283 -> 203 us vs the register-staged kernel; N = 96 / 288 shapes stay on the register-staged kernel (faster there).
template <int BM, int BN> static int glds_launch_t(const GemmArgs& a, hipStream_t st) {
  if constexpr (BN == 128) {
    if (a.K % 64 != 0) return glds_launch_cfg<BM, BN, 32, 3>(a, st);
  }
  if constexpr (BN == 128) {
    // 2 x 4 waves: fwd 3.95 -> 3.78, dgrad 3.66 -> 3.55 ms per step on the Swin stage-2..4 shapes, ViT-S/16 4.64 -> 4.38 /
    // 4.14 -> 4.07 (the activation epilogues gain most); option GLDS_WAVES = 4 keeps the 2 x 2 variant for comparison
    if (vtx_opt(VTX_OPT_GLDS_EPI) == 1) return glds_launch_pv<BM, BM == 128 ? 2 : 4>(a, st);
    if (vtx_opt(VTX_OPT_GLDS_WAVES) != 4) return glds_launch_cfg<BM, BN, 64, 2, 4>(a, st);
  }
  return glds_launch_cfg<BM, BN, 64, 2>(a, st);
}

bool gemm_glds_ok(int N, int K) { return (K % 64) == 0 || ((K % 32) == 0 && (N % 128) == 0); }

// Tile height.  64-row tiles (48 KB of LDS, 3 workgroups per CU) win wherever a launch has few tiles: the per-tile
// prologue / epilogue is hidden by the other resident workgroups.  With 2 x 4 waves per tile the 128 x 128 tile (64 KB,
// 2 workgroups = 16 waves per CU, 2/3 of the operand bytes per FLOP) wins once a launch has >= ~1.6 rounds of them:
// per shape (tools/bench_gemm.py, same box, 128 rows / 64 rows) ViT-S/16 0.80-0.95 on all eight GEMMs, Swin stage 4
// 0.87-0.96 on the wide ones, stage 3 0.92-0.99 (N >= 1152) but 1.05-1.10 on the N = 384 GEMMs (588 tiles), stage 2
// 0.93-0.96 (N = 768).  In the models: ViT-S/16 +4.9 %, Swin-S -0.9 % with 128 rows everywhere -> chosen per launch.
// Option GLDS_BM = 64 | 128 forces one.
static int glds_pick_bm(const GemmArgs& a, int bn) {
  const int force = vtx_opt(VTX_OPT_GLDS_BM);
  if (force == 64 || force == 128) return force;
  if (bn != 128 || a.K % 64 != 0) return 64;                 // the 2 x 4-wave tiles exist for 128 columns, 64-deep k-tiles
  // (a mapped launch computes Mk rows; the copy-only tiles behind them cost ~nothing)
  const long rows = a.perm != nullptr ? a.Mk : a.M;
  const long tiles128 = (long)((a.N + 127) / 128) * ((rows + 127) / 128);
  // ONE round of 128-row tiles (two 64-KB workgroups per CU = 512 resident) beats the 1.3-2 rounds of 64-row tiles the same
  // rows need: measured on the N = 384 stage-3 shapes at the row counts stochastic-depth compaction leaves (21 756 rows =
  // 510 tiles: fc2 forward 40.0 -> 31.4 us, fc1 dgrad 40.0 -> 30.1, qkv dgrad 31.0 -> 24.7, proj 15.5 -> 13.7; at 23 128
  // rows = 543 tiles the 64-row tile still wins: tools/probe/bm_probe.py).  Up to 384 tiles the 64-row tiles (768 resident)
  // fit one round themselves and are as fast or faster (Swin stage 4, 294 tiles: 17.1 vs 17.9, 49.6 vs 54.0 us: bm_probe2.py).
  if (tiles128 > 384 && tiles128 <= 512) return 128;
  return tiles128 >= 800 ? 128 : 64;
}

template <int BN> static int glds_launch_bn(const GemmArgs& a, hipStream_t st) {
  return glds_pick_bm(a, BN) == 64 ? glds_launch_t<64, BN>(a, st) : glds_launch_t<128, BN>(a, st);
}

bool gemm_glds_enabled() {
  return vtx_opt(VTX_OPT_GEMM_GLDS) != 0;
}

int gemm_glds_launch_mapped(const GemmArgs& a, hipStream_t st) {
  if (a.perm == nullptr || a.map_T <= 0 || a.Mk <= 0 || a.Mk > a.M) return VTX_ERR_SHAPE;
  if (a.N % 128 != 0 || a.K % 64 != 0 || vtx_opt(VTX_OPT_GLDS_EPI) != 1) return VTX_ERR_SHAPE;   // wave-private epilogue kernels only
  if (a.rowscale != nullptr && a.rows_per_scale != a.map_T) return VTX_ERR_SHAPE;
  if (3 * a.map_T < 126) return VTX_ERR_SHAPE;                      // TileRowMap: a 128-row tile spans at most four samples
  if (a.Mk < a.M && a.resid == nullptr) return VTX_ERR_SHAPE;       // copy-only rows need something to copy
  if (gemm_pp_ok(a)) return gemm_pp_launch(a, st);
  if (gemm_astat_ok(a)) return gemm_astat_launch(a, st);
  return glds_launch_bn<128>(a, st);
}

int gemm_glds_launch(const GemmArgs& a, hipStream_t st) {
  if (gemm_pp_ok(a)) return gemm_pp_launch(a, st);
  if (gemm_astat_ok(a)) return gemm_astat_launch(a, st);
  if (a.N % 128 == 0) return glds_launch_bn<128>(a, st);
  if (a.N % 96 == 0) return glds_launch_bn<96>(a, st);
  if (a.N <= 64) return glds_launch_bn<64>(a, st);
  return glds_launch_bn<128>(a, st);
}
