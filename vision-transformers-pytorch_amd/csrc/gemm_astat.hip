// A-STATIONARY, PERSISTENT LDS-DMA GEMM for gfx950 (bf16): C[M,N] = epilogue(A[M,K] . W[N,K]^T) with a SHORT contraction
// 192 <= K <= 384 (Swin-S stage 3 / ViT-S/16: qkv, fc1, proj forward; proj and fc2 dgrad) and N a multiple of 128.
//
// What the counters say about the tiled kernel on these shapes (profiles/round4_pmc_sq_gemm_in_model.txt): no unit is busy
// (MFMA ~18 %, LDS ~18 %, TA ~37 %, HBM 0.3), the waves are parked -- a 128 x 128 tile with K = 384 is six k-tiles between a
// prologue that waits for the first operands and an epilogue, every one of its k-tiles waits for the ONE k-tile of LDS-DMA the
// 2-stage ring keeps in flight, and the A rows are fetched again for every column tile.  In the model a round of resident
// workgroups takes 10-14 us for 2.6 us of MFMA work.  Here
//   * ONE persistent workgroup per CU walks a contiguous RANGE of the launch's 128 x 128 output tiles in strip-major order (all
//     column tiles of a 128-row strip, then the next strip); the ranges are equal to within one tile, so the launch has no
//     round quantisation (196 strips x 9 column tiles on 256 CUs: 6.9 tiles each instead of 1 strip for 196 of them);
//   * the strip's K x 128 panel of A (<= 96 KB) stays in LDS for every column tile of the strip the range covers; when the range
//     moves on to the next strip each 16-KB k-tile slot is refilled right after its last use, NKT - 1 k-steps before its first
//     use for the new strip;
//   * the weight streams through a ring of three 16-KB stages as ONE continuous sequence of k-tiles over all tiles of the
//     range: no pipeline drain between tiles, no per-tile prologue; counted s_waitcnt vmcnt + one raw s_barrier per k-step;
//   * ten waves: waves 0-7 multiply (4 x 2 wave tiles of 32 x 64) and store, waves 8-9 only REQUEST the streams.  vmcnt retires
//     in order per wave: a wave that stores a tile and then waits for a younger LDS-DMA request waits for the acknowledgement of
//     its stores first (measured on the first version of this kernel, where every wave requested and stored: ~3 us per column
//     tile).  With the requests on their own waves the multiplying waves never wait for memory inside the loop except for the
//     residual / z vectors, which are requested one tile period ahead;
//   * the weight rows of a wave's 64 columns go to LDS in the order n = 4 c + j (LDS row 16 j + c): after the four 16 x 16
//     products of an accumulator row a lane holds FOUR CONSECUTIVE output columns -- one wave-wide 8-byte store covers 4 rows x
//     128 contiguous bytes straight from the accumulators: no LDS staging, no epilogue barrier.  (v1 / v2 took the product
//     transposed for 16-byte vectors: 16 rows x 64 bytes per instruction, 64 cache lines per store instruction through the same
//     address path the weight stream's requests use -- the stores cost 29 of 47 us: gpurun_out/r4job8/ablate.log.)
//   * the epilogue of tile j is spread over the first k-steps of tile j + 1 (the accumulators are copied).
// Same element-wise epilogue expressions and the same k order inside the MFMA as gemm_glds.hip: bit-identical outputs.
// Stochastic-depth compaction (GemmArgs::perm): row maps from an LDS copy of perm; rows of dropped samples are copied by the
// request waves after their last request.
#include <type_traits>

#include "gemm_common.h"
#include "options.h"

// phase ablation for timing probes (tools/r4/build_variant.sh; results are garbage, durations are what is measured):
// 1 no MFMA | 2 no weight DMA inside the loop | 4 no epilogue stores | 8 no fragment reads | 16 no k-step barrier | 32 no epilogue at all | 64 the request waves do not wait for a k-tile | 128 stores confined to 2 MB | 256 half as many 16-byte stores | 512 stores of raw accumulator bits, no epilogue arithmetic
#ifndef ASTAT_ABLATE
#define ASTAT_ABLATE 0
#endif
// ASTAT_TRACE: shader-clock stamps of workgroup ASTAT_TRACE_WG's first request wave and first multiplying wave, per k-step, kept in
// LDS and dumped to a device array at the end (tools/r4/astat_trace.py; compiled out of the library)
#ifndef ASTAT_TRACE
#define ASTAT_TRACE 0
#endif
#if ASTAT_TRACE
#define AS_TRACE_STEPS 128
__device__ unsigned as_trace_buf[2][4][AS_TRACE_STEPS];
extern "C" int vtx_astat_trace_read(unsigned* dst) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(as_trace_buf), sizeof(as_trace_buf)) == hipSuccess ? 0 : -1;
}
#define AS_STAMP(kind, slot, q) do { if (tr_on && (q) < AS_TRACE_STEPS) { const unsigned t_ = (unsigned)__builtin_amdgcn_s_memtime(); if (lane == 0) strace[((kind) * 4 + (slot)) * AS_TRACE_STEPS + (q)] = t_; } } while (0)
#else
#define AS_STAMP(kind, slot, q)
#endif
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;


constexpr int AS_BM = 128, AS_BN = 128, AS_BK = 64;
constexpr int AS_KT_BYTES = AS_BM * AS_BK * 2;          // 16 KB: one k-tile slot of the A strip = one stage of the W ring
constexpr int AS_NSW = 3;                               // stages of the weight ring
constexpr int AS_MAXS = 512;                            // samples of the LDS copies of perm / rowscale
constexpr int AS_MAXN = 1536;                           // columns of the LDS copy of the bias
#ifndef AS_NLW
#define AS_NLW 4
#endif
#ifndef AS_NCWD
#define AS_NCWD 8          // multiplying waves: 8 (4 x 2 wave tiles of 32 x 64) | 4 (2 x 2 wave tiles of 64 x 64: a third less LDS fragment traffic, one
                           // multiplying wave per SIMD -- a TIMING PROBE: measured no faster (stage-3 qkv forward 28.9 vs 27.3 us, fc1 56.8 vs 58.4) and its
                           // residual / act' kinds spill at 224 registers, which the counted waits do not survive)
#endif
constexpr int AS_NCW = AS_NCWD, AS_NRW = AS_NLW;        // multiplying waves; request waves
constexpr int AS_WMT = 16 / AS_NCW;                     // 16-row MFMA tiles per wave tile along M: 2 | 4
static_assert(AS_NCW == 8 || AS_NCW == 4, "wave tilings");
constexpr int AS_NT = 64 * (AS_NCW + AS_NRW);           // threads
constexpr int AS_LW = 16 / AS_NRW;                      // DMA instructions per request wave and 16-KB k-tile
constexpr int AS_RPW = 128 / AS_NRW;                    // LDS rows per request wave

struct AstatSched {
  int ntn;            // column tiles per strip
  int ntiles;         // tiles of the COMPUTED strips (rows [0, Mk) of a mapped launch)
  int nwg;            // persistent workgroups (== gridDim.x)
  int copy_row0;      // mapped launch with a residual: rows [copy_row0, M) are copy-only (C = resid); == M otherwise
  int ragged;         // N % 128 == 64 (round 5: Swin-S stage-2 qkv, N = 576): in the LAST column tile of a strip the second wave column works
                      // on the FIRST one's 64 columns again -- same weight rows, same bits stored to the same addresses (as the rows past M of
                      // a partial last strip): every wave still stores every vector, the counted waits do not change
};

// logical row -> row of the operands (GemmArgs::perm through its LDS copy; identity without a map); *smp: the sample whose
// DropPath scale applies
template <bool MAPPED>
__device__ __forceinline__ int as_orow(const GemmArgs& p, const int* sperm, unsigned rs_magic, int row, int* smp) {
  if constexpr (!MAPPED) {
    if (smp) *smp = (int)__umulhi((unsigned)row, rs_magic);
    return row;
  } else {
    const int s = (int)__umulhi((unsigned)row, p.map_magic), sm = sperm[s];
    if (smp) *smp = s;                                   // (the scale table is in LOGICAL sample order: see srs)
    return sm * p.map_T + (row - s * p.map_T);
  }
}

// The residual / z vectors of a tile land in FIXED registers v152 .. v167 that the compiler never allocates (the kernel is
// compiled with amdgpu_num_vgpr(152); the clobber lists below make the kernel descriptor count 168 = three waves per SIMD):
// requested by inline asm, used one tile period later behind a hand-counted s_waitcnt.  Why not compiler-visible registers:
//   * as ordinary loads hipcc waits vmcnt(0) in front of every store of the loop (its bookkeeping gives up at the loop header:
//     ISA of the first v3 build) -- the acknowledgement of the stores just issued, ~1 us per k-step;
//   * as inline-asm OUTPUT operands the compiler believes the registers hold their value when the asm statement ends and is free
//     to copy them -- it did: a v_mov of the destination pair in front of the hand-written wait (ISA of the second build).
#if AS_NCWD == 8
constexpr int AS_EV0 = 152;          // 12 waves: three per SIMD, 168 registers each
#define AS_EV_CLOBBERS "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167"
#else
constexpr int AS_EV0 = 224;          // 8 waves: two per SIMD, 256 registers each; 16 vectors per lane and tile
#define AS_EV_CLOBBERS "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", \
                       "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255"
#endif
template <int V> __device__ __forceinline__ void as_request(int byte_off, const bf16* base) {
  asm volatile("global_load_dwordx2 v[%2:%3], %0, %1" ::"v"(byte_off), "s"(base), "i"(AS_EV0 + 2 * V), "i"(AS_EV0 + 2 * V + 1)
               : "memory", AS_EV_CLOBBERS);
}
// wait until at most N younger memory operations of this wave are outstanding, then read vector V
template <int V, int N> __device__ __forceinline__ u32x2 as_take() {
  unsigned lo, hi;
  asm volatile("s_waitcnt vmcnt(%2)\n\tv_mov_b32 %0, v[%3]\n\tv_mov_b32 %1, v[%4]" : "=v"(lo), "=v"(hi) : "i"(N), "i"(AS_EV0 + 2 * V), "i"(AS_EV0 + 2 * V + 1) : "memory");
  return u32x2{lo, hi};
}

// (Built and measured, not kept -- tools/r4/rejected/gemm_astat_flags.hip.txt: producer / consumer counters in LDS instead of the
// k-step barrier.  With three ring stages the request waves cannot run far enough ahead for the decoupling to pay, and a polled
// LDS counter costs more than s_barrier: 33.9 vs 29.8 us on the stage-3 qkv forward, 16.3 vs 9.7 us with stores and DMA compiled out.)

#ifndef ASTAT_AUX_A
#define ASTAT_AUX_A 0       // cache-policy bits of the A strip's LDS-DMA (2 = nt: the strip is read once per launch) -- A/B: profiles/round6_nt_load_screen.txt
#endif
#ifndef ASTAT_NT
#define ASTAT_NT 1          // non-temporal stores of C / z: the outputs stream past the L2 that holds the weight panel and the A strips (ViT-S/16 step -2.7 %, Swin-S -0.3 %)
#endif
template <bool NT> __device__ __forceinline__ void as_store4(bf16* dst, bf16x4 v) {
  if constexpr (NT && ASTAT_NT != 0) __builtin_nontemporal_store(v, reinterpret_cast<bf16x4*>(dst));
  else *reinterpret_cast<bf16x4*>(dst) = v;
}

// NKT = K / 64 k-tile slots of the strip resident in LDS
// ACT = GemmArgs::act as a COMPILE-TIME value (a run-time switch is a branch per epilogue vector: the MFMAs of the next tile could not
// be scheduled into the epilogue's basic blocks); RESID: C += resid (ACT 0 only); AUX: the activation forward also writes z
template <int NKT, bool MAPPED, int ACT, bool RESID, bool AUX>
__global__ __launch_bounds__(AS_NT) __attribute__((amdgpu_num_vgpr(AS_EV0))) void gemm_astat_kernel(GemmArgs p, AstatSched sc) {
  constexpr bool act_fwd = ACT == 1 || ACT == 3, act_bwd = ACT == 2 || ACT == 4;
  constexpr bool VEC = RESID || act_bwd;               // the epilogue reads one 8-byte vector per output vector (the residual, or z for act')
  static_assert(!(RESID && ACT != 0) && !(AUX && !act_fwd), "epilogue kinds of the hot path");
  constexpr int ROWB = 128;                            // bytes per LDS row
  constexpr int NV = 4 * AS_WMT;                       // 8-byte vectors per lane and tile: rows (i, r), columns 4 c_ .. 4 c_ + 3
  constexpr int NST = NV * (AUX ? 2 : 1);              // store instructions per multiplying wave and tile
  constexpr int SPI = (NV + NKT - 1) / NKT;            // epilogue vectors per k-step
  static_assert(NKT >= 3 && NKT <= 6, "a refilled A slot needs two k-steps to land; 6 slots + the ring fill the LDS");
  extern __shared__ __attribute__((aligned(16))) unsigned char as_smem[];
  unsigned char* const sa = as_smem;                                  // [NKT][128 rows][128 B]
  unsigned char* const sw = as_smem + NKT * AS_KT_BYTES;              // [AS_NSW][128 rows][128 B]
  // The small tables are SEPARATE LDS objects, not offsets into the dynamic array: hipcc keeps book of LDS that LDS-DMA may have
  // written and puts s_waitcnt vmcnt(0) in front of any LDS read it cannot prove disjoint from it -- a table at a run-time offset
  // of the DMA'd array cost every k-step a wait for all of the wave's stores (ISA of the first build with the row table).
  __shared__ __attribute__((aligned(16))) float sbias[AS_MAXN];       // bias
  __shared__ float srs[AS_MAXS];                                      // DropPath scale by sample
  __shared__ int sperm[AS_MAXS];                                      // perm
  // [2][128] per strip row {element offset of the row in C / resid / z, DropPath scale}: written by the request waves one tile before
  // the strip starts (they compute the rows for the A refill anyway), read by the multiplying waves with their fragments -- the
  // epilogue carries no row arithmetic (v_mul_hi / v_mul_lo are quarter rate, and an LDS lookup in the middle of the epilogue stalls
  // the wave's MFMAs behind its lgkmcnt wait)
  __shared__ __attribute__((aligned(16))) int2 srow[2 * AS_BM];
  __shared__ int srowa[2 * AS_BM];                                    // [2][128] element offset of a strip's rows in A (for the request waves); by strip parity:
                                                                      // the multiplying waves write the next strip's while the first strip's may still be unread
#if ASTAT_TRACE
  __shared__ unsigned strace[2 * 4 * AS_TRACE_STEPS];
#endif

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int live_rows = MAPPED ? p.Mk : p.M;
  const int ntn = sc.ntn;

  // ---- this workgroup's tile range.  Workgroup `did` lands on XCD did & 7: each XCD gets a contiguous band of ranges (neighbours
  // share a strip of A through its L2)
  const int did = blockIdx.x, nwg = sc.nwg;
  const int xq = nwg >> 3, xr = nwg & 7, xcd = did & 7;
  const int lid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (did >> 3);
  const int tq = sc.ntiles / nwg, trem = sc.ntiles - tq * nwg;
  const int t0 = lid * tq + min(lid, trem), t1 = t0 + tq + (lid < trem ? 1 : 0);
  const int Q = (t1 - t0) * NKT;                         // k-steps of this workgroup
#if ASTAT_TRACE
  const bool tr_on = lid == (ASTAT_TRACE - 1) && (wave == 0 || wave == AS_NCW);
  if (tr_on) for (int i = lane; i < 4 * AS_TRACE_STEPS; i += 64) strace[(wave == 0 ? 1 : 0) * 4 * AS_TRACE_STEPS + i] = 0;
#endif

  // ---- small tables: bias, DropPath scales, the sample order
  const int nsamp = MAPPED ? p.M / p.map_T : (p.rowscale ? (p.M + p.rows_per_scale - 1) / p.rows_per_scale : 1);    // (no scales: srs[0] = 1)
  // (no bias: the table is zeros and is read at the lane's in-tile column only, whatever N is)
  for (int i = threadIdx.x; i < (p.bias ? p.N : AS_BN); i += AS_NT) sbias[i] = p.bias ? p.bias[i] : 0.f;
  const int bias_on = p.bias ? 1 : 0;
  for (int i = threadIdx.x; i < nsamp; i += AS_NT) {
    // DropPath scale of the i-th sample IN PERM ORDER (a mapped launch without copy-only rows has M = Mk: the table covers the kept
    // samples only, while perm's values -- the physical samples rowscale is indexed by -- range over the whole batch)
    if constexpr (MAPPED) {
      const int sm = p.perm[i];
      sperm[i] = sm;
      srs[i] = p.rowscale ? p.rowscale[sm] : 1.f;
    } else {
      srs[i] = p.rowscale ? p.rowscale[i] : 1.f;
    }
  }
  const unsigned rs_magic = p.rowscale ? (unsigned)((0x100000000ull + (unsigned)p.rows_per_scale - 1) / (unsigned)p.rows_per_scale) : 0u;
  __syncthreads();                                        // (nothing is in flight yet: an ordinary barrier)
  auto fill_row = [&](int strip_, int r) {
    int smp = 0;
    // rows past M (a partial last strip; a mapped launch without copy-only rows has M = Mk, any multiple of T) are the LAST row again:
    // same operand row, same weight, same scale -> the same bits stored to the same address by several lanes; every lane stores every
    // vector, which is what the counted waits of the epilogue rely on
    const int row = as_orow<MAPPED>(p, sperm, rs_magic, min(strip_ * AS_BM + r, p.M - 1), &smp);
    srow[(strip_ & 1) * AS_BM + r] = int2{row * (int)p.ldc, __builtin_bit_cast(int, srs[smp])};
    // (A rows past the computed ones of a mapped launch read the last computed row: finite values, scaled by an exact 0)
    srowa[(strip_ & 1) * AS_BM + r] = as_orow<MAPPED>(p, sperm, rs_magic, min(strip_ * AS_BM + r, live_rows - 1), nullptr) * (int)p.lda;
  };
  if (threadIdx.x < AS_BM) fill_row(t0 / ntn, threadIdx.x);
  __syncthreads();

  const int lr = lane >> 3, slot = lane & 7;
  int tn = t0 % ntn, strip = t0 / ntn;
  if (wave >= AS_NCW) {
    // =================================================================== the request waves
    const int lw = wave - AS_NCW;
    if (Q > 0) {
      // W: LDS row R of a 128-column tile <- weight row 64 (R >> 6) + 4 (R & 15) + ((R >> 4) & 3); XOR swizzle of the 16-byte chunk by
      // the LDS row (applied on the source address: global_load_lds writes lane-linear)
      const bf16* wsrc[AS_LW];
#pragma unroll
      for (int j = 0; j < AS_LW; ++j) {
        const int r = lw * AS_RPW + j * 8 + lr;
        const int n = 64 * (r >> 6) + 4 * (r & 15) + ((r >> 4) & 3);
        wsrc[j] = (const bf16*)p.B + (int64_t)n * p.ldb + ((slot ^ (r & 7)) << 3);
      }
      const int64_t wrag = sc.ragged ? -64 * (int64_t)p.ldb : 0;     // ragged last column tile: LDS rows 64 .. 127 <- the weight rows of rows 0 .. 63
      const int64_t wtile = (int64_t)AS_BN * p.ldb;       // elements between column tiles of W
      auto issue_w = [&](int tnw, int kt, int stage) {
#pragma unroll
        for (int j = 0; j < AS_LW; ++j)
          __builtin_amdgcn_global_load_lds((gbl_void_t*)(wsrc[j] + tnw * wtile + kt * AS_BK + ((tnw == ntn - 1 && ((lw * AS_RPW + j * 8) >> 6)) ? wrag : 0)),
                                           (lds_void_t*)(sw + stage * AS_KT_BYTES + (lw * AS_RPW + j * 8) * ROWB), 16, 0, 0);
      };
      // A refill: LDS row r <- row r of the NEXT strip (row map applied; rows past the computed ones read the last one)
      // The row offsets come from the srowa table the multiplying waves fill (they issue no LDS-DMA: hipcc puts s_waitcnt vmcnt(0) in
      // front of an ordinary LDS read of a wave that has DMA in flight -- here the table is read by hand, once per strip)
      int aoff[AS_LW];
      auto read_aoff = [&](int strip_, bool by_hand) {
#pragma unroll
        for (int j = 0; j < AS_LW; ++j) {
          const int r = lw * AS_RPW + j * 8 + lr;
          const int* src = srowa + (strip_ & 1) * AS_BM + r;
          int o;
          if (by_hand) asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(o) : "v"((unsigned)(size_t)src) : "memory");
          else o = *src;
          aoff[j] = o + ((slot ^ (r & 7)) << 3);
        }
      };
      auto issue_a = [&](int kt) {
#pragma unroll
        for (int j = 0; j < AS_LW; ++j)
          __builtin_amdgcn_global_load_lds((gbl_void_t*)((const bf16*)p.A + aoff[j] + kt * AS_BK),
                                           (lds_void_t*)(sa + kt * AS_KT_BYTES + (lw * AS_RPW + j * 8) * ROWB), 16, 0, ASTAT_AUX_A);
      };
      // weight k-tile q + 2 is requested in k-step q
      int tnw = tn, ktw = 0, stw = 0;
      auto advance_w = [&]() {
        ktw += 1;
        if (ktw == NKT) { ktw = 0; tnw = tnw + 1 == ntn ? 0 : tnw + 1; }
        stw = stw + 1 == AS_NSW ? 0 : stw + 1;
      };
      // the first strip's A panel, every k-tile now (one latency), then the first two weight k-tiles
      read_aoff(strip, false);                            // (nothing in flight yet: an ordinary read)
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) issue_a(kt);
      issue_w(tnw, ktw, stw); advance_w();
      if (Q > 1) { issue_w(tnw, ktw, stw); advance_w(); }
      int pend = Q > 1 ? AS_LW : 0;                       // requests that may stay in flight at the next wait (younger than the k-tile it needs)
      bool refill_prev = false;                           // the k-step before this one read its A slot for the last time
      int q = 0;
      for (int t = t0; t < t1; ++t) {
        const bool last_of_strip = tn == ntn - 1 && t + 1 < t1;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt, ++q) {
          // k-tile q has landed when at most `pend` younger requests are outstanding (nothing else is ever requested here)
          AS_STAMP(0, 0, q);
          if (ASTAT_ABLATE & 64) { }
          else if (pend == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          else if (pend == AS_LW) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(AS_LW) : "memory");
          else asm volatile("s_waitcnt vmcnt(%0)" ::"i"(2 * AS_LW) : "memory");
          AS_STAMP(0, 1, q);
          if (!(ASTAT_ABLATE & 16)) __builtin_amdgcn_s_barrier();                     // barrier q: k-tile q is visible; every wave is past its reads of k-step q - 1
          AS_STAMP(0, 2, q);
          pend = 0;
          if (kt == 1 && last_of_strip) read_aoff(strip + 1, true);      // (written by the multiplying waves in front of barrier (t, 0))
          if (refill_prev) { issue_a(kt == 0 ? NKT - 1 : kt - 1); pend += AS_LW; }
          if (q + 2 < Q && !(ASTAT_ABLATE & 2)) { issue_w(tnw, ktw, stw); advance_w(); pend += AS_LW; }
          refill_prev = last_of_strip;
          AS_STAMP(0, 3, q);
        }
        tn += 1;
        if (tn == ntn) { tn = 0; strip += 1; }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if ASTAT_TRACE
      if (tr_on) for (int i = lane; i < 4 * AS_TRACE_STEPS; i += 64) as_trace_buf[0][0][i] = strace[i];
#endif
    }
    // ---- rows of dropped samples of a mapped launch: C = resid (their strips are not in any tile range)
    if constexpr (MAPPED) {
      const bf16* __restrict__ rs = (const bf16*)p.resid;
      bf16* __restrict__ cd = (bf16*)p.C;
      if (rs != nullptr && sc.copy_row0 < p.M) {
        const int vrow = p.N >> 3, nvec = (p.M - sc.copy_row0) * vrow;
        for (int v = (lid * AS_NRW + lw) * 64 + lane; v < nvec; v += nwg * AS_NRW * 64) {
          const int lrw = v / vrow, cv = v - lrw * vrow;
          const int64_t off = (int64_t)as_orow<MAPPED>(p, sperm, rs_magic, sc.copy_row0 + lrw, nullptr) * p.ldc + cv * 8;
          store8<bf16>(cd + off, load8<bf16>(rs + off));
        }
      }
    }
    return;
  }

  // ======================================================================= the eight multiplying waves
  if (Q == 0) return;
  const int wm = wave >> 1, wn = wave & 1;               // 4 x 2 waves of 32 x 64 | 2 x 2 waves of 64 x 64
  constexpr int WROWS = 16 * AS_WMT;                    // rows of a wave tile
  const int c_ = lane & 15, g_ = lane >> 4;

  // ---- rows this lane finishes: row(i, r) = strip * 128 + 32 wm + 16 i + 4 g_ + r, columns n0 + 64 wn + 4 c_ .. + 3 (v = 4 i + r).
  // Element offset of the row in C / resid / z (all share ldc), its DropPath scale and whether it exists are recomputed per
  // vector from the strip number (a handful of VALU instructions and, in a mapped launch, one LDS read behind the MFMAs): kept in
  // registers they are 17 per lane, and the kernel has 168 (three waves on two of the SIMDs).
  const bf16* __restrict__ vsrc = VEC ? (act_bwd ? (const bf16*)p.aux_in : (const bf16*)p.resid) : nullptr;
  // residual / z vectors of a tile: eight 8-byte loads per lane, ONE register pair per vector (as_request / as_take): vector v of
  // tile j + 1 is requested right behind the store of vector v of tile j, which consumed the pair, and used one tile period later.
  // vmcnt retires in order; between the request of a vector and its use this wave issues the (store, request) pairs of the 7 other
  // vectors = 14 operations (second tile of a range: the first tile's vectors were requested back to back, 7 - v requests + 2 v
  // operations; after the loop: the last period's pairs + v stores).  The waits below allow 7 + v (in the loop) and 7 (behind it)
  // outstanding operations: exact for the second tile, a little early afterwards (operations issued >= 4 k-steps ago).
  const int colq = wn * 64 + c_ * 4;                      // this lane's columns inside a tile
  auto row_entry = [&](int strip_, int v) {              // {element offset of row (i, r) of the strip, its DropPath scale}
    return srow[(strip_ & 1) * AS_BM + wm * WROWS + (v >> 2) * 16 + g_ * 4 + (v & 3)];
  };
  // (v is a compile-time constant after unrolling: the switches fold)
  auto request_vec = [&](int v, int off, int n0) {
    const bf16* base = vsrc + n0;
    switch (v) {
#define AS_RQ(V) case V: as_request<V>(off * 2, base); break;
      AS_RQ(0) AS_RQ(1) AS_RQ(2) AS_RQ(3) AS_RQ(4) AS_RQ(5) AS_RQ(6) AS_RQ(7)
#if AS_NCWD == 4
      AS_RQ(8) AS_RQ(9) AS_RQ(10) AS_RQ(11) AS_RQ(12) AS_RQ(13) AS_RQ(14) AS_RQ(15)
#endif
#undef AS_RQ
      default: break;
    }
  };
  // in the loop NV - 1 + v younger operations may stay outstanding, behind it NV - 1 (see above)
  auto take_vec = [&](int v, bool in_loop) -> u32x2 {
    if (in_loop) {
      switch (v) {
#define AS_TK(V) case V: return as_take<V, NV - 1 + V>();
        AS_TK(0) AS_TK(1) AS_TK(2) AS_TK(3) AS_TK(4) AS_TK(5) AS_TK(6) AS_TK(7)
#if AS_NCWD == 4
        AS_TK(8) AS_TK(9) AS_TK(10) AS_TK(11) AS_TK(12) AS_TK(13) AS_TK(14) AS_TK(15)
#endif
#undef AS_TK
        default: return u32x2{0u, 0u};
      }
    }
    switch (v) {
#define AS_TK(V) case V: return as_take<V, NV - 1>();
      AS_TK(0) AS_TK(1) AS_TK(2) AS_TK(3) AS_TK(4) AS_TK(5) AS_TK(6) AS_TK(7)
#if AS_NCWD == 4
      AS_TK(8) AS_TK(9) AS_TK(10) AS_TK(11) AS_TK(12) AS_TK(13) AS_TK(14) AS_TK(15)
#endif
#undef AS_TK
      default: return u32x2{0u, 0u};
    }
  };

  f32x4 acc[AS_WMT][4], accp[AS_WMT][4];                // the tile being multiplied / the finished one being stored
#pragma unroll
  for (int i = 0; i < AS_WMT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) { acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; accp[i][j] = acc[i][j]; }
  bf16* __restrict__ Cout = (bf16*)p.C;
  bf16* __restrict__ aux_out = (bf16*)p.aux_out;

  // the FINISHED tile (the one in accp)
  int n0p = 0, strip_p = -1;

  bf16x4 ohold = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
  // ---- vector v = (i, r) of the finished tile: accp[i][j][r] = (A . W^T)[row(i, r)][n0p + 64 wn + 4 c_ + j]
  auto finish_vec = [&](int v, bool in_loop, int2 re, f32x4 bia) {
    const int i = v >> 2, r = v & 3;
    float val[4];
    const float rsv = __builtin_bit_cast(float, re.y);
    const int off = re.x + colq;
#pragma unroll
    for (int j = 0; j < 4; ++j) val[j] = accp[i][j][r] + bia[j];
    bf16x4 evv = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
    if constexpr (VEC) evv = __builtin_bit_cast(bf16x4, take_vec(v, in_loop));
    bf16* const dst = (ASTAT_ABLATE & 128) ? Cout + ((n0p + off) & 0xFFFFC) : Cout + n0p + off;      // (128: every store inside 2 MB: the L2 absorbs them)
    if constexpr (act_fwd) {
      bf16x4 z;
#pragma unroll
      for (int j = 0; j < 4; ++j) z[j] = (bf16)val[j];
      if constexpr (ACT == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) val[j] = silu_f((float)z[j]);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) val[j] = gelu_f((float)z[j]);
      }
      if constexpr (AUX) as_store4<true>(aux_out + n0p + off, z);      // (non-temporal for every ASTAT_NT != 0: read next by the backward only)
    } else if constexpr (act_bwd) {
      if constexpr (ACT == 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) val[j] *= dsilu_f((float)evv[j]);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) val[j] *= dgelu_f((float)evv[j]);
      }
    }
    bf16x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float rvj = (VEC && !act_bwd) ? (float)evv[j] : 0.f;      // (a residual next to act' is not routed here: gemm_astat_ok)
      o[j] = (bf16)(val[j] * rsv + rvj);
    }
    if constexpr ((ASTAT_ABLATE & 256) != 0) {       // timing probe: the same bytes as HALF as many 16-byte stores (rows 2 rp + (c_ & 1), columns 8 (c_ >> 1) ..: values misplaced)
      if ((v & 1) == 0) ohold = o;
      else {
        bf16x8 w8;
#pragma unroll
        for (int j = 0; j < 4; ++j) { w8[j] = ohold[j]; w8[4 + j] = o[j]; }
        bf16* d8 = Cout + n0p + row_entry(strip_p, (v & ~1) | (c_ & 1)).x + wn * 64 + (c_ >> 1) * 8;
        *reinterpret_cast<bf16x8*>(d8) = w8;
      }
    } else if constexpr ((ASTAT_ABLATE & 512) != 0) {  // timing probe: the stores without the epilogue arithmetic (raw accumulator bits)
      as_store4<true>(Cout + n0p + (strip_p * AS_BM + wm * WROWS + i * 16 + g_ * 4 + r) * (int)p.ldc + wn * 64 + c_ * 4,
                __builtin_bit_cast(bf16x4, u32x2{__builtin_bit_cast(unsigned, accp[i][0][r]), __builtin_bit_cast(unsigned, accp[i][1][r])}));
    } else if (!(ASTAT_ABLATE & 4) || (float)o[0] + (float)o[1] + (float)o[2] + (float)o[3] == 12345.678f) as_store4<(ASTAT_NT == 1 && !RESID) || ASTAT_NT == 2 || (ASTAT_NT == 3 && ACT != 0 && !RESID)>(dst, o);      // (ASTAT_NT: 1 every output but the residual stream | 2 all | 3 only the ff-wide outputs of the activation kinds | 4 only z | 0 none -- tools/r6/build_variant.sh A/B)
  };

  if constexpr (VEC) {
#pragma unroll
    for (int v = 0; v < NV; ++v) request_vec(v, row_entry(strip, v).x + colq, tn * AS_BN - ((sc.ragged && tn == ntn - 1 && wn == 1) ? 64 : 0));
  }
  int stage = 0;
#if ASTAT_TRACE
  int qc = 0;
#endif
  // one tile period: NKT k-steps of tile (strip, tn); EPI: with the previous tile's epilogue behind the MFMAs (a compile-time flag:
  // as a run-time branch it splits every k-step into basic blocks and the MFMAs cannot be scheduled among the epilogue's VALU work)
  auto tile_period = [&](auto EPI, bool last_of_strip) {
    const int n0 = tn * AS_BN - ((sc.ragged && tn == ntn - 1 && wn == 1) ? 64 : 0);     // (ragged last tile: see AstatSched)
    if (last_of_strip) {                                    // the next strip's row tables (see srow / srowa), one row per lane
      if (lane < AS_BM / AS_NCW) fill_row(strip + 1, wave * (AS_BM / AS_NCW) + lane);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (written before this wave reaches the barrier below)
    }
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      AS_STAMP(1, 0, qc);
      if (!(ASTAT_ABLATE & 16)) __builtin_amdgcn_s_barrier();                       // barrier q (see the request waves)
      AS_STAMP(1, 1, qc);
      const unsigned char* la = sa + kt * AS_KT_BYTES;
      const unsigned char* lb = sw + stage * AS_KT_BYTES;
      // row entries and bias of the epilogue vectors of this k-step, requested with the fragments (one lgkmcnt wait for all)
      int2 rep[NV];
      int roc[NV];
      f32x4 bia = {0.f, 0.f, 0.f, 0.f};
      if constexpr (decltype(EPI)::value && !(ASTAT_ABLATE & 32)) {
        bia = *reinterpret_cast<const f32x4*>(sbias + n0p * bias_on + colq);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          if ((v * NKT) / NV != kt) continue;
          rep[v] = row_entry(strip_p, v);
          if constexpr (VEC) roc[v] = row_entry(strip, v).x;
        }
      }
      // every fragment of the k-step is requested now, behind the table entries (the LDS answers in order) ...
      Vec8<bf16> fa[2][AS_WMT], fb[2][4];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int i = 0; i < AS_WMT; ++i) {
          const int r = wm * WROWS + i * 16 + c_;
          if (!(ASTAT_ABLATE & 8)) fa[ks][i] = load8<bf16>(reinterpret_cast<const bf16*>(la + r * ROWB + (((ks * 4 + g_) ^ (r & 7)) << 4)));
          else fa[ks][i] = vec8_zero<bf16>();
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = wn * 64 + j * 16 + c_;
          if (!(ASTAT_ABLATE & 8)) fb[ks][j] = load8<bf16>(reinterpret_cast<const bf16*>(lb + r * ROWB + (((ks * 4 + g_) ^ (r & 7)) << 4)));
          else fb[ks][j] = vec8_zero<bf16>();
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // ... and while the 96 KB of fragments of the eight waves come out of the LDS (~400 cycles in which the matrix pipe has nothing to
      // do: every wave left the barrier together) this k-step's share of the PREVIOUS tile's epilogue runs: it needs the table entries
      // only.  Each vector is followed by this tile's request for the same register pair.
      if constexpr (decltype(EPI)::value && !(ASTAT_ABLATE & 32)) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          if ((v * NKT) / NV != kt) continue;              // (spread over ALL k-steps of the period)
          finish_vec(v, true, rep[v], bia);
          if constexpr (VEC) request_vec(v, roc[v] + colq, n0);
        }
      }
      if (!(ASTAT_ABLATE & 1)) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int i = 0; i < AS_WMT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) mma16(fa[ks][i], fb[ks][j], acc[i][j]);
      } else {
        asm volatile("" ::"v"(fa[0][0].v), "v"(fa[1][1].v), "v"(fb[0][0].v), "v"(fb[0][1].v), "v"(fb[1][2].v), "v"(fb[1][3].v));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // this wave's reads of the k-step are done before it reaches the next barrier
      AS_STAMP(1, 2, qc);
      if (kt == NKT - 1) {
#pragma unroll
        for (int i = 0; i < AS_WMT; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) { accp[i][j] = acc[i][j]; acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        strip_p = strip;
        n0p = n0;
      }
      stage = stage + 1 == AS_NSW ? 0 : stage + 1;
#if ASTAT_TRACE
      qc += 1;
#endif
    }
    tn += 1;
    if (tn == ntn) { tn = 0; strip += 1; }
  };
  tile_period(std::false_type{}, tn == ntn - 1 && t0 + 1 < t1);
  for (int t = t0 + 1; t < t1; ++t) tile_period(std::true_type{}, tn == ntn - 1 && t + 1 < t1);
  // ---- the last tile's epilogue
  {
    const f32x4 bia = *reinterpret_cast<const f32x4*>(sbias + n0p * bias_on + colq);
    int2 rep[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) rep[v] = row_entry(strip_p, v);
#pragma unroll
    for (int v = 0; v < NV; ++v) finish_vec(v, false, rep[v], bia);
  }
#if ASTAT_TRACE
  if (tr_on) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); for (int i = lane; i < 4 * AS_TRACE_STEPS; i += 64) as_trace_buf[1][0][i] = strace[4 * AS_TRACE_STEPS + i]; }
#endif
}


// ------------------------------------------------------------------------------------------------------------------ host side
static size_t astat_smem(int K) { return (size_t)(K / 64 + AS_NSW) * AS_KT_BYTES; }     // dynamic part: A slots + weight ring (the tables are static)

static int astat_cus() { return vtx_cu_count_cached(); }

bool gemm_astat_ok(const GemmArgs& a) {
  const int mode = vtx_opt(VTX_OPT_GEMM_ASTAT);
  if (mode == 0) return false;
  if (a.K % 64 != 0 || a.K < 192 || a.K > 384 || a.N % 64 != 0 || a.N < 256 || (a.N > AS_MAXN && a.bias != nullptr)) return false;
  // N % 128 == 64 (a ragged last column tile, see AstatSched): plain / bias-only launches (the stage-2 qkv forward of Swin-S, N = 576)
  if (a.N % 128 != 0 && (a.resid != nullptr || a.aux_in != nullptr || a.aux_out != nullptr || a.act != 0 || a.perm != nullptr)) return false;
  if ((a.lda % 8) || (a.ldb % 8) || (a.ldc % 8)) return false;
  if ((int64_t)a.M * a.ldc >= (1ll << 30) || (int64_t)a.M * a.lda >= (1ll << 30)) return false;    // 32-bit byte offsets
  if (a.kscale != nullptr || a.ksum_out != nullptr) return false;
  if ((a.act == 2 || a.act == 4) && a.resid != nullptr) return false;     // (no hot-path launch has both)
  if ((a.act == 1 || a.act == 3) && a.resid != nullptr) return false;
  // in place + a partial last strip: the duplicated last row is stored by several lanes and waves, and a duplicate in another
  // wave may read an operand that aliases C (resid, z of act', A) after the row was already stored
  if ((a.resid == a.C || a.aux_in == a.C || a.aux_out == a.C || a.A == a.C) && a.M % AS_BM != 0) return false;
  const long rows = a.perm != nullptr ? a.Mk : a.M;
  if (rows <= 0) return false;
  const long tiles = (rows + AS_BM - 1) / AS_BM * ((a.N + AS_BN - 1) / AS_BN);
  if (mode != 2 && 4 * tiles < (mode == 3 ? 5L : 8L) * astat_cus()) return false;      // under two tiles per CU (mode 3: 1.25): the tiled kernels' job
  if (a.rowscale != nullptr && (a.rows_per_scale <= 0 || (a.M + a.rows_per_scale - 1) / a.rows_per_scale > AS_MAXS)) return false;
  if (a.perm != nullptr) {
    if (a.map_T <= 0 || a.M % a.map_T != 0 || a.M / a.map_T > AS_MAXS) return false;
    if (a.rowscale != nullptr && a.rows_per_scale != a.map_T) return false;
    if (a.Mk < a.M && a.resid == nullptr) return false;
  }
  return true;
}

template <int NKT, bool MAPPED, int ACT, bool RESID, bool AUX> static int astat_launch_k(const GemmArgs& a, hipStream_t st) {
  const size_t smem = astat_smem(a.K);
  auto kern = gemm_astat_kernel<NKT, MAPPED, ACT, RESID, AUX>;
  if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return VTX_ERR_LAUNCH;
  const int rows = MAPPED ? a.Mk : a.M;
  const int nstrips = (rows + AS_BM - 1) / AS_BM;
  AstatSched sc;
  sc.ntn = (a.N + AS_BN - 1) / AS_BN;
  sc.ragged = a.N % AS_BN != 0;
  sc.ntiles = nstrips * sc.ntn;
  sc.nwg = sc.ntiles < astat_cus() ? sc.ntiles : astat_cus();
  sc.copy_row0 = (MAPPED && a.resid != nullptr && nstrips * AS_BM < a.M) ? nstrips * AS_BM : a.M;
  hipLaunchKernelGGL(kern, dim3(sc.nwg), dim3(AS_NT), smem, st, a, sc);
  return vtx_check_launch();
}
template <int NKT, bool MAPPED> static int astat_launch_m(const GemmArgs& a, hipStream_t st) {
  const bool aux = a.aux_out != nullptr;
  switch (a.act) {
    case 0: return a.resid != nullptr ? astat_launch_k<NKT, MAPPED, 0, true, false>(a, st) : astat_launch_k<NKT, MAPPED, 0, false, false>(a, st);
    case 1: return aux ? astat_launch_k<NKT, MAPPED, 1, false, true>(a, st) : astat_launch_k<NKT, MAPPED, 1, false, false>(a, st);
    case 2: return astat_launch_k<NKT, MAPPED, 2, false, false>(a, st);
    case 3: return aux ? astat_launch_k<NKT, MAPPED, 3, false, true>(a, st) : astat_launch_k<NKT, MAPPED, 3, false, false>(a, st);
    case 4: return astat_launch_k<NKT, MAPPED, 4, false, false>(a, st);
    default: return VTX_ERR_SHAPE;
  }
}
template <int NKT> static int astat_launch_n(const GemmArgs& a, hipStream_t st) {
  return a.perm != nullptr ? astat_launch_m<NKT, true>(a, st) : astat_launch_m<NKT, false>(a, st);
}

int gemm_astat_launch(const GemmArgs& a, hipStream_t st) {
  switch (a.K / 64) {
    case 3: return astat_launch_n<3>(a, st);
    case 4: return astat_launch_n<4>(a, st);
    case 5: return astat_launch_n<5>(a, st);
    case 6: return astat_launch_n<6>(a, st);
    default: return VTX_ERR_SHAPE;
  }
}
