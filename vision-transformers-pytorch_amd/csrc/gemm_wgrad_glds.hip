// LDS-DMA weight-gradient GEMM for gfx950 (bf16): dW[N,Kin] = c * sum_m keep[m] dy[m,N]^T x[m,Kin], fp32 out;
// N, Kin multiples of 8 and >= 64 (128 x 128 output tiles, ragged edges zero-filled).
//
// Both operands of a weight gradient are contracted over their ROW index (tokens), i.e. both are
// "transposed" for the MFMA.  The register-staged kernel (gemm.hip, TA = TB = true) transposes 4x8
// micro-tiles in VGPRs and pays ds_write for it; here the token-major tiles go HBM -> LDS untouched with
// global_load_lds_dwordx4 (coalesced 256-byte rows, double-buffered, one barrier per 64-token k-tile) and
// the transpose happens in the LDS READ: ds_read_b64_tr_b16 hands each lane 4 consecutive tokens of one
// column, which is exactly the k-slot layout mma16 wants (semantics probed on hardware, tools/probe/).
//   * LDS image [64 tokens][128 cols] bf16, 256-byte rows, 16-byte chunk q of row r stored in slot
//     q ^ (((r & 3) | ((r >> 3 & 1) << 2)) << 1): the 8 rows one tr-read cycle touches land on 8 different
//     32-byte bank groups (conflict-free), applied on the SOURCE address because LDS-DMA writes lane-linear;
//   * DropPath: the per-sample scale is {0, c}.  Dropped samples' dy rows are fetched from a zero row
//     (their tokens contribute nothing), c is applied once to the accumulators;
//   * bias gradient = column sums of the dy tile, accumulated from LDS with 16-byte reads by the blocks
//     of the first Kin tile; split-K over tokens into fp32 slabs + fixed-order reduce (deterministic).
#ifndef WG_AUX
#define WG_AUX 0            // cache-policy bits of the operand LDS-DMA (both operands are streamed once): 2 = nt -- A/B: profiles/round6_nt_load_screen.txt
#endif
#include "gemm_common.h"
#include "options.h"

// phase ablation for IN-MODEL timing (tools/probe/build_ablate.sh wgrad N): 1 no DMA requests | 2 no fragment reads + MFMA |
// 4 no epilogue (slab / output stores) | 8 no bias-gradient column sums.  0 in the library.
#ifndef WG_ABLATE
#define WG_ABLATE 0
#endif

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __attribute__((aligned(256))) unsigned int vtx_zero_row[128];   // 512 zero bytes

constexpr int WG_MAXPROB = 8;        // weight gradients per grouped launch
constexpr int WG_MAXSAMPLES = 512;   // DropPath liveness table entries (samples one split-K slice may span)

// One weight gradient of a grouped launch: dW[N][Kin] = c * sum_m keep[m] dy[m,:]^T x[m,:]
struct WgradProb {
  const bf16* dy; const bf16* x;
  float* slab;        // [nz][N][Kin] fp32 partials (nz > 1) -- unused when nz == 1
  float* out;         // final dW [N][Kin]
  float* ksum_part;   // [nz][N] bias-gradient partials (nz > 1) or null
  float* ksum_out;    // final dbias [N] or null
  const float* rowscale;   // [samples] in {0, scale_const} or null (DropPath)
  int64_t ld_dy, ld_x;
  int N, Kin;
  int ntk;            // Kin tiles (128 wide); N tiles = ceil(N / 128)
  int tile0;          // first tile id of this problem inside a slice
  int live_only;      // rowscale marks live samples only; the constant is NOT applied (dy already carries it)
  // stochastic-depth compaction: the contraction runs over the Mtok tokens of the KEPT samples only, in the order of `perm`
  // (logical token t is row perm[t / rps] * rps + t % rps of dy and x); no liveness then: every token of the loop is live
  const int* perm;    // [samples] int32 or null
  int Mtok;           // tokens of this problem (== WgradArgs::M without a map)
  float scale;        // applied to the accumulators of a mapped problem (the DropPath constant, or 1)
};

struct WgradArgs {
  WgradProb pr[WG_MAXPROB];
  int nprob;
  int M;              // tokens (shared by all problems of the launch)
  int rows_per_scale; float scale_const;
  int kchunk;         // tokens per split-K slice (multiple of 64)
  int nz;             // slices
  int ntiles;         // tiles per slice over all problems; grid = ntiles * nz workgroups
};

__device__ __forceinline__ int wg_swz(int r) { return ((r & 3) | (((r >> 3) & 1) << 2)) << 1; }

// 8 k-slots (tokens tok0 + 8g + {0..7}) of column `col` of a swizzled [64][128] tile, for lane (c, g)
__device__ __forceinline__ Vec8<bf16> wg_frag(const unsigned char* tile, int tok0, int col0, int lane) {
  const int p = lane & 15, g = lane >> 4;
  const int n = col0 + ((p & 3) << 2);                   // this lane's 4-column piece (cols n..n+3)
  s16x4 v[2];
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int r = tok0 + g * 8 + half * 4 + (p >> 2);    // row supplied by this lane
    const unsigned char* a = tile + r * 256 + ((((n >> 3) ^ wg_swz(r))) << 4) + ((n & 7) << 1);
    v[half] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a);
  }
  // whole-vector bitcasts only: element-wise short -> bf16 bit_casts of the tr-read result were miscompiled
  // (every element became element 0) by hipcc 7.2
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 w = __builtin_shufflevector(v[0], v[1], 0, 1, 2, 3, 4, 5, 6, 7);
  Vec8<bf16> f;
  f.v = __builtin_bit_cast(bf16x8, w);
  return f;
}

// The same fragment through INLINE-ASM transpose reads.  hipcc gives the ds_read_tr builtin no memory operand, so its
// waitcnt pass must assume the read may touch LDS that an in-flight LDS-DMA is still writing and puts `s_waitcnt
// vmcnt(0)` in front of the first fragment read of every k-tile: the DMA of tile kt + 1, requested a few instructions
// earlier, is drained before tile kt is multiplied -- no overlap inside a workgroup, an iteration costs the DMA latency
// PLUS the compute (found in the ISA in round 2; the plain ds_read_b128 of gemm_glds.hip are not affected).  An asm
// read is invisible to that pass; the price is doing its lgkmcnt bookkeeping by hand (wg_wait: the wait statement names
// every destination "+v", so no consumer is scheduled above it and the registers stay allocated until the data landed).
// lane_off = byte offset of the lane's 4-column piece in row (p >> 2) + 8 g of a tile (swizzle included: it depends on
// (r & 3, (r >> 3) & 1) = (p >> 2, g & 1) only, not on the k-step or the half); OFF = operand base + 8 KB k-step + 1 KB half.
__device__ __forceinline__ unsigned wg_lane_off(int col0, int lane) {
  const int p = lane & 15, g = lane >> 4;
  const int n = col0 + ((p & 3) << 2);
  const int r = g * 8 + (p >> 2);
  return (unsigned)(r * 256 + ((((n >> 3) ^ wg_swz(r))) << 4) + ((n & 7) << 1));
}
template <int OFF> __device__ __forceinline__ void wg_tr2(s16x4& lo, s16x4& hi, unsigned addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(addr), "i"(OFF));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "i"(OFF + 1024));
}
__device__ __forceinline__ Vec8<bf16> wg_join(const s16x4& lo, const s16x4& hi) {
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 w = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  Vec8<bf16> f;
  f.v = __builtin_bit_cast(bf16x8, w);
  return f;
}
// the 12 fragment halves (2 A + 4 B fragments) of k-step KS of the tile at LDS byte offset tb
template <int KS, int OPB_>
__device__ __forceinline__ void wg_read12(s16x4 (&r)[12], const unsigned (&fa)[2], const unsigned (&fb)[4], unsigned tb) {
  wg_tr2<KS * 8192>(r[0], r[1], fa[0] + tb);            wg_tr2<KS * 8192>(r[2], r[3], fa[1] + tb);
  wg_tr2<OPB_ + KS * 8192>(r[4], r[5], fb[0] + tb);     wg_tr2<OPB_ + KS * 8192>(r[6], r[7], fb[1] + tb);
  wg_tr2<OPB_ + KS * 8192>(r[8], r[9], fb[2] + tb);     wg_tr2<OPB_ + KS * 8192>(r[10], r[11], fb[3] + tb);
}
// wait until at most N LDS operations of this wave are outstanding; the 12 fragment halves of one k-step become defined
template <int N> __device__ __forceinline__ void wg_wait(s16x4 (&a)[12]) {
  asm volatile("s_waitcnt lgkmcnt(%12)"
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]),
                 "+v"(a[9]), "+v"(a[10]), "+v"(a[11])
               : "i"(N));
}
__device__ __forceinline__ void wg_mma12(const s16x4 (&r)[12], f32x4 (&acc)[2][4]) {
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) mma16(wg_join(r[2 * i], r[2 * i + 1]), wg_join(r[4 + 2 * j], r[5 + 2 * j]), acc[i][j]);
}

// BKT tokens per k-tile, NS-stage LDS ring.  The operands are streamed from HBM (activations, read once per XCD
// through its L2), so a k-tile costs an HBM-miss latency: what matters is how many bytes each CU keeps in flight.
// NS-1 tiles are always requested ahead (counted s_waitcnt vmcnt + raw s_barrier, never a draining __syncthreads);
// nothing but the DMA uses vmcnt inside the loop: DropPath liveness comes from a per-workgroup LDS table.
//
// GROUPED: one launch computes up to WG_MAXPROB weight gradients over the same tokens (the four of a transformer
// layer's backward: fc2, fc1, proj, qkv).  Split-K exists only to fill the chip (512 resident workgroups); a single
// Swin stage-3 weight gradient has 9-36 output tiles and needs 14-57 slices -- 512 fp32 slab tiles of 64 KB written and
// re-read per launch -- while the layer's four together have 108 tiles and need 4: a quarter of the slab traffic and
// of the launches, and a 4x longer k-loop per workgroup to amortise its prologue / epilogue.
//
// SPLIT-K SUM: plain fp32 slab stores; the host follows with ONE slab_reduce_multi launch for all weight and bias slabs
// of the group (slice order, deterministic).  An in-launch sum by each tile's last-arriving workgroup (write-through
// stores, ticket counters) was built in round 2, measured 27-76 us SLOWER per launch (a synchronised store-drain / ticket /
// read-back burst at the end of every launch: profiles/round2_wgrad_phase_ablation.txt) and removed in round 3.
// NW waves per workgroup: 4 (2 x 2 waves of 64 x 64) or 8 (4 x 2 waves of 32 x 64 -- half the DMA requests and MFMAs per
// wave and k-tile, twice the waves per SIMD to interleave them).
template <int BKT, int NS, int NW, bool MAPPED = false>
__global__ __launch_bounds__(64 * NW, (NW == 8 && MAPPED) ? 4 : 2) void wgrad_glds_kernel(WgradArgs p) {   // (mapped: two 64-KB workgroups per CU need <= 128 registers)
  constexpr int BT = 128, ROWB = 256, OPB = BKT * ROWB;   // operand tile bytes
  constexpr int STAGE = 2 * OPB;
  constexpr int NT = 64 * NW;                              // threads
  constexpr int WMT = 16 / NW;                             // 16-row tiles per wave along N: 4 (NW 4: 2 x 2 waves) | 2 (NW 8: 4 x 2)
  constexpr int RPW = BKT / NW;                            // tile rows per wave
  constexpr int IPW = RPW / 4;                             // DMA instructions per wave and operand (4 rows x 256 B each)
  constexpr int LPT = 2 * IPW;                             // DMA instructions per wave and k-tile
  constexpr int RING = NS * STAGE;
  static_assert(RING >= 64 * (BT + 4) * 4, "C staging must fit");
  // ONE shared array (a second __shared__ object makes hipcc drain vmcnt before the first ds_read of every k-step):
  // [NS][A | B] ring, then the DropPath liveness table
  extern __shared__ __attribute__((aligned(16))) unsigned char wg_smem[];
  unsigned char* live_tab = wg_smem + RING;
  int* perm_tab = reinterpret_cast<int*>(wg_smem + RING + WG_MAXSAMPLES);   // [WG_MAXSAMPLES] rows-of-sample bases (mapped)

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;                  // wm: 0..NW/2-1 (rows wm * 16 WMT ..), wn: column half
  const int c_ = lane & 15, g_ = lane >> 4;

  // workgroup -> (slice, problem, tile): XCD-contiguous remap, then slice-major so that the tiles of one slice (which
  // stream the same token rows) sit on the same XCD
  const int nblk = gridDim.x;
  const int did = blockIdx.x;
  const int xq = nblk >> 3, xr = nblk & 7, xcd = did & 7;
  const int lid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (did >> 3);
  const int tz = lid / p.ntiles;
  const int tile = lid - tz * p.ntiles;
  int pi = 0;
#pragma unroll
  for (int i = 1; i < WG_MAXPROB; ++i)
    if (i < p.nprob && tile >= p.pr[i].tile0) pi = i;
  const WgradProb& q = p.pr[pi];
  const int lt = tile - q.tile0;
  const int tk = lt % q.ntk, tnn = lt / q.ntk;
  const int n0 = tnn * BT, k0 = tk * BT;
  const int N = q.N, Kin = q.Kin;
  const bf16* __restrict__ gdy = q.dy;
  const bf16* __restrict__ gx = q.x;
  const int64_t ld_dy = q.ld_dy, ld_x = q.ld_x;
  const float* rowscale = q.rowscale;
  const int mbeg = tz * p.kchunk;
  const int mend = min(q.Mtok, mbeg + p.kchunk);
  const int nkt = mend > mbeg ? (mend - mbeg + BKT - 1) / BKT : 0;

  // DropPath liveness of the samples this slice touches (host guarantees they fit the table)
  const int s0 = mbeg / p.rows_per_scale;
  const int* __restrict__ gperm = q.perm;
  constexpr bool mapped = MAPPED;                           // compile-time variant: the plain kernel carries none of this
  if (mapped && nkt > 0) {
    const int ns = (mend - 1) / p.rows_per_scale - s0 + 1;
    for (int i = threadIdx.x; i < ns; i += NT) perm_tab[i] = gperm[s0 + i] * p.rows_per_scale;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (rowscale != nullptr && nkt > 0) {
    const int ns = (mend - 1) / p.rows_per_scale - s0 + 1;
    for (int i = threadIdx.x; i < ns; i += NT) live_tab[i] = rowscale[s0 + i] != 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // DMA piece geometry: one instruction = 4 rows x 256 B; wave w owns rows RPW w .. RPW w + RPW - 1 of each operand tile.
  //
  // The issue path runs once per k-tile in EVERY wave, next to 16 MFMAs: its instruction count is what bounds the loop
  // (profiles/round2_wgrad_phase_ablation.txt: 108 of 160 us with ~150 instructions per k-tile -- 64-bit address
  // products, column / slice-end / DropPath selects, two LDS round trips for the liveness bytes).  So everything that
  // does not change from tile to tile is decided ONCE per lane: the source pointers of the FULL tiles advance by a
  // per-lane constant stride (0 for lanes whose 16-byte chunk lies past the operand's last column: they stay on the
  // zero row, so the excess rows / columns of the accumulator tile are exact zeros and are never stored); the DropPath
  // liveness byte of the next tile's rows is fetched from the LDS table one tile ahead; only the slice's partial last
  // tile (rows past `mend` come from the zero row) takes the general code.
  const int prow = lane >> 4, pslot = lane & 15;
  const bf16* zero = reinterpret_cast<const bf16*>(vtx_zero_row);
  const int rps = p.rows_per_scale;
  const bool has_rs = rowscale != nullptr;                  // wave-uniform
  const int nfull = (mend - mbeg) / BKT;                    // k-tiles that lie completely inside the slice
  const bf16* pa[IPW];                                      // next full tile's source of this lane, dy / x
  const bf16* pb[IPW];
  const bf16* pz[IPW];                                      // the lane's chunk of the zero row
  unsigned inca[IPW], incb[IPW];                            // byte stride per k-tile (0: parked on the zero row)
  int smp[IPW], rem[IPW];                                   // (sample - s0, token within the sample) of the next tile's row
  int nlive[IPW];                                           // its DropPath liveness (read one tile ahead)
#pragma unroll
  for (int j = 0; j < IPW; ++j) {
    const int r = wave * RPW + j * 4 + prow;
    const int qq = pslot ^ wg_swz(r);
    const int tok = mbeg + r;
    const bool ca = n0 + (qq << 3) < N, cb = k0 + (qq << 3) < Kin;
    pz[j] = zero + (qq << 3);
    smp[j] = tok / rps - s0;
    rem[j] = tok % rps;
    if (mapped) {
      // compacted contraction: pa / pb = the lane's COLUMN base, inca / incb = bytes per row (0: parked on the zero row),
      // nlive = the operands' row of this lane's logical token in the NEXT tile (sample base from the LDS table + offset)
      pa[j] = ca ? gdy + n0 + (qq << 3) : pz[j];
      pb[j] = cb ? gx + k0 + (qq << 3) : pz[j];
      inca[j] = ca ? (unsigned)(ld_dy * 2) : 0u;
      incb[j] = cb ? (unsigned)(ld_x * 2) : 0u;
      nlive[j] = nkt > 0 ? perm_tab[smp[j]] + rem[j] : 0;
    } else {
      pa[j] = ca ? gdy + (int64_t)tok * ld_dy + n0 + (qq << 3) : pz[j];
      pb[j] = cb ? gx + (int64_t)tok * ld_x + k0 + (qq << 3) : pz[j];
      inca[j] = ca ? (unsigned)(BKT * ld_dy * 2) : 0u;
      incb[j] = cb ? (unsigned)(BKT * ld_x * 2) : 0u;
      nlive[j] = has_rs ? live_tab[smp[j]] : 1;
    }
  }

  auto issue = [&](int kt, int buf) __attribute__((always_inline)) {          // called with kt = 0, 1, 2, ... in order
    unsigned char* sa = wg_smem + buf * STAGE + wave * RPW * ROWB;
    unsigned char* sb = sa + OPB;
    if (WG_ABLATE & 1) return;
    if (mapped && kt < nfull) {
      // every token of a compacted contraction is live: row addresses from the sample table, no zero-row redirection
#pragma unroll
      for (int j = 0; j < IPW; ++j) {
        const uint64_t ro = (uint64_t)(unsigned)nlive[j];
        const char* srca = reinterpret_cast<const char*>(pa[j]) + ro * inca[j];
        const char* srcb = reinterpret_cast<const char*>(pb[j]) + ro * incb[j];
        __builtin_amdgcn_global_load_lds((gbl_void_t*)srca, (lds_void_t*)(sa + j * 4 * ROWB), 16, 0, WG_AUX);
        __builtin_amdgcn_global_load_lds((gbl_void_t*)srcb, (lds_void_t*)(sb + j * 4 * ROWB), 16, 0, WG_AUX);
      }
#pragma unroll
      for (int j = 0; j < IPW; ++j) {
        rem[j] += BKT;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const bool w = rem[j] >= rps;
          rem[j] -= w ? rps : 0;
          smp[j] += w ? 1 : 0;
        }
        if (rps * 2 < BKT) { while (rem[j] >= rps) { rem[j] -= rps; ++smp[j]; } }
        nlive[j] = perm_tab[min(smp[j], WG_MAXSAMPLES - 1)] + rem[j];   // consumed by the NEXT call (a tile past the slice reads a stale entry: never issued)
      }
    } else if (kt < nfull) {
#pragma unroll
      for (int j = 0; j < IPW; ++j) {
        // BOTH operands of a dropped sample's rows come from the zero row: with stochastic-depth compaction the rows of
        // dropped samples are never written (0 x garbage could be 0 x NaN)
        const bf16* srca = nlive[j] ? pa[j] : pz[j];
        const bf16* srcb = nlive[j] ? pb[j] : pz[j];
        __builtin_amdgcn_global_load_lds((gbl_void_t*)srca, (lds_void_t*)(sa + j * 4 * ROWB), 16, 0, WG_AUX);
        __builtin_amdgcn_global_load_lds((gbl_void_t*)srcb, (lds_void_t*)(sb + j * 4 * ROWB), 16, 0, WG_AUX);
        pa[j] = reinterpret_cast<const bf16*>(reinterpret_cast<const char*>(pa[j]) + inca[j]);
        pb[j] = reinterpret_cast<const bf16*>(reinterpret_cast<const char*>(pb[j]) + incb[j]);
      }
      if (has_rs) {
#pragma unroll
        for (int j = 0; j < IPW; ++j) {
          rem[j] += BKT;
#pragma unroll
          for (int t = 0; t < 2; ++t) {                       // branch-free for samples of >= BKT / 2 tokens
            const bool w = rem[j] >= rps;
            rem[j] -= w ? rps : 0;
            smp[j] += w ? 1 : 0;
          }
          if (rps * 2 < BKT) { while (rem[j] >= rps) { rem[j] -= rps; ++smp[j]; } }   // (wave-uniform condition)
          nlive[j] = live_tab[smp[j]];                        // consumed by the NEXT call: the LDS latency is off the path
        }
      }
    } else {                                                  // the slice's partial last tile
#pragma unroll
      for (int j = 0; j < IPW; ++j) {
        const int r = wave * RPW + j * 4 + prow;
        const int tok = mbeg + kt * BKT + r;
        const int qq = pslot ^ wg_swz(r);
        bool live = tok < mend;
        if (live && has_rs) live = live_tab[tok / rps - s0] != 0;
        int64_t ro = tok;
        if (mapped && live) { const int sq = tok / rps; ro = (int64_t)perm_tab[sq - s0] + (tok - sq * rps); }
        const bf16* srcb = (live && k0 + (qq << 3) < Kin) ? gx + ro * ld_x + k0 + (qq << 3) : pz[j];
        const bf16* srca = (live && n0 + (qq << 3) < N) ? gdy + ro * ld_dy + n0 + (qq << 3) : pz[j];
        __builtin_amdgcn_global_load_lds((gbl_void_t*)srca, (lds_void_t*)(sa + j * 4 * ROWB), 16, 0, WG_AUX);
        __builtin_amdgcn_global_load_lds((gbl_void_t*)srcb, (lds_void_t*)(sb + j * 4 * ROWB), 16, 0, WG_AUX);
      }
    }
  };

  f32x4 acc[WMT][4];
#pragma unroll
  for (int i = 0; i < WMT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // bias gradient partials: thread (chunk = tid & 15, row group = tid >> 4) sums 8 columns over rows rg, rg+16, ...
  const bool have_ksum = q.ksum_out != nullptr && tk == 0 && !(WG_ABLATE & 8);
  float ks8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int s2 = 0; s2 < NS - 1; ++s2)
    if (s2 < nkt) issue(s2, s2);
  if (nkt >= NS - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"i"((NS - 2) * LPT) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // LDS byte addresses of this lane's fragment pieces in ring stage 0, k-step 0, half 0 (asm transpose reads, NW 8)
  const unsigned smem0 = (unsigned)(uintptr_t)(lds_void_t*)wg_smem;
  unsigned fa_off[2], fb_off[4];
#pragma unroll
  for (int i = 0; i < 2; ++i) fa_off[i] = smem0 + wg_lane_off(wm * 32 + i * 16, lane);
#pragma unroll
  for (int j = 0; j < 4; ++j) fb_off[j] = smem0 + wg_lane_off(wn * 64 + j * 16, lane);

  int buf = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    const bool refill = kt + NS - 1 < nkt;
    if (refill) issue(kt + NS - 1, buf == 0 ? NS - 1 : buf - 1);
    const unsigned char* la = wg_smem + buf * STAGE;
    const unsigned char* lb = la + OPB;
    if (have_ksum) {
      const int ch = threadIdx.x & 15, rg = threadIdx.x >> 4;
#pragma unroll
      for (int rr = 0; rr < BKT / (NT / 16); ++rr) {
        const int r = rg + rr * (NT / 16);
        Vec8<bf16> t = load8<bf16>(reinterpret_cast<const bf16*>(la + r * ROWB + ((ch ^ wg_swz(r)) << 4)));
#pragma unroll
        for (int e = 0; e < 8; ++e) ks8[e] += t.get(e);
      }
    }
    if constexpr ((WG_ABLATE & 2) != 0) {
    } else if constexpr (NW == 8) {
      // fragment reads issued up front (asm, see wg_tr2); the MFMAs of k-step 0 start once its 12 reads have returned
      // (LDS returns in order: the 12 younger ones of k-step 1 may still be outstanding)
      const unsigned tb = (unsigned)(buf * STAGE);
      s16x4 r0[12];
      wg_read12<0, OPB>(r0, fa_off, fb_off, tb);
      if constexpr (BKT == 64) {
        s16x4 r1[12];
        wg_read12<1, OPB>(r1, fa_off, fb_off, tb);
        wg_wait<12>(r0);
        wg_mma12(r0, reinterpret_cast<f32x4 (&)[2][4]>(acc));
        wg_wait<0>(r1);
        wg_mma12(r1, reinterpret_cast<f32x4 (&)[2][4]>(acc));
      } else {
        wg_wait<0>(r0);
        wg_mma12(r0, reinterpret_cast<f32x4 (&)[2][4]>(acc));
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < BKT / 32; ++ks) {
        Vec8<bf16> fa[WMT], fb[4];
#pragma unroll
        for (int i = 0; i < WMT; ++i) fa[i] = wg_frag(la, ks * 32, wm * (16 * WMT) + i * 16, lane);
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = wg_frag(lb, ks * 32, wn * 64 + j * 16, lane);
#pragma unroll
        for (int i = 0; i < WMT; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) mma16(fa[i], fb[j], acc[i][j]);
      }
    }
    // tile kt+1 must have landed; up to NS-2 younger tiles stay in flight (vmcnt retires in order)
    if (refill) asm volatile("s_waitcnt vmcnt(%0)" ::"i"((NS - 2) * LPT) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    buf = buf + 1 == NS ? 0 : buf + 1;
  }

  if (WG_ABLATE & 4) { if (acc[0][0][0] == 12345.678f) q.out[threadIdx.x] = acc[1][3][2]; return; }
  const float sc = mapped ? q.scale : ((rowscale != nullptr && !q.live_only) ? p.scale_const : 1.f);
  const bool split = p.nz > 1;                                // wave-uniform (kernel argument)
  if (have_ksum) {
    float* red = reinterpret_cast<float*>(wg_smem);         // [NT / 16 row groups][128 cols]
    const int ch = threadIdx.x & 15, rg = threadIdx.x >> 4;
#pragma unroll
    for (int e = 0; e < 8; ++e) red[rg * 128 + ch * 8 + e] = ks8[e];
    __syncthreads();
    if (threadIdx.x < 128) {
      float s = 0.f;
#pragma unroll
      for (int qd = 0; qd < NT / 16; ++qd) s += red[qd * 128 + threadIdx.x];
      if (n0 + (int)threadIdx.x < N) {
        if (split) q.ksum_part[(int64_t)tz * N + n0 + threadIdx.x] = s * sc;
        else q.ksum_out[n0 + threadIdx.x] = s * sc;
      }
    }
    __syncthreads();
  }
  if (sc != 1.f) {
#pragma unroll
    for (int i = 0; i < WMT; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] *= sc;
  }
  // fp32 tile through LDS in two passes of 64 rows, 32-byte row pieces out:
  // acc[i][j][r] = C[n0 + 16 WMT wm + 16 i + 4 g + r][k0 + 64 wn + 16 j + c]
  constexpr int CSTR = BT + 4;
  constexpr int RW = 16 * WMT;                    // rows per wave row: 64 (NW 4) | 32 (NW 8)
  constexpr int WPP = 64 / RW;                    // wave rows per pass
  float* cbuf = reinterpret_cast<float*>(wg_smem);
  float* Cout = split ? q.slab + (int64_t)tz * N * Kin : q.out;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    if (wm / WPP == pass) {
#pragma unroll
      for (int i = 0; i < WMT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            cbuf[((wm % WPP) * RW + i * 16 + g_ * 4 + r) * CSTR + wn * 64 + j * 16 + c_] = acc[i][j][r];
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < (64 * 16) / NT; ++it) {
      const int v = threadIdx.x + NT * it;
      const int lr = v >> 4, cv = v & 15;
      const int row = n0 + 64 * pass + lr, col = k0 + cv * 8;
      if (row < N && col < Kin) {
        const float* cp = cbuf + lr * CSTR + cv * 8;
        float* dst = Cout + (int64_t)row * Kin + col;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(cp), hi = *reinterpret_cast<const f32x4*>(cp + 4);
        *reinterpret_cast<f32x4*>(dst) = lo;
        *reinterpret_cast<f32x4*>(dst + 4) = hi;
      }
    }
    if (pass == 0) __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// WIDE TILES (round 4): 128 (N) x 384 (Kin) output tiles, one workgroup per CU.
//
// What bounds the 128 x 128 kernel above is not HBM but what every CU pulls through its L2 port and its LDS per unit of
// MFMA work: a layer's four weight gradients at C = 384 are 108 tiles that each stream 256 columns of the operands, 4.5x the
// unique bytes (1.1 GB of L2 -> LDS traffic per Swin-S stage-3 layer, 2.8 GB per ViT-S/16 layer; the DMA stream is issue bound
// at ~23 cycles per 1-KB instruction, profiles/round4_l2_dma_rate_probe.txt), 24 transpose reads per 16 MFMAs, and every
// wave both requests and multiplies.  Here a tile is 128 columns of dy x 384 columns of x (every C = 384 layer: 9 + 3 + 12
// + 12 = 36 whole tiles, no ragged edge), 8 multiplying waves of 64 x 96 (20 transpose reads per 24 MFMAs) and 4 waves that
// only request: per MFMA 0.67x the L2 -> LDS bytes and 0.55x the LDS reads.  32-token k-steps through a 4-stage ring of 32 KB
// (three k-steps = 96 KB per CU in flight: the first workgroup of an XCD to touch a line pays an HBM miss), one s_barrier per
// k-step; the multiplying waves touch no vector memory inside the loop (their transpose reads need no hand-written waits).
// x fragments are the MFMA's row operand: a lane ends with 4 consecutive Kin columns of one dW row -- 16-byte stores straight
// from the accumulators, no staging.  Everything else (grouping, slices + slab reduce, DropPath liveness, compaction row map,
// bias-gradient column sums) as above; slices fill 256 workgroups instead of 512.
// Measured and not kept: an L2 prefetch by the multiplying waves (one dword of every line 4 / 8 / 12 k-steps ahead, into a
// register nobody reads): ViT-S/16 layer 211 -> 278 us -- the touch pulls every line into the L1 as well and doubles the
// L2 -> CU traffic, which is what the request path is bound by (tools/r4/rejected/wgrad_wide_l2_prefetch.log).
// timing-only switches (results garbage): 1 no DMA requests | 2 no fragment reads + MFMA | 4 no slab / output stores
#ifndef WW_ABLATE
#define WW_ABLATE 0
#endif
#ifndef WW_NT_STORE
#define WW_NT_STORE 0          // 1: split-K slab stores non-temporal (A/B builds: tools/r5/build_ww_ablate.sh 0 -DWW_NT_STORE=1)
#endif
constexpr int WW_BK = 32, WW_NS = 4, WW_PANEL = WW_BK * 256, WW_STAGE = 4 * WW_PANEL, WW_RING = WW_NS * WW_STAGE;
constexpr int WW_NCW = 8, WW_NLW = 4, WW_NT = 64 * (WW_NCW + WW_NLW);
// (DMA instructions per request wave and k-step: 2 row groups x (1 + NXP) panels -- 8 for the 384-column tiles)
constexpr int WW_SMEM = WW_RING + WG_MAXSAMPLES + WG_MAXSAMPLES * 4;

__device__ __forceinline__ Vec8<bf16> ww_frag(const unsigned char* a) {
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a);
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 1024));
  return wg_join(lo, hi);
}

// TG (round 5, option WGRAD_WIDE = 2): the two row halves of the multiplying waves (waves 0-3 | 4-7: one wave of each per SIMD) run HALF A
// K-STEP APART, as in csrc/gemm_pp.hip -- while one group multiplies k-step u out of registers (24 MFMAs = 384 cycles) the other reads its
// 20 transposed fragments of k-step u from LDS, so the matrix pipe of a SIMD always has a wave in its MFMA segment; two s_barrier per
// k-step (phases 2u: group 0 reads, group 1 multiplies u - 1; 2u + 1: group 0 multiplies, group 1 reads).  A stage is last read in
// phase 2u + 1 and refilled from phase 2u + 2 on; k-step u + 1 is visible before phase 2u + 2.  Same products in the same order: same bits.
// J (round 5, last session): 16-column accumulator tiles per multiplying wave -- a workgroup's tile is 128 x 64 J columns of dW
// (J = 6: 384 | 5: 320, PVT stage 3 | 4: 256, Twins stage 3 / 512-wide stages | 3: 192, Swin stage 2) held in ceil(J / 2) x panels of
// 128 columns; a ring stage keeps its four 8-KB panel slots whatever J is.  N need not be a multiple of 128 any more: dy chunks at or
// past column N and x chunks past the tile's last column come from the zero row, rows >= N are not stored.
template <bool MAPPED, bool TG = false, int J = 6>
__global__ __launch_bounds__(WW_NT, 1) void wgrad_wide_kernel(WgradArgs p) {
  constexpr int NXP = (J + 1) / 2, KW = 64 * J, WW_LPT = 2 * (1 + NXP);
  extern __shared__ __attribute__((aligned(16))) unsigned char wg_smem[];
  unsigned char* live_tab = wg_smem + WW_RING;
  int* perm_tab = reinterpret_cast<int*>(wg_smem + WW_RING + WG_MAXSAMPLES);

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nblk = gridDim.x, did = blockIdx.x;
  const int xq = nblk >> 3, xr = nblk & 7, xcd = did & 7;
  const int lid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (did >> 3);
  const int tz = lid / p.ntiles;
  const int tile = lid - tz * p.ntiles;
  int pi = 0;
#pragma unroll
  for (int i = 1; i < WG_MAXPROB; ++i)
    if (i < p.nprob && tile >= p.pr[i].tile0) pi = i;
  const WgradProb& q = p.pr[pi];
  const int lt = tile - q.tile0;
  const int tq = lt % q.ntk, tp = lt / q.ntk;               // (ntk = Kin / KW here)
  const int n0 = tp * 128, k0 = tq * KW;
  const int N = q.N, Kin = q.Kin;
  const bf16* __restrict__ gdy = q.dy;
  const bf16* __restrict__ gx = q.x;
  const int64_t ld_dy = q.ld_dy, ld_x = q.ld_x;
  const float* rowscale = q.rowscale;
  const int mbeg = ((WW_ABLATE & 8) ? 0 : tz) * p.kchunk;   // (8: every slice streams the tokens of slice 0 -- timing only)
  const int mend = min(q.Mtok, mbeg + p.kchunk);
  const int nkt = mend > mbeg ? (mend - mbeg + WW_BK - 1) / WW_BK : 0;
  const int rps = p.rows_per_scale;
  const int s0 = mbeg / rps;
  const bool has_rs = rowscale != nullptr;
  if (MAPPED && nkt > 0) {
    const int ns = (mend - 1) / rps - s0 + 1;
    for (int i = threadIdx.x; i < ns; i += WW_NT) perm_tab[i] = q.perm[s0 + i] * rps;
  }
  if (has_rs && nkt > 0) {
    const int ns = (mend - 1) / rps - s0 + 1;
    for (int i = threadIdx.x; i < ns; i += WW_NT) live_tab[i] = rowscale[s0 + i] != 0.f;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const bool have_ksum = q.ksum_out != nullptr && tq == 0;
  float ks8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  f32x4 acc[4][J];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < J; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (wave >= WW_NCW) {
    // ---------------- request waves: wave lw owns token rows 8 lw .. 8 lw + 7 of the four panels (dy | x0 | x1 | x2) of a k-step
    const int lw = wave - WW_NCW;
    const int prow = lane >> 4, pslot = lane & 15;
    const int qq = pslot ^ (((prow & 3) | ((lw & 1) << 2)) << 1);      // wg_swz(r): r & 3 = prow, (r >> 3) & 1 = lw & 1 for both row groups
    const bf16* zero = reinterpret_cast<const bf16*>(vtx_zero_row);
    const bf16* pz = zero + (qq << 3);
    const bool a_ok = n0 + (qq << 3) < N;                                // this lane's 8 dy columns exist (N % 8 == 0)
    bool b_ok[NXP];
#pragma unroll
    for (int pn = 0; pn < NXP; ++pn) b_ok[pn] = pn * 128 + (qq << 3) < KW && k0 + pn * 128 + (qq << 3) < Kin;
    const int nfull = (mend - mbeg) / WW_BK;
    const bf16* pa[2];
    const bf16* pb[2];
    int smp[2], rem[2], nlive[2];
    const unsigned inca = MAPPED ? (unsigned)(ld_dy * 2) : (unsigned)(WW_BK * ld_dy * 2);
    const unsigned incb = MAPPED ? (unsigned)(ld_x * 2) : (unsigned)(WW_BK * ld_x * 2);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = lw * 8 + j * 4 + prow;
      const int tok = mbeg + r;
      smp[j] = tok / rps - s0;
      rem[j] = tok % rps;
      if (MAPPED) {
        pa[j] = gdy + n0 + (qq << 3);
        pb[j] = gx + k0 + (qq << 3);
        nlive[j] = nkt > 0 ? perm_tab[min(smp[j], WG_MAXSAMPLES - 1)] + rem[j] : 0;
      } else {
        pa[j] = gdy + (int64_t)tok * ld_dy + n0 + (qq << 3);
        pb[j] = gx + (int64_t)tok * ld_x + k0 + (qq << 3);
        nlive[j] = has_rs ? live_tab[min(smp[j], WG_MAXSAMPLES - 1)] : 1;
      }
    }
    auto advance = [&](int j) __attribute__((always_inline)) {
      rem[j] += WW_BK;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const bool w = rem[j] >= rps;
        rem[j] -= w ? rps : 0;
        smp[j] += w ? 1 : 0;
      }
      if (rps * 2 < WW_BK) { while (rem[j] >= rps) { rem[j] -= rps; ++smp[j]; } }
    };
    auto issue = [&](int kt, int stage) __attribute__((always_inline)) {   // called with kt = 0, 1, 2, ... in order
      unsigned char* sa = wg_smem + stage * WW_STAGE + lw * 8 * 256;
      if (WW_ABLATE & 1) return;
      if (kt < nfull) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const char* srca;
          const char* srcb;
          if (MAPPED) {
            const uint64_t ro = (uint64_t)(unsigned)nlive[j];
            srca = reinterpret_cast<const char*>(pa[j]) + ro * inca;
            srcb = reinterpret_cast<const char*>(pb[j]) + ro * incb;
          } else {
            srca = reinterpret_cast<const char*>(pa[j]);
            srcb = reinterpret_cast<const char*>(pb[j]);
          }
          const bool lv = MAPPED || nlive[j];                            // (a dropped sample's rows: the zero row in every panel)
          unsigned char* d = sa + j * 4 * 256;
          __builtin_amdgcn_global_load_lds((gbl_void_t*)((lv && a_ok) ? srca : reinterpret_cast<const char*>(pz)), (lds_void_t*)d, 16, 0, WG_AUX);
#pragma unroll
          for (int pn = 0; pn < NXP; ++pn)
            __builtin_amdgcn_global_load_lds((gbl_void_t*)((lv && b_ok[pn]) ? srcb + pn * 256 : reinterpret_cast<const char*>(pz)),
                                             (lds_void_t*)(d + (1 + pn) * WW_PANEL), 16, 0, WG_AUX);
          if (!MAPPED) {
            pa[j] = reinterpret_cast<const bf16*>(reinterpret_cast<const char*>(pa[j]) + inca);
            pb[j] = reinterpret_cast<const bf16*>(reinterpret_cast<const char*>(pb[j]) + incb);
          }
        }
        if (MAPPED) {
#pragma unroll
          for (int j = 0; j < 2; ++j) { advance(j); nlive[j] = perm_tab[min(smp[j], WG_MAXSAMPLES - 1)] + rem[j]; }
        } else if (has_rs) {
#pragma unroll
          for (int j = 0; j < 2; ++j) { advance(j); nlive[j] = live_tab[min(smp[j], WG_MAXSAMPLES - 1)]; }
        }
      } else {                                                            // the slice's partial last k-step
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int r = lw * 8 + j * 4 + prow;
          const int tok = mbeg + kt * WW_BK + r;
          bool live = tok < mend;
          if (live && has_rs) live = live_tab[tok / rps - s0] != 0;
          int64_t ro = tok;
          if (MAPPED && live) { const int sq = tok / rps; ro = (int64_t)perm_tab[sq - s0] + (tok - sq * rps); }
          const bf16* srca = (live && a_ok) ? gdy + ro * ld_dy + n0 + (qq << 3) : pz;
          const bf16* srcb = gx + ro * ld_x + k0 + (qq << 3);
          unsigned char* d = sa + j * 4 * 256;
          __builtin_amdgcn_global_load_lds((gbl_void_t*)srca, (lds_void_t*)d, 16, 0, WG_AUX);
#pragma unroll
          for (int pn = 0; pn < NXP; ++pn)
            __builtin_amdgcn_global_load_lds((gbl_void_t*)((live && b_ok[pn]) ? srcb + pn * 128 : pz), (lds_void_t*)(d + (1 + pn) * WW_PANEL), 16, 0, WG_AUX);
        }
      }
    };
#pragma unroll
    for (int s2 = 0; s2 < WW_NS - 1; ++s2)
      if (s2 < nkt) issue(s2, s2);
    if (nkt >= WW_NS - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"i"((WW_NS - 2) * WW_LPT) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int buf = 0;
    for (int kt = 0; kt < nkt; ++kt) {
      const bool refill = kt + WW_NS - 1 < nkt;
      if (refill) issue(kt + WW_NS - 1, buf == 0 ? WW_NS - 1 : buf - 1);
      if constexpr (TG) __builtin_amdgcn_s_barrier();                       // end of phase 2 kt
      // k-step kt + 1 must have landed; two younger ones stay in flight (vmcnt retires in order)
      if (refill) asm volatile("s_waitcnt vmcnt(%0)" ::"i"((WW_NS - 2) * WW_LPT) : "memory");
      else if (kt + WW_NS - 2 < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"i"((WW_NS - 3) * WW_LPT) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                                         // (TG: end of phase 2 kt + 1)
      buf = buf + 1 == WW_NS ? 0 : buf + 1;
    }
    if constexpr (TG) __builtin_amdgcn_s_barrier();                         // phase 2 nkt: group 1 multiplies its last k-step
  } else {
    // ---------------- multiplying waves: wr = row half of the dy columns (64), wc = quarter of the x columns (16 J: 96 of 384)
    const int wr = wave >> 2, wc = wave & 3;
    unsigned fp_off[4], fq_off[J];
#pragma unroll
    for (int i = 0; i < 4; ++i) fp_off[i] = wg_lane_off(wr * 64 + i * 16, lane);
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int qc = wc * (16 * J) + j * 16;
      fq_off[j] = (unsigned)((1 + (qc >> 7)) * WW_PANEL) + wg_lane_off(qc & 127, lane);
    }
    const int kch = threadIdx.x & 15, krg = threadIdx.x >> 4;             // bias-gradient sums: 16-byte chunk, token row (0..31)
    const unsigned ks_off = (unsigned)(krg * 256 + ((kch ^ wg_swz(krg)) << 4));
    __builtin_amdgcn_s_barrier();
    int buf = 0;
    if constexpr (TG) {
      if (wr == 1) __builtin_amdgcn_s_barrier();                           // group 1 sits out phase 0
      for (int kt = 0; kt < nkt; ++kt) {
        const unsigned char* st = wg_smem + buf * WW_STAGE;
        Vec8<bf16> fp[4], fq[J];
        // ---- read segment (the partner group multiplies meanwhile)
#pragma unroll
        for (int i = 0; i < 4; ++i) fp[i] = ww_frag(st + fp_off[i]);
#pragma unroll
        for (int j = 0; j < J; ++j) fq[j] = ww_frag(st + fq_off[j]);
        if (have_ksum) {
          Vec8<bf16> t = load8<bf16>(reinterpret_cast<const bf16*>(st + ks_off));
#pragma unroll
          for (int e = 0; e < 8; ++e) ks8[e] += t.get(e);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(fp[i].v));
#pragma unroll
        for (int j = 0; j < J; ++j) asm volatile("" : "+v"(fq[j].v));
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- MFMA segment (the partner group reads its fragments meanwhile)
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i) mma16(fq[j], fp[i], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        buf = buf + 1 == WW_NS ? 0 : buf + 1;
      }
      if (wr == 0) __builtin_amdgcn_s_barrier();                           // group 0 sits out the last phase
    } else {
    for (int kt = 0; kt < nkt; ++kt) {
      const unsigned char* st = wg_smem + buf * WW_STAGE;
      Vec8<bf16> fp[4], fq[J];
      if (!(WW_ABLATE & 2)) {
#pragma unroll
      for (int i = 0; i < 4; ++i) fp[i] = ww_frag(st + fp_off[i]);
#pragma unroll
      for (int j = 0; j < J; ++j) fq[j] = ww_frag(st + fq_off[j]);
#pragma unroll
      for (int j = 0; j < J; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) mma16(fq[j], fp[i], acc[i][j]);
      }
      if (have_ksum) {
        Vec8<bf16> t = load8<bf16>(reinterpret_cast<const bf16*>(st + ks_off));
#pragma unroll
        for (int e = 0; e < 8; ++e) ks8[e] += t.get(e);
      }
      __builtin_amdgcn_s_barrier();
      buf = buf + 1 == WW_NS ? 0 : buf + 1;
    }
    }
  }

  const float sc = MAPPED ? q.scale : ((has_rs && !q.live_only) ? p.scale_const : 1.f);
  const bool split = p.nz > 1;
  if (have_ksum) {
    float* red = reinterpret_cast<float*>(wg_smem);         // [32 token rows][128 cols]
    if (wave < WW_NCW) {
      const int kch = threadIdx.x & 15, krg = threadIdx.x >> 4;
#pragma unroll
      for (int e = 0; e < 8; ++e) red[krg * 128 + kch * 8 + e] = ks8[e];
    }
    __syncthreads();
    if (threadIdx.x < 128) {
      float s = 0.f;
#pragma unroll
      for (int qd = 0; qd < 32; ++qd) s += red[qd * 128 + threadIdx.x];
      if (n0 + (int)threadIdx.x < N) {
        if (split) q.ksum_part[(int64_t)tz * N + n0 + threadIdx.x] = s * sc;
        else q.ksum_out[n0 + threadIdx.x] = s * sc;
      }
    }
  }
  if (wave >= WW_NCW) return;
  if ((WW_ABLATE & 4) && acc[0][0][0] != 12345.678f) return;
  // acc[i][j][r] = dW[n0 + 64 wr + 16 i + c][k0 + 16 J wc + 16 j + 4 g + r]
  const int wr = wave >> 2, wc = wave & 3, c_ = lane & 15, g_ = lane >> 4;
  const int row0 = n0 + wr * 64 + c_;
  float* Cout = (split ? q.slab + (int64_t)tz * N * Kin : q.out) + (int64_t)row0 * Kin + k0 + wc * (16 * J) + g_ * 4;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < J; ++j) {
      f32x4 v = acc[i][j] * sc;
      vmem_guard(v);
      if (row0 + i * 16 < N) {
#if WW_NT_STORE
        if (split) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(Cout + (int64_t)(i * 16) * Kin + j * 16));
        else
#endif
        *reinterpret_cast<f32x4*>(Cout + (int64_t)(i * 16) * Kin + j * 16) = v;
      }
    }
}

// workgroups the chip keeps resident (256 CUs x 2: 64 KB ring each): the split-K slices are sized to it.
// Ring: 64-token k-tiles x 2 stages, the whole next tile requested at the top of an iteration.  Measured against it on
// the stage-3/4 shapes (per step): 32 tokens x 4 stages 7.48 vs 6.89 ms, 32 x 3 (3 workgroups per CU, 768 slices)
// 7.22 ms -- twice the barriers per token cost more than the deeper prefetch returns; the next tile's requests spread
// between the MFMA rows (with or without scheduling barriers) 5.11 vs 4.60 ms -- the pieces issued late land late.
int wgrad_glds_resident() { return 512; }
int wgrad_glds_max_problems() { return WG_MAXPROB; }

bool wgrad_glds_ok(int dtype, int N, int Kin, const float* rowscale, float scale_const) {
  const int on = vtx_opt(VTX_OPT_WGRAD_GLDS);
  // any N, Kin that are multiples of 8 (16-byte DMA chunks) and at least half a tile wide: edge tiles are zero-filled.
  // Against the register-staged kernel on the ragged shapes (Swin-S stage 1/2, PVT-Small stage 1/3, through DropPath):
  // 5-20 % faster on every one, 1 365 -> 1 238 us summed (tools/probe/wgrad_shapes.py).
  const bool shape_ok = (N % 8) == 0 && (Kin % 8) == 0 && N >= 64 && Kin >= 64;
  return on && dtype == VTX_BF16 && shape_ok && (rowscale == nullptr || scale_const > 0.f);
}

template <int BKT, int NS, int NW, bool MAPPED = false> static int wgrad_glds_launch_cfg(const WgradArgs& a, hipStream_t st) {
  constexpr int smem = NS * 2 * BKT * 256 + WG_MAXSAMPLES + (MAPPED ? WG_MAXSAMPLES * 4 : 0);
  auto kern = wgrad_glds_kernel<BKT, NS, NW, MAPPED>;
  if (smem > 64 * 1024 &&
      hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
    return VTX_ERR_LAUNCH;
  hipLaunchKernelGGL(kern, dim3(a.ntiles * a.nz), dim3(64 * NW), smem, st, a);
  return vtx_check_launch();
}

// Split-K slices of a (grouped) launch: tiles x slices should just fill ONE round of resident workgroups -- a partly
// filled second round leaves CUs idle for the whole kernel because every workgroup runs the same long k-loop.
int wgrad_glds_slices(int64_t mtok, int ntiles, bool wide) {
  int target = vtx_opt(VTX_OPT_WGRAD_BLOCKS);
  if (target <= 0) target = wgrad_glds_resident();
  if (wide) target = vtx_cu_count_cached();                // one 131-KB workgroup per CU
  int nz = target / ntiles;
  const int64_t maxz = (mtok + 255) / 256;
  if (nz > maxz) nz = (int)maxz;
  if (nz < 1) nz = 1;
  if (nz > 256) nz = 256;
  return nz;
}

int wgrad_glds_tiles(int N, int Kin) { return ((N + 127) / 128) * ((Kin + 127) / 128); }

// A group takes 128 x 64 J tiles (J = 6, 5, 4, 3: the widest whose width divides every Kin of the group) when whole slices of them fill
// >= 85 % of the CUs (C = 384 layers: 36 tiles x 7 slices = 252; a C = 768 layer's 144 tiles of 384 columns would leave 112 CUs idle and
// its 216 tiles of 256 columns 40: 128 x 128 tiles there).  N is any multiple of 8 of at least 64 (wgrad_glds_ok): the last row tile may
// be ragged.  Option WGRAD_WIDE bit 2 (value 4 | 5 | 6) keeps the round-4 rule: 384-column tiles of whole-tile problems only.
// J = 4 (256 columns: the C = 256 layers of Twins-SVT stage 3) is opt-in (bit 3, value 9): 67 -> 60 us per layer stand-alone, but the
// Twins-SVT-S step does not move with it -- +0.03, +0.08, -0.05 ms on three boxes, inside the run-to-run spread of that (host-bound)
// step, whose occasional 10.8-ms runs show up under every setting (profiles/round5_wgrad_wide_all_widths.txt).
static int wgrad_wide_rule(int nprob, const int* N, const int* Kin, int* Jout, bool r4, bool j4, int fill = 85) {
  const int cus = vtx_cu_count_cached();
  // 256-column tiles WITHOUT the opt-in: the fallback of a group whose widths divide by 384 but whose 384-column tiles miss the fill rule
  // (C = 768: 144 tiles = 56 %; 216 tiles of 256 columns = 84 %), from 75 % on -- Swin-S stage 4 131 -> 118 us per layer stand-alone,
  // the Swin-S step 14.392 -> 14.352 ms (4 x 100-step runs each, every pair ordered the same way: profiles/round5_wgrad_wide_all_widths.txt)
  bool all384 = !r4;
  for (int i = 0; i < nprob; ++i) all384 = all384 && Kin[i] % 384 == 0;
  for (int J = 6; J >= (r4 ? 6 : 3); --J) {
    if (J == 4 && !j4 && !all384) continue;
    const int need = (J == 4 && !j4 && fill > 75) ? 75 : fill;
    const int kw = 64 * J;
    int tiles = 0;
    bool ok = true;
    for (int i = 0; i < nprob && ok; ++i) {
      ok = Kin[i] % kw == 0 && N[i] % 8 == 0 && N[i] >= 64 && !(r4 && N[i] % 128);
      tiles += ((N[i] + 127) / 128) * (Kin[i] / kw);
    }
    if (!ok || tiles < 1 || tiles > cus) continue;
    if (100 * ((cus / tiles) * tiles) >= need * cus) {
      if (Jout) *Jout = J;
      return tiles;
    }
  }
  return 0;
}
int wgrad_wide_tiles_any(int nprob, const int* N, const int* Kin, int* Jout) { return wgrad_wide_rule(nprob, N, Kin, Jout, false, true, 50); }
int wgrad_wide_tiles(int nprob, const int* N, const int* Kin, int* Jout) {
  const int on = vtx_opt(VTX_OPT_WGRAD_WIDE);
  // (bits 4 .. 11: fill threshold in percent when not 0 -- probe switch, e.g. 9 + (70 << 4) = 1129)
  return on ? wgrad_wide_rule(nprob, N, Kin, Jout, (on & 4) != 0, (on & 8) != 0, ((on >> 4) & 255) ? ((on >> 4) & 255) : 85) : 0;
}

// slabs / ksum_part: nz > 1 only ([nz][N][Kin] / [nz][N] per problem, carved from the caller's workspace by the host)
template <bool MAPPED, bool TG, int J> static int wgrad_wide_launch_cfg(const WgradArgs& a, hipStream_t st) {
  auto kern = wgrad_wide_kernel<MAPPED, TG, J>;
  if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, WW_SMEM) != hipSuccess) return VTX_ERR_LAUNCH;
  hipLaunchKernelGGL(kern, dim3(a.ntiles * a.nz), dim3(WW_NT), WW_SMEM, st, a);
  return vtx_check_launch();
}

// wide: 0 = 128 x 128 tiles | J = 3 .. 6: 128 x 64 J tiles (wgrad_wide_tiles)
int wgrad_glds_group_launch(int nprob, const WgradProbHost* hp, int64_t mtok, int rows_per_scale, float scale_const,
                            int nz, int kchunk, hipStream_t st, int wide) {
  if (nprob < 1 || nprob > WG_MAXPROB) return VTX_ERR_SHAPE;
  WgradArgs a;
  int t0 = 0;
  bool any_scale = false;
  for (int i = 0; i < nprob; ++i) {
    if (wide && (wide < 3 || wide > 6 || hp[i].Kin % (64 * wide) || hp[i].N % 8)) return VTX_ERR_SHAPE;
    WgradProb& q = a.pr[i];
    q.dy = (const bf16*)hp[i].dy; q.x = (const bf16*)hp[i].x; q.slab = hp[i].slab; q.out = hp[i].out;
    q.ksum_part = hp[i].ksum_part; q.ksum_out = hp[i].ksum_out; q.rowscale = hp[i].rowscale;
    q.ld_dy = hp[i].ld_dy; q.ld_x = hp[i].ld_x; q.N = hp[i].N; q.Kin = hp[i].Kin;
    q.ntk = wide ? hp[i].Kin / (64 * wide) : (hp[i].Kin + 127) / 128; q.tile0 = t0; q.live_only = hp[i].live_only;
    q.perm = hp[i].perm; q.Mtok = hp[i].perm ? hp[i].Mtok : (int)mtok; q.scale = hp[i].scale;
    any_scale = any_scale || hp[i].perm != nullptr;          // (the sample table of a mapped problem has the same bound)
    t0 += wide ? ((hp[i].N + 127) / 128) * (hp[i].Kin / (64 * wide)) : wgrad_glds_tiles(hp[i].N, hp[i].Kin);
    any_scale = any_scale || hp[i].rowscale != nullptr;
  }
  for (int i = nprob; i < WG_MAXPROB; ++i) a.pr[i] = a.pr[0];
  a.nprob = nprob; a.M = (int)mtok; a.rows_per_scale = rows_per_scale > 0 ? rows_per_scale : 1;
  a.scale_const = scale_const; a.kchunk = kchunk; a.nz = nz; a.ntiles = t0;
  if (any_scale && kchunk / a.rows_per_scale + 2 > WG_MAXSAMPLES) return VTX_ERR_SHAPE;
  // 8 waves per workgroup: every shape of Swin-S / ViT-S 9-11 % faster than with 4 (stage-2..4 weight gradients 5.20 ->
  // 4.72 ms, ViT-S/16 4.66 -> 4.16 ms per step); 16 waves (one workgroup per CU: 73 registers) 5.6 vs 4.25 ms.
  // option WG_WAVES = 4 keeps the 2 x 2 variant for comparison
  if (wide) {
    bool wmapped = false;
    for (int i = 0; i < nprob; ++i) wmapped = wmapped || hp[i].perm != nullptr;
    for (int i = 0; i < nprob; ++i)
      if (wmapped && hp[i].perm == nullptr) return VTX_ERR_SHAPE;
    const bool tg = (vtx_opt(VTX_OPT_WGRAD_WIDE) & 3) == 2;     // two wave groups half a k-step apart (384-column tiles only)
    if (wide == 6) {
      if (tg) return wmapped ? wgrad_wide_launch_cfg<true, true, 6>(a, st) : wgrad_wide_launch_cfg<false, true, 6>(a, st);
      return wmapped ? wgrad_wide_launch_cfg<true, false, 6>(a, st) : wgrad_wide_launch_cfg<false, false, 6>(a, st);
    }
    if (wide == 5) return wmapped ? wgrad_wide_launch_cfg<true, false, 5>(a, st) : wgrad_wide_launch_cfg<false, false, 5>(a, st);
    if (wide == 4) return wmapped ? wgrad_wide_launch_cfg<true, false, 4>(a, st) : wgrad_wide_launch_cfg<false, false, 4>(a, st);
    // (row-mapped problems come from compacted layers, whose C and ff are multiples of 128: a group whose widths also divide by 192
    //  divides by 384 and has taken J = 6 -- no row-mapped 192-column instantiation)
    return wmapped ? VTX_ERR_SHAPE : wgrad_wide_launch_cfg<false, false, 3>(a, st);
  }
  if (vtx_opt(VTX_OPT_WG_WAVES) == 4) return wgrad_glds_launch_cfg<64, 2, 4>(a, st);
  // (ring geometries measured in round 2 and removed: 64 tokens x 3 stages, 32 x 3 / 4 / 5 -- all slower than 64 x 2,
  //  profiles/round2_wgrad_ring_variants.txt)
  bool mapped = false;
  for (int i = 0; i < nprob; ++i) mapped = mapped || hp[i].perm != nullptr;
  if (mapped) {
    for (int i = 0; i < nprob; ++i)
      if (hp[i].perm == nullptr) return VTX_ERR_SHAPE;        // all problems of a launch are mapped, or none
    return wgrad_glds_launch_cfg<64, 2, 8, true>(a, st);
  }
  return wgrad_glds_launch_cfg<64, 2, 8>(a, st);
}
