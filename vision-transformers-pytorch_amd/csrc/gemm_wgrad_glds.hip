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
#include "gemm_common.h"

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __attribute__((aligned(256))) unsigned int vtx_zero_row[128];   // 512 zero bytes

struct WgradArgs {
  const bf16* dy; const bf16* x; float* C; float* ksum_out;
  int M;            // tokens
  int N, Kin;       // dW is [N][Kin]
  int64_t ld_dy, ld_x;
  const float* rowscale; int rows_per_scale; float scale_const;   // rowscale[sample] in {0, scale_const} or null
  int kchunk;       // tokens per grid.z slice (multiple of 64)
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

// BKT tokens per k-tile, NS-stage LDS ring.  The operands are streamed from HBM (activations, read once per XCD
// through its L2), so a k-tile costs an HBM-miss latency: what matters is how many bytes each CU keeps in flight.
// NS-1 tiles are always requested ahead (counted s_waitcnt vmcnt + raw s_barrier, never a draining __syncthreads);
// nothing but the DMA uses vmcnt inside the loop: DropPath liveness comes from a per-workgroup LDS table.
constexpr int WG_MAXSAMPLES = 512;

// NW waves per workgroup: 4 (2 x 2 waves of 64 x 64) or 8 (4 x 2 waves of 32 x 64 -- half the DMA requests and MFMAs per
// wave and k-tile, twice the waves per SIMD to interleave them).
template <int BKT, int NS, int NW>
__global__ __launch_bounds__(64 * NW) void wgrad_glds_kernel(WgradArgs p) {
  constexpr int BT = 128, ROWB = 256, OPB = BKT * ROWB;   // operand tile bytes
  constexpr int STAGE = 2 * OPB;
  constexpr int NT = 64 * NW;                              // threads
  constexpr int WMT = 16 / NW;                             // 16-row tiles per wave along N: 4 (NW 4: 2 x 2 waves) | 2 (NW 8: 4 x 2)
  constexpr int RPW = BKT / NW;                            // tile rows per wave
  constexpr int IPW = RPW / 4;                             // DMA instructions per wave and operand (4 rows x 256 B each)
  constexpr int LPT = 2 * IPW;                             // DMA instructions per wave and k-tile
  static_assert(NS * STAGE >= 64 * (BT + 4) * 4, "C staging must fit");
  extern __shared__ __attribute__((aligned(16))) unsigned char wg_smem[];   // [NS][A | B]
  __shared__ unsigned char live_tab[WG_MAXSAMPLES];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;                  // wm: 0..NW/2-1 (rows wm * 16 WMT ..), wn: column half
  const int c_ = lane & 15, g_ = lane >> 4;

  const int ntk = gridDim.x, ntn = gridDim.y;
  const int nblk = ntk * ntn * gridDim.z;
  const int did = (blockIdx.z * ntn + blockIdx.y) * ntk + blockIdx.x;
  const int xq = nblk >> 3, xr = nblk & 7, xcd = did & 7;
  const int lid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (did >> 3);
  const int tk = lid % ntk;                 // Kin tile
  const int tnn = (lid / ntk) % ntn;        // N (feature) tile
  const int tz = lid / (ntk * ntn);
  const int n0 = tnn * BT, k0 = tk * BT;
  const int mbeg = tz * p.kchunk;
  const int mend = min(p.M, mbeg + p.kchunk);
  const int nkt = (mend - mbeg + BKT - 1) / BKT;

  // DropPath liveness of the samples this slice touches (host guarantees they fit the table)
  const int s0 = mbeg / p.rows_per_scale;
  if (p.rowscale != nullptr && nkt > 0) {
    const int ns = (mend - 1) / p.rows_per_scale - s0 + 1;
    for (int i = threadIdx.x; i < ns; i += NT) live_tab[i] = p.rowscale[s0 + i] != 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // DMA piece geometry: one instruction = 4 rows x 256 B; wave w owns rows RPW w .. RPW w + RPW - 1 of each operand tile
  const int prow = lane >> 4, pslot = lane & 15;
  const bf16* zero = reinterpret_cast<const bf16*>(vtx_zero_row);
  // (sample - s0, token within the sample) of this lane's rows in the NEXT tile to be requested; advanced per issue
  int smp[IPW], rem[IPW];
#pragma unroll
  for (int j = 0; j < IPW; ++j) {
    const int tok = mbeg + wave * RPW + j * 4 + prow;
    smp[j] = tok / p.rows_per_scale - s0;
    rem[j] = tok % p.rows_per_scale;
  }

  // ragged N / Kin (multiples of 8): 16-byte chunks past the operand's last column come from the zero row, so the
  // excess rows / columns of the accumulator tile are exact zeros (and are never stored)
  bool cok_a[IPW], cok_b[IPW];
#pragma unroll
  for (int j = 0; j < IPW; ++j) {
    const int q = pslot ^ wg_swz(wave * RPW + j * 4 + prow);
    cok_a[j] = n0 + (q << 3) < p.N;
    cok_b[j] = k0 + (q << 3) < p.Kin;
  }

  auto issue = [&](int kt, int buf) {          // called with kt = 0, 1, 2, ... in order
    unsigned char* sa = wg_smem + buf * STAGE + wave * RPW * ROWB;
    unsigned char* sb = sa + OPB;
#pragma unroll
    for (int j = 0; j < IPW; ++j) {
      const int r = wave * RPW + j * 4 + prow;
      const int tok = mbeg + kt * BKT + r;
      const int q = pslot ^ wg_swz(r);
      bool live = tok < mend;
      const bf16* srcb = (live && cok_b[j]) ? p.x + (int64_t)tok * p.ld_x + k0 + (q << 3) : zero + (q << 3);
      if (live && p.rowscale != nullptr) live = live_tab[smp[j]] != 0;
      const bf16* srca = (live && cok_a[j]) ? p.dy + (int64_t)tok * p.ld_dy + n0 + (q << 3) : zero + (q << 3);
      __builtin_amdgcn_global_load_lds((gbl_void_t*)srca, (lds_void_t*)(sa + j * 4 * ROWB), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gbl_void_t*)srcb, (lds_void_t*)(sb + j * 4 * ROWB), 16, 0, 0);
      if (p.rowscale != nullptr) {
        rem[j] += BKT;
        while (rem[j] >= p.rows_per_scale) { rem[j] -= p.rows_per_scale; ++smp[j]; }
      }
    }
  };

  f32x4 acc[WMT][4];
#pragma unroll
  for (int i = 0; i < WMT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // bias gradient partials: thread (chunk = tid & 15, row group = tid >> 4) sums 8 columns over rows rg, rg+16, ...
  const bool do_ksum = p.ksum_out != nullptr && tk == 0;
  float ks8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int s2 = 0; s2 < NS - 1; ++s2)
    if (s2 < nkt) issue(s2, s2);
  if (nkt >= NS - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"i"((NS - 2) * LPT) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  int buf = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    const bool refill = kt + NS - 1 < nkt;
    if (refill) issue(kt + NS - 1, buf == 0 ? NS - 1 : buf - 1);
    const unsigned char* la = wg_smem + buf * STAGE;
    const unsigned char* lb = la + OPB;
    if (do_ksum) {
      const int ch = threadIdx.x & 15, rg = threadIdx.x >> 4;
#pragma unroll
      for (int rr = 0; rr < BKT / (NT / 16); ++rr) {
        const int r = rg + rr * (NT / 16);
        Vec8<bf16> t = load8<bf16>(reinterpret_cast<const bf16*>(la + r * ROWB + ((ch ^ wg_swz(r)) << 4)));
#pragma unroll
        for (int e = 0; e < 8; ++e) ks8[e] += t.get(e);
      }
    }
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
    // tile kt+1 must have landed; up to NS-2 younger tiles stay in flight (vmcnt retires in order)
    if (refill) asm volatile("s_waitcnt vmcnt(%0)" ::"i"((NS - 2) * LPT) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    buf = buf + 1 == NS ? 0 : buf + 1;
  }

  const float sc = p.rowscale != nullptr ? p.scale_const : 1.f;
  if (do_ksum) {
    float* red = reinterpret_cast<float*>(wg_smem);         // [NT / 16 row groups][128 cols]
    const int ch = threadIdx.x & 15, rg = threadIdx.x >> 4;
#pragma unroll
    for (int e = 0; e < 8; ++e) red[rg * 128 + ch * 8 + e] = ks8[e];
    __syncthreads();
    if (threadIdx.x < 128) {
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < NT / 16; ++q) s += red[q * 128 + threadIdx.x];
      if (n0 + (int)threadIdx.x < p.N) p.ksum_out[(int64_t)tz * p.N + n0 + threadIdx.x] = s * sc;
    }
    __syncthreads();
  }
  if (sc != 1.f) {
#pragma unroll
    for (int i = 0; i < WMT; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] *= sc;
  }
  if constexpr (NW == 4) {
    GemmArgs e;
    e.C = p.C; e.M = p.N; e.N = p.Kin; e.ldc = p.Kin;
    e.bias = nullptr; e.resid = nullptr; e.rowscale = nullptr; e.rows_per_scale = 1; e.aux_out = nullptr;
    e.aux_in = nullptr; e.act = 0;
    EpiOperands<bf16, 128, 128> eo;
    eo.load(e, n0, k0, wn, c_);               // no epilogue operands here: compiles to constants
    gemm_epilogue<bf16, float, 128, 128>(e, acc, wg_smem, n0, k0, tz, wm, wn, c_, g_, eo);
  } else {
    // fp32 slab tile through LDS in two passes of 64 rows (waves wm = 2 pass, 2 pass + 1 own them), 32-byte row pieces out:
    // acc[i][j][r] = C[n0 + 32 wm + 16 i + 4 g + r][k0 + 64 wn + 16 j + c]
    constexpr int CSTR = BT + 4;
    float* cbuf = reinterpret_cast<float*>(wg_smem);
    float* Cout = p.C + (int64_t)tz * p.N * p.Kin;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      if ((wm >> 1) == pass) {
#pragma unroll
        for (int i = 0; i < WMT; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              cbuf[((wm & 1) * 32 + i * 16 + g_ * 4 + r) * CSTR + wn * 64 + j * 16 + c_] = acc[i][j][r];
      }
      __syncthreads();
#pragma unroll
      for (int it = 0; it < (64 * 16) / NT; ++it) {
        const int v = threadIdx.x + NT * it;
        const int lr = v >> 4, cv = v & 15;
        const int row = n0 + 64 * pass + lr, col = k0 + cv * 8;
        if (row < p.N && col < p.Kin) {
          const float* cp = cbuf + lr * CSTR + cv * 8;
          float* dst = Cout + (int64_t)row * p.Kin + col;
          *reinterpret_cast<f32x4*>(dst) = *reinterpret_cast<const f32x4*>(cp);
          *reinterpret_cast<f32x4*>(dst + 4) = *reinterpret_cast<const f32x4*>(cp + 4);
        }
      }
      if (pass == 0) __syncthreads();
    }
  }
}

// workgroups the chip keeps resident (256 CUs x 2: 64 KB ring each): the split-K slices are sized to it.
// Ring: 64-token k-tiles x 2 stages, the whole next tile requested at the top of an iteration.  Measured against it on
// the stage-3/4 shapes (per step): 32 tokens x 4 stages 7.48 vs 6.89 ms, 32 x 3 (3 workgroups per CU, 768 slices)
// 7.22 ms -- twice the barriers per token cost more than the deeper prefetch returns; the next tile's requests spread
// between the MFMA rows (with or without scheduling barriers) 5.11 vs 4.60 ms -- the pieces issued late land late.
int wgrad_glds_resident() { return 512; }

bool wgrad_glds_ok(int dtype, int N, int Kin, const float* rowscale, float scale_const) {
  static int on = -1;
  if (on < 0) { const char* ev = getenv("VTX_WGRAD_GLDS"); on = ev ? atoi(ev) : 1; }
  // any N, Kin that are multiples of 8 (16-byte DMA chunks) and at least half a tile wide: edge tiles are zero-filled.
  // Against the register-staged kernel on the ragged shapes (Swin-S stage 1/2, PVT-Small stage 1/3, through DropPath):
  // 5-20 % faster on every one, 1 365 -> 1 238 us summed (tools/probe/wgrad_shapes.py).
  const bool shape_ok = (N % 8) == 0 && (Kin % 8) == 0 && N >= 64 && Kin >= 64;
  return on && dtype == VTX_BF16 && shape_ok && (rowscale == nullptr || scale_const > 0.f);
}

template <int BKT, int NS, int NW> static int wgrad_glds_launch_cfg(const WgradArgs& a, int nz, hipStream_t st) {
  constexpr int smem = NS * 2 * BKT * 256;
  auto kern = wgrad_glds_kernel<BKT, NS, NW>;
  if (smem + WG_MAXSAMPLES > 64 * 1024 &&
      hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
    return VTX_ERR_LAUNCH;
  dim3 grid((a.Kin + 127) / 128, (a.N + 127) / 128, nz);
  hipLaunchKernelGGL(kern, grid, dim3(64 * NW), smem, st, a);
  return vtx_check_launch();
}

int wgrad_glds_launch(const void* dy, const void* x, float* C, float* ksum_out, int64_t mtok, int N, int Kin,
                      int64_t ld_dy, int64_t ld_x, const float* rowscale, int rows_per_scale, float scale_const,
                      int nz, int kchunk, hipStream_t st) {
  WgradArgs a;
  a.dy = (const bf16*)dy; a.x = (const bf16*)x; a.C = C; a.ksum_out = ksum_out; a.M = (int)mtok; a.N = N; a.Kin = Kin;
  a.ld_dy = ld_dy; a.ld_x = ld_x; a.rowscale = rowscale; a.rows_per_scale = rows_per_scale;
  a.scale_const = scale_const; a.kchunk = kchunk;
  if (rowscale != nullptr && kchunk / rows_per_scale + 2 > WG_MAXSAMPLES) return VTX_ERR_SHAPE;
  // 8 waves per workgroup: every shape of Swin-S / ViT-S 9-11 % faster than with 4 (stage-2..4 weight gradients 5.20 ->
  // 4.72 ms, ViT-S/16 4.66 -> 4.16 ms per step); 16 waves (one workgroup per CU: 73 registers) 5.6 vs 4.25 ms.
  // VTX_WG_WAVES=4 keeps the 2 x 2 variant for comparison
  static int nw = -1;
  if (nw < 0) { const char* e = getenv("VTX_WG_WAVES"); nw = e ? atoi(e) : 8; }
  if (nw == 4) return wgrad_glds_launch_cfg<64, 2, 4>(a, nz, st);
  return wgrad_glds_launch_cfg<64, 2, 8>(a, nz, st);
}
