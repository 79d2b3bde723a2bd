// Full-width strip GEMM with a two-group ("ping-pong") main loop for gfx950: C[M, N] = epilogue(A[M, K] . B[N, K]^T), bf16 in, fp32
// accumulate, N = 128 NF (NF = 3: the N = 384 layers of ViT-S/16 and Swin-S stage 3), K % 32 == 0, long contractions (K >= 768).
// Reference shapes: models/layer.py:186-196 (fc2 forward, fc1 dgrad), models/vit.py:23-25,43 / models/swin_transformer.py:128
// (qkv dgrad) -- the launches where the 128 x 128-tile kernels of gemm_glds.hip sit at 0.19-0.28 of the MFMA peak and the
// vendor's plain GEMM is faster.
//
// Why another structure.  What bounds the tiled kernels (profiles/round4_astat_gemm.md) is neither HBM nor the MFMA pipe but the
// per-k-step chain barrier -> fragment reads -> MFMAs -> barrier that all waves of a workgroup walk in lockstep, plus L2 -> LDS
// bytes: a 128 x 128 tile pulls 32 KB per 512 MFMA cycles, 1.5x what the DMA path delivers per CU.  Here
//   * ONE workgroup per CU owns a strip of BM = 16 WMF rows (WMF = 4 .. 8) over ALL N columns: A is read from HBM exactly once,
//     the weight (L2-resident) once per strip; L2 -> LDS bytes per MFMA cycle are 0.64x those of the 128 x 128 tile;
//   * eight waves, each all BM rows x 16 NF columns (WMF x NF accumulator tiles, 21-24 MFMAs per 10-11 fragment reads);
//   * the waves form two groups (waves 0-3 / 4-7: one wave of each group per SIMD) that run HALF A K-STEP APART: while one group
//     multiplies k-step u out of registers, the other reads its fragments of k-step u from LDS -- the matrix pipe of every SIMD
//     always has one wave in its MFMA segment and the LDS one wave in its read segment; one s_barrier per phase (two per k-step);
//   * operands arrive by LDS-DMA (global_load_lds_dwordx4) into a ring of NS 32-deep k-steps (A rows | B rows, 64-byte rows,
//     16-byte chunk q of row r in slot q ^ {0,3,2,1}[(r >> 2) & 3]: conflict-free ds_read_b128 fragments), requested NS - 1
//     k-steps ahead from inside the MFMA segments and retired by counted vmcnt waits -- the fragment reads are inline asm, so
//     hipcc never sees an LDS read next to an outstanding DMA (it would drain the queue with vmcnt(0)).
// Element values: the same products in the same k order and the same epilogue expression per element as gemm_glds_pv_kernel /
// gemm_astat_kernel -- the three are bitwise interchangeable (tests/test_gpu_dispatch.py).
#include <stdlib.h>
#include <utility>

#include "gemm_common.h"
#include "options.h"

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

namespace {

constexpr int ST_NT = 512;                 // 8 waves
constexpr int ST_ROWB = 64;                // bytes per LDS row (32 bf16)
constexpr int ST_PASS = 4;                 // 16-row accumulator tiles staged per epilogue pass

__device__ __forceinline__ int st_swz(int r) { return (4 - ((r >> 2) & 3)) & 3; }        // {0, 3, 2, 1}[(r >> 2) & 3]

template <int OFF> __device__ __forceinline__ void st_ds_read16(bf16x8& d, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "i"(OFF) : "memory");
}
template <int N_, int... I>
__device__ __forceinline__ void st_read_frags(bf16x8 (&f)[N_], unsigned addr, std::integer_sequence<int, I...>) {
  (st_ds_read16<I * 1024>(f[I], addr), ...);
}
__device__ __forceinline__ void st_pin1(bf16x8& x) { asm volatile("" : "+v"(x)); }
template <int N_, int... I> __device__ __forceinline__ void st_pin(bf16x8 (&f)[N_], std::integer_sequence<int, I...>) {
  (st_pin1(f[I]), ...);
}

template <int WMF, int NF, int NS> constexpr int st_smem_bytes() { return NS * (WMF * 1024 + NF * 8192); }

// logical row -> row of the row-indexed operands (stochastic-depth compaction: GemmArgs::perm)
template <bool MAPPED> __device__ __forceinline__ int st_orow(const GemmArgs& p, int row, int* smp) {
  if constexpr (!MAPPED) {
    if (smp) *smp = row / p.rows_per_scale;
    return row;
  } else {
    const int s = (int)__umulhi((unsigned)row, p.map_magic), sm = p.perm[s];
    if (smp) *smp = sm;
    return sm * p.map_T + (row - s * p.map_T);
  }
}

template <int WMF, int NF, int NS, bool MAPPED>
__global__ __launch_bounds__(ST_NT, 2) void gemm_strip_kernel(GemmArgs p) {
  constexpr int BM = 16 * WMF, BN = 128 * NF, WC = 16 * NF;            // strip rows, columns, columns per wave
  constexpr int A_BYTES = WMF * 1024, STAGE = A_BYTES + NF * 8192;
  constexpr int LPS = NF + 1;                                          // DMA instructions per wave and k-step
  static_assert(NS >= 3 && NS * STAGE <= 160 * 1024, "ring must fit the LDS");
  extern __shared__ __attribute__((aligned(16))) unsigned char st_smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wave >> 2;                                           // waves w and w + 4 share a SIMD
  const int c_ = lane & 15, g_ = lane >> 4;
  const int rows = MAPPED ? p.Mk : p.M;                                // rows this launch computes
  const int m0 = blockIdx.x * BM;
  const int S = p.K >> 5;                                              // 32-deep k-steps

  if (MAPPED && m0 >= rows) {
    // copy-only strip of a mapped launch: rows of DROPPED samples (DropPath scale 0): C = resid, no operands touched
    const bf16* __restrict__ rs = (const bf16*)p.resid;
    bf16* __restrict__ cd = (bf16*)p.C;
    if (rs != nullptr)
      for (int v = threadIdx.x; v < BM * (BN / 8); v += ST_NT) {
        const int lrow = m0 + v / (BN / 8), col = (v % (BN / 8)) * 8;
        if (lrow < p.M) {
          const int64_t off = (int64_t)st_orow<MAPPED>(p, lrow, nullptr) * p.ldc + col;
          store8<bf16>(cd + off, load8<bf16>(rs + off));
        }
      }
    return;
  }

  // ---- DMA sources: one instruction = 16 rows x 64 B (lane: row lane / 4, chunk lane % 4, swizzled on the source side)
  const int lr = lane >> 2, sw = (lane & 3) ^ st_swz(lr);
  const int ablk = wave < WMF ? wave : WMF - 1;                       // waves WMF .. 7 repeat the last A block (same bytes)
  const bf16* pA;
  {
    // rows past the computed ones are never stored: any valid, finite row will do -- the last computed one
    const int r = min(m0 + ablk * 16 + lr, rows - 1);
    pA = (const bf16*)p.A + (int64_t)st_orow<MAPPED>(p, r, nullptr) * p.lda + (sw << 3);
  }
  const bf16* pB[NF];
#pragma unroll
  for (int j = 0; j < NF; ++j) pB[j] = (const bf16*)p.B + (int64_t)(wave * WC + j * 16 + lr) * p.ldb + (sw << 3);

  auto issue = [&](int sl) {                                           // the next k-step of this wave's pieces -> ring slot sl
    unsigned char* base = st_smem + sl * STAGE;
    __builtin_amdgcn_global_load_lds((gbl_void_t*)pA, (lds_void_t*)(base + ablk * 1024), 16, 0, 0);
    pA += 32;
#pragma unroll
    for (int j = 0; j < NF; ++j) {
      __builtin_amdgcn_global_load_lds((gbl_void_t*)pB[j], (lds_void_t*)(base + A_BYTES + (wave * NF + j) * 1024), 16, 0, 0);
      pB[j] += 32;
    }
  };

  // ---- fragment addresses (LDS byte addresses; + slot * STAGE per k-step, + 1024 per 16-row tile as an immediate)
  const unsigned lds0 = (unsigned)(size_t)st_smem;
  const unsigned fragA = lds0 + c_ * ST_ROWB + ((g_ ^ st_swz(c_)) << 4);
  const unsigned fragB = fragA + A_BYTES + wave * (NF * 1024);

  f32x4 acc[WMF][NF];
#pragma unroll
  for (int i = 0; i < WMF; ++i)
#pragma unroll
    for (int j = 0; j < NF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- prologue: k-steps 0 .. NS - 2 by everybody, NS - 1 by group 1 (group 0 requests it in its first MFMA segment)
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) issue(s);
  if (grp == 1) {
    issue(NS - 1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(LPS * (NS - 1)) : "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(LPS * (NS - 2)) : "memory");
  }
  __builtin_amdgcn_s_barrier();                                        // k-step 0 is in LDS
  if (grp == 1) __builtin_amdgcn_s_barrier();                          // group 1 sits out phase 0

  int rs = 0;                                                          // ring slot of k-step u
  int is = grp ? 0 : NS - 1;                                           // ring slot of the next request
  int nxt = grp ? NS : NS - 1;                                         // next k-step to request
  bf16x8 fa[WMF], fb[NF];
  for (int u = 0; u < S; ++u) {
    // ---------------- read segment: fragments of k-step u (the partner group multiplies meanwhile)
    const bool more = u + 1 < S, full = u + NS <= S;
    {
      const unsigned so = (unsigned)(rs * STAGE);
      st_read_frags(fb, fragB + so, std::make_integer_sequence<int, NF>{});
      st_read_frags(fa, fragA + so, std::make_integer_sequence<int, WMF>{});
    }
    if (grp == 1 && more) {                                            // k-step u + 1 of this wave's pieces has landed
      if (full) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(LPS * (NS - 2)) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    st_pin(fb, std::make_integer_sequence<int, NF>{});
    st_pin(fa, std::make_integer_sequence<int, WMF>{});
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---------------- MFMA segment (the partner group reads its fragments meanwhile)
    if (nxt < S) {                                                     // group 0: k-step u + NS - 1, group 1: k-step u + NS
      issue(is);
      ++nxt;
      is = is + 1 == NS ? 0 : is + 1;
    }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < WMF; ++i)
#pragma unroll
      for (int j = 0; j < NF; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    if (grp == 0 && more) {
      if (full) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(LPS * (NS - 2)) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    rs = rs + 1 == NS ? 0 : rs + 1;
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();                          // group 0 sits out the last phase
  // (no DMA is outstanding and nobody reads the ring any more: it becomes the staging buffer of the epilogue)

  // ---------------- epilogue: ST_PASS x 16 rows per pass through LDS as fp32 (acc + bias), stored as whole 768-byte rows
  // acc[i][j][r] = C[m0 + 16 i + 4 g + r][wave WC + 16 j + c]
  constexpr int CSTR = BN + 4;                                         // floats per staged row
  constexpr int VROW = BN / 8;                                         // 8-element vectors per row
  static_assert(ST_PASS * 16 * CSTR * 4 <= NS * STAGE, "epilogue staging must fit the ring");
  float* cbuf = reinterpret_cast<float*>(st_smem);
  bf16* __restrict__ Cout = (bf16*)p.C;
  const bf16* __restrict__ resid = (const bf16*)p.resid;
  const bf16* __restrict__ aux_in = (const bf16*)p.aux_in;
  bf16* __restrict__ aux_out = (bf16*)p.aux_out;
  const bool act_fwd = p.act == 1 || p.act == 3, act_bwd = p.act == 2 || p.act == 4;
  float bcol[NF];
#pragma unroll
  for (int j = 0; j < NF; ++j) bcol[j] = p.bias ? p.bias[wave * WC + j * 16 + c_] : 0.f;

#pragma unroll
  for (int i0 = 0; i0 < WMF; i0 += ST_PASS) {
    const int ni = WMF - i0 < ST_PASS ? WMF - i0 : ST_PASS;            // tiles of this pass
    if (i0) __syncthreads();                                           // the previous pass has been read
#pragma unroll
    for (int ii = 0; ii < ST_PASS; ++ii) {
      if (i0 + ii < WMF) {
#pragma unroll
        for (int j = 0; j < NF; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            cbuf[(ii * 16 + g_ * 4 + r) * CSTR + wave * WC + j * 16 + c_] = acc[i0 + ii < WMF ? i0 + ii : 0][j][r] + bcol[j];
      }
    }
    __syncthreads();
    const int nvec = ni * 16 * VROW;
    for (int v = threadIdx.x; v < nvec; v += ST_NT) {
      const int lr2 = v / VROW, cv = v - lr2 * VROW;
      const int lrow = m0 + i0 * 16 + lr2;
      if (lrow >= p.M) continue;
      int srow;
      const int64_t off = (int64_t)st_orow<MAPPED>(p, lrow, &srow) * p.ldc + cv * 8;
      if (MAPPED && lrow >= rows) {                                    // a dropped sample's row inside the last computed strip
        if (resid) store8<bf16>(Cout + off, load8<bf16>(resid + off));
        continue;
      }
      const float* cp = cbuf + lr2 * CSTR + cv * 8;
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

int st_cus() { return vtx_cu_count_cached(); }

template <int WMF, int NF, int NS, bool MAPPED> int st_launch_k(const GemmArgs& a, hipStream_t st) {
  constexpr int smem = st_smem_bytes<WMF, NF, NS>();
  auto kern = gemm_strip_kernel<WMF, NF, NS, MAPPED>;
  if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) return VTX_ERR_LAUNCH;
  // (a mapped launch with dropped samples and a residual: copy-only strips behind the computed ones)
  const int rows = (MAPPED && (a.Mk == a.M || a.resid == nullptr)) ? a.Mk : a.M;
  hipLaunchKernelGGL(kern, dim3((rows + 16 * WMF - 1) / (16 * WMF)), dim3(ST_NT), smem, st, a);
  return vtx_check_launch();
}
template <int WMF, int NF, int NS> int st_launch_m(const GemmArgs& a, hipStream_t st) {
  return a.perm != nullptr ? st_launch_k<WMF, NF, NS, true>(a, st) : st_launch_k<WMF, NF, NS, false>(a, st);
}
template <int NF, int NS> int st_launch_w(const GemmArgs& a, int wmf, hipStream_t st) {
  switch (wmf) {
    case 4: return st_launch_m<4, NF, NS>(a, st);
    case 5: return st_launch_m<5, NF, NS>(a, st);
    case 6: return st_launch_m<6, NF, NS>(a, st);
    case 7: return st_launch_m<7, NF, NS>(a, st);
    case 8: return st_launch_m<8, NF, NS>(a, st);
    default: return VTX_ERR_SHAPE;
  }
}

// Strip height: the fewest rounds of one-workgroup-per-CU strips, then the smallest strips that still fit that many rounds
// (a strip's time grows with its rows; its weight traffic does not shrink with them, so below one round nothing is gained).
int st_pick_wmf(int rows) {
  const int cus = st_cus();
  int best = 8, best_cost = 1 << 30;
  for (int w = 8; w >= 4; --w) {
    const int tiles = (rows + 16 * w - 1) / (16 * w);
    const int rounds = (tiles + cus - 1) / cus;
    const int cost = rounds * (w + 2);                                 // (+2: prologue / epilogue of a strip in 16-row units)
    if (cost < best_cost) { best_cost = cost; best = w; }
  }
  return best;
}

}  // namespace

bool gemm_strip_ok(const GemmArgs& a) {
  const int mode = vtx_opt(VTX_OPT_GEMM_STRIP);
  if (mode == 0) return false;
  if (a.N != 384 || a.K % 32 != 0 || a.K < (mode >= 2 ? 256 : 768)) return false;
  if ((a.lda % 8) || (a.ldb % 8) || (a.ldc % 8)) return false;
  if (a.kscale != nullptr || a.ksum_out != nullptr) return false;
  if ((a.act == 2 || a.act == 4) && a.aux_in == nullptr) return false;
  const long rows = a.perm != nullptr ? a.Mk : a.M;
  if (rows <= 0) return false;
  if (mode < 2 && rows < 64L * st_cus()) return false;                 // under a CU-filling round of the smallest strips: the tiled kernels' job
  if (a.perm != nullptr) {
    if (a.map_T <= 0 || a.M % a.map_T != 0) return false;
    if (a.rowscale != nullptr && a.rows_per_scale != a.map_T) return false;
    if (a.Mk < a.M && a.resid == nullptr) return false;
  }
  return true;
}

int gemm_strip_launch(const GemmArgs& a, hipStream_t st) {
  const int mode = vtx_opt(VTX_OPT_GEMM_STRIP);
  int wmf = st_pick_wmf(a.perm != nullptr ? a.Mk : a.M), ns = 5;
  if (mode >= 100) { wmf = (mode / 10) % 10; ns = mode % 10; }         // forced geometry (tools / tests): 1WN -> WMF = W, NS = N
  if (ns == 4) return st_launch_w<3, 4>(a, wmf, st);
  if (ns == 5) return st_launch_w<3, 5>(a, wmf, st);
  return VTX_ERR_SHAPE;
}
