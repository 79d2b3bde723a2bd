// Round 5 probe: how fast can the CUs stream the ROW PANELS of a row-major bf16 operand (the A of the long-K GEMMs: M x 3072 B,
// one panel of BM rows per workgroup) from HBM into LDS with global_load_lds_dwordx4, as a function of
//   * the contiguous bytes requested per row at a time (CB = 128: one line per row and k-tile, as csrc/gemm_pp.hip does; 256 / 512 / 1024:
//     2 / 4 / 8 adjacent lines of a row requested back to back -- DRAM page locality),
//   * the bytes in flight per CU, the number of requesting waves, the number of workgroups that read the same panel (DUP = 2: the two
//     column tiles of N = 384), the panel height.
// No consumer, nothing reads the LDS.  Rotates over NSET operand sets (> the 256-MB Infinity Cache).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/astream tools/r5/astream_probe.hip && /tmp/astream
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;
extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

// One instruction: RPI rows x (1024 / RPI) bytes (RPI = 8: 8 rows x 128 B; 4: 4 x 256; 2: 2 x 512; 1: 1 x 1024).
// A "step" covers CB bytes of every row of the panel: BM * CB / 1024 instructions, dealt round-robin to the LW waves;
// a wave keeps at most DEPTH of its instructions in flight.
template <int RPI, int DEPTH>
__global__ __launch_bounds__(512) void astream_kernel(const unsigned char* __restrict__ A, int M, int rowbytes, int BM, int CB, int LW, int dup,
                                                      int ring_bytes, float* sink) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave >= LW) return;
  const int xq = blockIdx.x >> 3, xcd = blockIdx.x & 7;
  const int tm = xcd + 8 * (xq / dup);
  const int m0 = tm * BM;
  if (m0 >= M) return;
  constexpr int BPR = 1024 / RPI;                       // bytes per row and instruction
  const int lr = lane / (64 / RPI), lc = (lane % (64 / RPI)) * 16;
  const int ipr = CB / BPR;                             // instructions per row group and step
  const int groups = BM / RPI;                          // row groups per step
  const int per_step = groups * ipr;
  const int steps = rowbytes / CB;
  unsigned off = wave * 1024;                           // LDS write pointer of this wave (ring)
  int outstanding = 0;
  for (int s = 0; s < steps; ++s) {
    for (int i = wave; i < per_step; i += LW) {
      const int grp = i / ipr, piece = i - grp * ipr;
      int r = m0 + grp * RPI + lr;
      r = r < M ? r : M - 1;
      const unsigned char* src = A + (size_t)r * rowbytes + (size_t)s * CB + piece * BPR + lc;
      __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(smem + off), 16, 0, 0);
      off += LW * 1024;
      if (off >= (unsigned)ring_bytes) off = wave * 1024;
      if (++outstanding > DEPTH) { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(DEPTH) : "memory"); outstanding = DEPTH; }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (M < 0) sink[blockIdx.x] = smem[threadIdx.x];
}

// Split-miss variant (DUP = 2 only): the two workgroups of a panel each own HALF of its rows.  Per step a workgroup requests its OWN half
// LEAD steps ahead (HBM misses) and the OTHER half for the current step (the partner asked for those lines LEAD steps ago: L2 hits when the
// two run in step).  Optionally also BR rows x CB of an L2-resident panel per step (the weight stream of the GEMM).
template <int DEPTH>
__global__ __launch_bounds__(512) void asplit_kernel(const unsigned char* __restrict__ A, const unsigned char* __restrict__ Bw, int M, int rowbytes, int BM,
                                                     int CB, int lead, int split, int BR, int ring_bytes, float* sink) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int xq = blockIdx.x >> 3, xcd = blockIdx.x & 7;
  const int tm = xcd + 8 * (xq >> 1), tn = xq & 1;
  const int m0 = tm * BM;
  if (m0 >= M) return;
  const int lr = lane >> 3, lc = (lane & 7) * 16;
  const int ipr = CB / 128, half_groups = BM / 16;      // row groups of 8 per half
  const int steps = rowbytes / CB;
  unsigned off = wave * 1024;
  int outstanding = 0;
  auto req = [&](const unsigned char* src) {
    __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(smem + off), 16, 0, 0);
    off += 8 * 1024;
    if (off >= (unsigned)ring_bytes) off = wave * 1024;
    if (++outstanding > DEPTH) { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(DEPTH) : "memory"); outstanding = DEPTH; }
  };
  // waves 0-1: own half (or rows 0 .. BM/2 when not split), waves 2-3: other half, waves 4-7: the weight rows
  for (int s = -lead; s < steps; ++s) {
    if (wave < 4) {
      const int mine = wave < 2;
      const int hs = split ? (mine ? tn : 1 - tn) : (mine ? 0 : 1);
      const int st = split ? (mine ? s + lead : s) : s + lead;
      if (st >= 0 && st < steps)
        for (int i = (wave & 1); i < half_groups * ipr; i += 2) {
          const int grp = i / ipr, piece = i - grp * ipr;
          int r = m0 + hs * (BM / 2) + grp * 8 + lr;
          r = r < M ? r : M - 1;
          req(A + (size_t)r * rowbytes + (size_t)st * CB + piece * 128 + lc);
        }
    } else if (BR > 0 && s >= 0) {
      for (int i = wave - 4; i < BR / 8 * ipr; i += 4) {
        const int grp = i / ipr, piece = i - grp * ipr;
        req(Bw + (size_t)(tn * BR + grp * 8 + lr) * rowbytes + (size_t)s * CB + piece * 128 + lc);
      }
    }
    __builtin_amdgcn_s_barrier();                       // the streams of a workgroup advance step by step (as behind a consumer)
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (M < 0) sink[blockIdx.x] = smem[threadIdx.x];
}

template <int DEPTH> void run_split(unsigned char** sets, const unsigned char* Bw, int nset, int M, int rowbytes, int BM, int CB, int lead, int split, int BR, float* sink) {
  const int ring = 131072;
  auto k = asplit_kernel<DEPTH>;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, ring);
  const int ntm = (M + BM - 1) / BM;
  const int grid = 8 * ((ntm + 7) / 8) * 2;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(512), ring, 0, sets[i % nset], Bw, M, rowbytes, BM, CB, lead, split, BR, ring, sink);
  hipEventRecord(e0);
  const int iters = 16;
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(512), ring, 0, sets[i % nset], Bw, M, rowbytes, BM, CB, lead, split, BR, ring, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / iters, mb = (double)M * rowbytes / 1e6;
  printf("BM %3d x2  CB %4d  %s lead %d  weight rows %3d  in flight <= %2d per wave : %6.1f us  %5.2f TB/s unique A\n", BM, CB, split ? "SPLIT-MISS" : "both-request ",
         lead, BR, DEPTH, us, mb / us);
  fflush(stdout);
}

template <int RPI, int DEPTH>
void run(unsigned char** sets, int nset, int M, int rowbytes, int BM, int CB, int LW, int dup, float* sink) {
  if (CB < 1024 / RPI) return;
  const int ring = 131072;
  auto k = astream_kernel<RPI, DEPTH>;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, ring);
  const int ntm = (M + BM - 1) / BM;
  const int grid = 8 * ((ntm + 7) / 8) * dup;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(512), ring, 0, sets[i % nset], M, rowbytes, BM, CB, LW, dup, ring, sink);
  hipEventRecord(e0);
  const int iters = 16;
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(512), ring, 0, sets[i % nset], M, rowbytes, BM, CB, LW, dup, ring, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / iters, mb = (double)M * rowbytes / 1e6;
  printf("BM %3d (%3d panels x %d) CB %4d  instr %d x %4d B  waves %d  in flight <= %3d KB/CU : %6.1f us  %5.2f TB/s unique, %5.2f TB/s requested\n", BM, ntm, dup,
         CB, RPI, 1024 / RPI, LW, DEPTH * LW, us, mb / us, mb * dup / us);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 25088, rowbytes = argc > 2 ? atoi(argv[2]) : 3072, nset = 4;
  unsigned char* sets[4];
  for (int i = 0; i < nset; ++i) { hipMalloc(&sets[i], (size_t)M * rowbytes + 4096); hipMemset(sets[i], 1, (size_t)M * rowbytes); }
  float* sink; hipMalloc(&sink, 1 << 16);
  printf("A = %d rows x %d B (%.1f MB), %d sets\n", M, rowbytes, (double)M * rowbytes / 1e6, nset);
  if (argc > 3) {
    unsigned char* Bw; hipMalloc(&Bw, (size_t)384 * rowbytes + 4096); hipMemset(Bw, 1, (size_t)384 * rowbytes);
    for (int BR : {0, 192})
      for (int CB : {128, 256})
        for (int lead : {1, 2, 3, 4, 6}) {
          if (lead * CB > 1024) continue;
          run_split<14>(sets, Bw, nset, M, rowbytes, 224, CB, lead, 0, BR, sink);
          run_split<14>(sets, Bw, nset, M, rowbytes, 224, CB, lead, 1, BR, sink);
          run_split<28>(sets, Bw, nset, M, rowbytes, 224, CB, lead, 1, BR, sink);
        }
    return 0;
  }
  for (int dup = 1; dup <= 2; ++dup)
    for (int BM : {224, 128, 104}) {
      if (dup == 2 && BM == 104) continue;
      for (int CB : {128, 256, 512, 1024, 3072}) {
        if (rowbytes % CB) continue;
        for (int LW : {4, 8}) {
          run<8, 7>(sets, nset, M, rowbytes, BM, CB, LW, dup, sink);
          run<8, 14>(sets, nset, M, rowbytes, BM, CB, LW, dup, sink);
          if (CB >= 256) run<4, 14>(sets, nset, M, rowbytes, BM, CB, LW, dup, sink);
          if (CB >= 1024) run<1, 14>(sets, nset, M, rowbytes, BM, CB, LW, dup, sink);
        }
      }
    }
  return 0;
}
