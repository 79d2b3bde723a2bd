// Round 5 probe: the memory side of the window-attention forward alone -- every (image, window) problem reads the q | k | v pieces of its 49
// tokens and writes an o piece, nothing else -- as a function of HOW MANY HEADS one workgroup takes at a time.  Head dim 32 in bf16 is
// 64 bytes: with one head per workgroup (the kernels of csrc/attention_win.hip) every 128-byte line is requested by two workgroups; W heads
// per workgroup request W x 64 contiguous bytes per token and operand.  4 waves per workgroup, one 16-token tile per wave (the four-wave
// forward's shape), the next problem's loads in flight while this one's are "used" (summed into a register that is stored as o).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/headline tools/r5/headline_probe.hip && /tmp/headline
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) float f4;

// grid: nblk * (nH / W) workgroups; workgroup b: head group b % (nH / W) fastest on one XCD (as wa_block_map), block = the rest
template <int W>
__global__ __launch_bounds__(256) void probe(const f4* __restrict__ qkv, f4* __restrict__ o, int nbn, int nblk, int nH, int H, int nWx, int nW) {
  const int ngrp = nH / W;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int hg = slot % ngrp, blk = (slot / ngrp) * 8 + xcd;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int PPT = 4 * W;                     // 16-byte pieces per token and operand
  constexpr int TPI = 64 / PPT;                  // tokens per load instruction
  constexpr int NI = 16 / TPI;                   // instructions per 16-token tile and operand
  const int lt = lane / PPT, lp = lane % PPT;
  const int ld4 = 3 * nH * 4;                    // row stride in 16-byte units (hd = nH * 32 bf16 = nH * 4 units)
  f4 cur[3 * NI], nxt[3 * NI];
  int crow[NI], nrow[NI];
  auto request = [&](int bn, int t, f4* r, int* rows) {
    const int n = bn % nW, b = bn / nW;
    const int wi = n / nWx, wj = n - wi * nWx;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      int tok = 16 * t + i * TPI + lt;
      tok = tok < 49 ? tok : 0;
      const int ay = tok / 7, ax = tok - ay * 7;
      const int row = (b * H + wi * 7 + ay) * H + wj * 7 + ax;
      rows[i] = row;
      const f4* p = qkv + (size_t)row * ld4 + hg * PPT + lp;
      r[3 * i] = p[0];
      r[3 * i + 1] = p[nH * 4];
      r[3 * i + 2] = p[2 * nH * 4];
    }
  };
  if (blk < nbn) request(blk, w, nxt, nrow);
  int it = 0;
  for (int bn = blk; bn < nbn; bn += nblk, ++it) {
    const int t = (w + it) & 3;
#pragma unroll
    for (int i = 0; i < 3 * NI; ++i) cur[i] = nxt[i];
#pragma unroll
    for (int i = 0; i < NI; ++i) crow[i] = nrow[i];
    if (bn + nblk < nbn) request(bn + nblk, (w + it + 1) & 3, nxt, nrow);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int tok = 16 * t + i * TPI + lt;
      if (tok < 49) o[(size_t)crow[i] * (nH * 4) + hg * PPT + lp] = cur[3 * i] + cur[3 * i + 1] + cur[3 * i + 2];
    }
  }
}

template <int W> void run(f4** sets, f4* o, int nset, int B, int H, int nH, int occ) {
  const int nWx = H / 7, nW = nWx * nWx, nbn = B * nW, ngrp = nH / W;
  int per = 256 * occ / ngrp;
  if (per < 1) per = 1;
  const int ppw = (nbn + per - 1) / per;
  int nblk = (nbn + ppw - 1) / ppw;
  nblk = (nblk + 7) / 8 * 8;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(probe<W>, dim3(nblk * ngrp), dim3(256), 0, 0, sets[i % nset], o, nbn, nblk, nH, H, nWx, nW);
  hipEventRecord(e0);
  const int iters = 20;
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(probe<W>, dim3(nblk * ngrp), dim3(256), 0, 0, sets[i % nset], o, nbn, nblk, nH, H, nWx, nW);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / iters, mb = 4.0 * B * H * H * nH * 64 / 1e6;
  printf("%2dx%-2d heads %2d : %d head%s per workgroup (%3d B per token and operand), %d workgroups per CU : %6.1f us  %5.2f TB/s\n", H, H, nH, W, W > 1 ? "s" : " ",
         64 * W, occ, us, mb / us);
  fflush(stdout);
}

int main() {
  const int B = 128, nset = 3;
  for (int stage = 0; stage < 3; ++stage) {
    const int H = 56 >> stage, nH = 3 << stage;
    const size_t n16 = (size_t)B * H * H * 3 * nH * 4;
    f4* sets[3]; f4* o;
    for (int i = 0; i < nset; ++i) { hipMalloc(&sets[i], n16 * 16); hipMemset(sets[i], 0, n16 * 16); }
    hipMalloc(&o, n16 * 16 / 3);
    for (int occ : {5, 8}) {
      run<1>(sets, o, nset, B, H, nH, occ);
      if (nH % 2 == 0) run<2>(sets, o, nset, B, H, nH, occ);
      if (nH % 3 == 0) run<3>(sets, o, nset, B, H, nH, occ);
      if (nH % 4 == 0) run<4>(sets, o, nset, B, H, nH, occ);
    }
    for (int i = 0; i < nset; ++i) hipFree(sets[i]);
    hipFree(o);
  }
  return 0;
}
