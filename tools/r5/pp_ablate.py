#!/usr/bin/env python3
"""Phase ablation of the two-group GEMM (VTX_LIBVTX=tools/r5/ablate/libvtx_pp.so; results garbage, durations measured).
Also: the same launches with A collapsed onto one L2-resident row block (lda = 0) -- what the loop does when nothing comes from HBM."""
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "vision-transformers-pytorch_amd"))
import torch

from vtx import _lib, ops, options

dev = torch.device("cuda")
lib = _lib.load()


def timeit(fn, iters=24):
    for i in range(4):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def raw(a, w, c, M, N, K, lda):
    _lib.check(lib.vtx_gemm(0, 1, a.data_ptr(), w.data_ptr(), c.data_ptr(), M, N, K, lda, K, N, None, None, None, 1, None, None, 0,
                            ops._stream()), "vtx_gemm")


NAMES = {0: "full kernel", 1: "no MFMA", 2: "no DMA in loop", 3: "no MFMA, no DMA", 4: "no fragment reads", 5: "DMA + barriers only",
         6: "MFMA + barriers only", 7: "barriers only", 8: "no A requests", 16: "no B requests", 13: "B requests + barriers only",
         21: "A requests + barriers only", 32: "no barriers", 34: "no DMA, no barriers"}
for (M, K) in [(25088, 1536), (50432, 1536)]:
    N, nset = 384, 4
    A = [torch.randn(M, K, device=dev).bfloat16() for _ in range(nset)]
    W = [(torch.randn(N, K, device=dev) * 0.05).bfloat16() for _ in range(nset)]
    C = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(nset)]
    print(f"M={M} K={K} (WMF 7)")
    for abl in (0, 1, 2, 3, 4, 5, 6, 7, 13, 21, 34):
        mode = 107 if abl == 0 else 1000 * abl
        with options.override(GEMM_PP=mode):
            t = timeit(lambda i: raw(A[i % nset], W[i % nset], C[i % nset], M, N, K, K))
            t0 = timeit(lambda i: raw(A[i % nset], W[i % nset], C[i % nset], M, N, K, 0))
        print(f"  {NAMES[abl]:28s} {t:7.1f} us   | A rows collapsed (lda = 0, L2-resident) {t0:7.1f} us", flush=True)
    with options.override(GEMM_PP=0):
        t = timeit(lambda i: raw(A[i % nset], W[i % nset], C[i % nset], M, N, K, K))
        t0 = timeit(lambda i: raw(A[i % nset], W[i % nset], C[i % nset], M, N, K, 0))
    print(f"  {'tiled kernel':28s} {t:7.1f} us   | A rows collapsed {t0:7.1f} us")
