#!/usr/bin/env python3
"""Experiment: are the GEMM's resident workgroups losing time because they run in lock-step (all in the load phase,
then all in the epilogue)?  Compare one launch over M rows with two concurrent launches over M/2 rows each."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))),
                                "vision-transformers-pytorch_amd"))
import torch

from vtx import ops

dev = torch.device("cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for M, N, K in ((25088, 1152, 384), (25088, 384, 384), (25088, 1536, 384), (25088, 384, 1536), (50176, 1152, 384),
                (12544, 1152, 384), (100352, 1152, 384)):
    x = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    one = timeit(lambda: ops.gemm(x, w, 0, out=out))
    h = M // 2

    def two():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            ops.gemm(x[:h], w, 0, out=out[:h])
        with torch.cuda.stream(s2):
            ops.gemm(x[h:], w, 0, out=out[h:])
        cur.wait_stream(s1); cur.wait_stream(s2)

    t2 = timeit(two)
    fl = 2.0 * M * N * K
    print(f"M={M:6d} N={N:4d} K={K:4d}  one launch {one:7.1f} us ({fl / one / 1e6:6.1f} TF/s)   two half launches on two "
          f"streams {t2:7.1f} us ({fl / t2 / 1e6:6.1f} TF/s)")
