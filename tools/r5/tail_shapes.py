#!/usr/bin/env python3
"""Round 5: the GEMM shapes of Swin-S stages 1-2 that the per-shape table (bench.py --shape-table) shows below 3.5 TB/s, under other
dispatch options (GPU box only).  us per launch over rotating operand sets, GB/s of the algorithmic bytes, bitwise against the default.

    python tools/r5/tail_shapes.py"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "vision-transformers-pytorch_amd"))

import torch

from vtx import ops, options

dev = torch.device("cuda")


def timeit(fn, nset, iters=16):
    for i in range(3):
        fn(i % nset)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i % nset)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    T = 49
    shapes = [("s2 qkv fwd", 100352, 576, 192, "bias"), ("s2 qkv dgrad", 100352, 192, 576, "plain"), ("s2 proj fwd", 100352, 192, 192, "bias+resid"),
              ("s2 proj dgrad", 100352, 192, 192, "plain"), ("s2 fc2 fwd", 100352, 192, 768, "bias+resid"), ("s2 fc1 dgrad", 100352, 192, 768, "plain"),
              ("s1 fc2 fwd", 401408, 96, 384, "bias+resid"), ("s1 fc1 dgrad", 401408, 96, 384, "plain"), ("s1 qkv dgrad", 401408, 96, 288, "plain"),
              ("s1 proj dgrad", 401408, 96, 96, "plain"), ("merge 2->3", 25088, 384, 768, "plain"), ("merge 1->2", 100352, 192, 384, "plain"),
              # Twins-SVT-S / PVT-Small shapes whose N is not a multiple of 192 (128- / 256-column tiles of the two-group kernel)
              ("tw3 fc2 fwd", 25088, 256, 1024, "bias+resid"), ("tw3 fc1 dgrad", 25088, 256, 1024, "plain"), ("tw3 proj", 25088, 256, 256, "bias+resid"),
              ("tw2 fc2 fwd", 100352, 128, 512, "bias+resid"), ("tw2 fc1 dgrad", 100352, 128, 512, "plain"),
              ("tw4 fc2 fwd", 6272, 512, 2048, "bias+resid"), ("tw4 fc1 dgrad", 6272, 512, 2048, "plain"), ("tw4 fc1 fwd", 6272, 2048, 512, "bias"),
              ("pvt2 fc2 fwd", 100352, 128, 1024, "bias+resid"), ("pvt2 fc1 dgrad", 100352, 128, 1024, "plain"),
              ("pvt4 fc2 fwd", 6400, 512, 2048, "bias+resid"), ("tw3 fc1 fwd", 25088, 1024, 256, "bias")]
    if len(sys.argv) > 1:
        shapes = [s for s in shapes if any(k in s[0] for k in sys.argv[1:])]
    variants = [("default", {}), ("PP=2", dict(GEMM_PP=2)), ("ASTAT=0 PP=0 SKINNY=0", dict(GEMM_ASTAT=0, GEMM_PP=0, GEMM_SKINNY=0))]
    g = torch.Generator(device=dev).manual_seed(1)
    for name, M, N, K, ename in shapes:
        nset = 3
        A = [torch.randn(M, K, device=dev, generator=g).bfloat16() for _ in range(nset)]
        W = [(torch.randn(N, K, device=dev, generator=g) * 0.05).bfloat16() for _ in range(nset)]
        bias = torch.randn(N, device=dev, generator=g)
        res = torch.randn(M, N, device=dev, generator=g).bfloat16()
        keep = (torch.rand(M // T, device=dev, generator=g) > 0.2).float() / 0.8
        kw = {"plain": {}, "bias": dict(bias=bias), "bias+resid": dict(bias=bias, resid=res, rowscale=keep, rows_per_scale=T)}[ename]
        outs = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(nset)]
        nbytes = 2 * (M * K + N * K + M * N * (1 + (ename == "bias+resid")))
        row = [f"{name:14s} {M:6d} x {N:4d} x {K:4d} {ename:10s} {nbytes / 1e6:5.0f} MB"]
        ref = None
        for vname, opt in variants:
            with options.override(**opt):
                kn = ops.gemm_kernel_name(torch.bfloat16, N, 0, K=K, M=M, vec=ename == "bias+resid", bias=ename != "plain")
                o = ops.gemm(A[0], W[0], 0, **kw)
                torch.cuda.synchronize()
                same = ""
                if ref is None:
                    ref = o.clone()
                elif not torch.equal(o, ref):
                    same = f" (max diff {(o.float() - ref.float()).abs().max().item():.3g})"
                del o
                t = timeit(lambda i: ops.gemm(A[i], W[i], 0, out=outs[i], **kw), nset)
            row.append(f"{vname}: {kn.split('<')[0].replace('gemm_', '').replace('_kernel', '')} {t:6.1f} us {nbytes / t / 1e3:5.0f} GB/s{same}")
        print(" | ".join(row), flush=True)
        del A, W, res, outs, ref


if __name__ == "__main__":
    main()
