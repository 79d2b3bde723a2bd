#!/usr/bin/env python3
"""Round 5: the weight-resident streaming GEMMs (csrc/gemm_skinny.hip) on the short-side shapes of Swin-S stage 1 / PVT / Twins
stage 1-2 (GPU box only): waves per workgroup, the vector-epilogue variants, the long-K (N <= 128) variant, against the tiled kernels.

    python tools/r5/skinny_check.py [--quick]
Bitwise against the tiled kernels (GEMM_SKINNY=0) for every variant; us per launch over rotating operand sets and GB/s of the
algorithmic bytes (operands once + outputs once)."""
import argparse
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--m", type=int, default=401408)
    a = ap.parse_args()
    M = a.m
    T = 3136
    # name, N, K, epilogue
    shapes = [("qkv fwd", 288, 96, "bias"), ("fc1 fwd", 384, 96, "silu+z"), ("proj fwd", 96, 96, "bias+resid"), ("fc2 dgrad", 384, 96, "dsilu"),
              ("proj dgrad", 96, 96, "plain"), ("fc2 fwd", 96, 384, "bias+resid"), ("fc1 dgrad", 96, 384, "plain"), ("qkv dgrad", 96, 288, "plain"),
              ("pvt1 fc2 fwd", 64, 512, "bias+resid"), ("pvt1 fc1 dgrad", 64, 512, "plain"), ("pvt1 fc1 fwd", 512, 64, "silu+z"),
              ("pvt2 fc2 fwd (M/4)", 128, 1024, "bias+resid")]
    if a.quick:
        shapes = shapes[:8]
    variants = [("tiled", dict(GEMM_SKINNY=0)), ("sk w4", dict(GEMM_SKINNY=2, SKINNY_WAVES=4)), ("sk w8", dict(GEMM_SKINNY=2, SKINNY_WAVES=8)),
                ("sk w16", dict(GEMM_SKINNY=2, SKINNY_WAVES=16))]
    g = torch.Generator(device=dev).manual_seed(1)
    for name, N, K, ename in shapes:
        m = M // 4 if "M/4" in name else M
        nset = 3
        A = [torch.randn(m, K, device=dev, generator=g).bfloat16() for _ in range(nset)]
        W = [(torch.randn(N, K, device=dev, generator=g) * 0.05).bfloat16() for _ in range(nset)]
        bias = torch.randn(N, device=dev, generator=g)
        res = torch.randn(m, N, device=dev, generator=g).bfloat16()
        keep = (torch.rand(m // T, device=dev, generator=g) > 0.2).float() / 0.8
        kw = {"plain": {}, "bias": dict(bias=bias), "bias+resid": dict(bias=bias, resid=res, rowscale=keep, rows_per_scale=T),
              "dsilu": dict(act=ops.ACT_DSILU, aux_in=res), "silu+z": dict(bias=bias, act=ops.ACT_SILU, want_aux=True)}[ename]
        outs = [torch.empty(m, N, device=dev, dtype=torch.bfloat16) for _ in range(nset)]
        nbytes = 2 * (m * K + N * K + m * N * (1 + (ename in ("bias+resid", "dsilu", "silu+z"))))
        row = [f"{name:20s} M={m:6d} N={N:4d} K={K:4d} {ename:10s} {nbytes / 1e6:6.0f} MB"]
        ref = None
        for vname, opt in variants:
            with options.override(**opt):
                o = ops.gemm(A[0], W[0], 0, **kw)
                o = o if isinstance(o, tuple) else (o,)
                torch.cuda.synchronize()
                if ref is None:
                    ref = [t.clone() for t in o]
                    same = ""
                else:
                    same = "" if all(torch.equal(x, y) for x, y in zip(o, ref)) else " MISMATCH"
                del o
                t = timeit(lambda i: ops.gemm(A[i], W[i], 0, out=outs[i], **kw), nset)
            row.append(f"{vname} {t:6.1f} us {nbytes / t / 1e3:5.0f} GB/s{same}")
        print(" | ".join(row), flush=True)
        del A, W, res, outs, ref


if __name__ == "__main__":
    main()
