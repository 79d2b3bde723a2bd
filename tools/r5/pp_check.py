#!/usr/bin/env python3
"""Round 5: the two-group GEMM (csrc/gemm_pp.hip) against the tiled / A-stationary kernels and hipBLASLt (GPU box only).

    python tools/r5/pp_check.py [--quick] [--no-correctness]
Correctness: bitwise against the library's other kernels (VTX_GEMM_PP=0) for every forced tile height and epilogue, max error
against an fp32 reference, repeated-launch race screen.  Timing: rotating operand sets (they do not fit the 256-MB Infinity Cache),
us per launch.
"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "vision-transformers-pytorch_amd"))

import torch
import torch.nn.functional as F

from vtx import ops, options

dev = torch.device("cuda")
OFF = dict(GEMM_PP=0)


def timeit(fn, nset, iters=24):
    for i in range(4):
        fn(i % nset)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i % nset)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def make(M, N, K, nset, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    sets = []
    for _ in range(nset):
        a = torch.randn(M, K, device=dev, generator=g).bfloat16()
        w = (torch.randn(N, K, device=dev, generator=g) * 0.05).bfloat16()
        sets.append((a, w))
    bias = torch.randn(N, device=dev, generator=g)
    res = torch.randn(M, N, device=dev, generator=g).bfloat16()
    z = torch.randn(M, N, device=dev, generator=g).bfloat16()
    return sets, bias, res, z


def epilogues(bias, res, z, M, T):
    ns = (M + T - 1) // T
    g = torch.Generator(device=dev).manual_seed(5)
    keep = (torch.rand(ns, device=dev, generator=g) > 0.2).float() / 0.8
    return {
        "plain": {},
        "bias": dict(bias=bias),
        "bias+resid+droppath": dict(bias=bias, resid=res, rowscale=keep, rows_per_scale=T),
        "dsilu": dict(act=ops.ACT_DSILU, aux_in=z),
        "silu+z": dict(bias=bias, act=ops.ACT_SILU, want_aux=True),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--no-correctness", action="store_true")
    a = ap.parse_args()
    print(f"CUs {ops.cu_count()}")
    if not a.no_correctness:
        bad = 0
        for (M, N, K) in [(25088, 384, 1536), (21756, 384, 1152), (5000, 768, 768), (777, 192, 256), (6272, 2304, 768), (3333, 1152, 384),
                          (225, 192, 64), (449, 384, 128)]:
            sets, bias, res, z = make(M, N, K, 1)
            x, w = sets[0]
            for ename, kw in epilogues(bias, res, z, M, 196).items():
                with options.override(**OFF):
                    ref = ops.gemm(x, w, 0, **kw)
                ref = ref if isinstance(ref, tuple) else (ref,)
                for wmf in (4, 5, 6, 7):
                    with options.override(GEMM_PP=100 + wmf):
                        out = ops.gemm(x, w, 0, **kw)
                    out = out if isinstance(out, tuple) else (out,)
                    torch.cuda.synchronize()
                    if not all(torch.equal(o, r) for o, r in zip(out, ref)):
                        bad += 1
                        d = max((o.float() - r.float()).abs().max().item() for o, r in zip(out, ref))
                        nbad = sum((o != r).sum().item() for o, r in zip(out, ref))
                        print(f"MISMATCH M={M} N={N} K={K} {ename} WMF={wmf}: max diff {d:.4g}, {nbad} elements")
            f32 = x.float() @ w.float().t()
            with options.override(GEMM_PP=2):
                got = ops.gemm(x, w, 0)
            err = ((got.float() - f32).abs().max() / f32.abs().max()).item()
            print(f"M={M} N={N} K={K}: two-group kernel vs fp32 matmul rel-max err {err:.3g}")
        print("correctness:", "OK (bitwise equal to the other kernels)" if bad == 0 else f"{bad} MISMATCHES")
        sets, bias, res, z = make(25088, 384, 1536, 1)
        x, w = sets[0]
        with options.override(GEMM_PP=107):
            first = ops.gemm(x, w, 0, bias=bias, resid=res)
            nd = 0
            for _ in range(300):
                nd += int(not torch.equal(ops.gemm(x, w, 0, bias=bias, resid=res), first))
        print(f"race screen: {nd} of 300 repeated launches differ")

    shapes = [("swin3 fc2 fwd", 25088, 384, 1536, "bias+resid+droppath"), ("swin3 fc1 dgrad", 25088, 384, 1536, "plain"),
              ("swin3 qkv dgrad", 25088, 384, 1152, "plain"), ("swin3 compacted fc2 fwd", 21756, 384, 1536, "bias+resid+droppath"),
              ("vit fc2 fwd", 50432, 384, 1536, "bias+resid+droppath"), ("vit fc1 dgrad", 50432, 384, 1536, "plain"),
              ("vit qkv dgrad", 50432, 384, 1152, "plain"),
              ("swin4 fc2 fwd", 6272, 768, 3072, "bias+resid+droppath"), ("swin4 qkv fwd", 6272, 2304, 768, "bias"),
              ("swin4 fc1 dgrad", 6272, 768, 3072, "plain"), ("swin4 fc2 dgrad", 6272, 3072, 768, "dsilu"),
              ("swin4 fc1 fwd", 6272, 3072, 768, "silu+z"),
              ("swin2 fc2 fwd", 100352, 192, 768, "bias+resid+droppath"), ("swin2 fc1 dgrad", 100352, 192, 768, "plain"),
              ("swin3 qkv fwd", 25088, 1152, 384, "bias"), ("swin3 fc1 fwd", 25088, 1536, 384, "silu+z"),
              ("swin3 proj fwd", 25088, 384, 384, "bias+resid+droppath"), ("swin3 fc2 dgrad", 25088, 1536, 384, "dsilu"),
              ("vit qkv fwd", 50432, 1152, 384, "bias"), ("vit fc1 fwd", 50432, 1536, 384, "silu+z"),
              ("swin2 qkv fwd", 100352, 576, 192, "bias"), ("swin2 fc1 fwd", 100352, 768, 192, "silu+z")]
    if a.quick:
        shapes = shapes[:2] + shapes[4:5] + shapes[7:8]
    for name, M, N, K, ename in shapes:
        nset = max(2, min(4, int(6e8 // (M * (K + 2 * N) * 2))))
        sets, bias, res, z = make(M, N, K, nset)
        kw = epilogues(bias, res, z, M, 196)[ename]
        outs = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(nset)]
        row = [f"{name:24s} M={M:6d} N={N:4d} K={K:4d}"]
        with options.override(**OFF):
            t = timeit(lambda i: ops.gemm(sets[i][0], sets[i][1], 0, out=outs[i], **kw), nset)
        row.append(f"r4 kernels {t:6.1f}")
        t = timeit(lambda i: F.linear(sets[i][0], sets[i][1]), nset)
        row.append(f"hipBLASLt(plain) {t:6.1f}")
        for wmf in (4, 5, 6, 7):
            with options.override(GEMM_PP=100 + wmf):
                t = timeit(lambda i: ops.gemm(sets[i][0], sets[i][1], 0, out=outs[i], **kw), nset)
            row.append(f"W{wmf} {t:6.1f}")
        with options.override(GEMM_PP=2):
            t = timeit(lambda i: ops.gemm(sets[i][0], sets[i][1], 0, out=outs[i], **kw), nset)
        row.append(f"auto {t:6.1f}")
        print(" | ".join(row), flush=True)
        del sets, outs, res, z


if __name__ == "__main__":
    main()
