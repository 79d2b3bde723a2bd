#!/usr/bin/env python3
"""Round 5: the full-width strip GEMM (csrc/gemm_strip.hip) against the tiled kernels and hipBLASLt (GPU box only).

    python tools/r5/strip_check.py [--quick]
Correctness: bitwise against the tiled kernel (VTX_GEMM_STRIP=0) for every forced geometry and epilogue, max error against an
fp32 reference.  Timing: rotating operand sets (4 x A does not fit the 256-MB Infinity Cache), us per launch.
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
        "bias+resid+droppath": dict(bias=bias, resid=res, rowscale=keep, rows_per_scale=T),
        "dsilu": dict(act=ops.ACT_DSILU, aux_in=z),
        "silu+z": dict(bias=bias, act=ops.ACT_SILU, want_aux=True),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    a = ap.parse_args()
    N = 384
    print(f"CUs {ops.cu_count()}")
    # ---------------- correctness
    bad = 0
    for (M, K) in [(25088, 1536), (21756, 1152), (5000, 768), (777, 256)]:
        sets, bias, res, z = make(M, N, K, 1)
        x, w = sets[0]
        for ename, kw in epilogues(bias, res, z, M, 196).items():
            with options.override(GEMM_STRIP=0):
                ref = ops.gemm(x, w, 0, **kw)
            ref = ref if isinstance(ref, tuple) else (ref,)
            for wmf in (4, 5, 6, 7, 8):
                for ns in (4, 5):
                    if a.quick and (wmf, ns) not in ((8, 5), (7, 5), (5, 4)):
                        continue
                    with options.override(GEMM_STRIP=100 + 10 * wmf + ns):
                        out = ops.gemm(x, w, 0, **kw)
                    out = out if isinstance(out, tuple) else (out,)
                    torch.cuda.synchronize()
                    same = all(torch.equal(o, r) for o, r in zip(out, ref))
                    if not same:
                        bad += 1
                        d = max((o.float() - r.float()).abs().max().item() for o, r in zip(out, ref))
                        nbad = sum((o != r).sum().item() for o, r in zip(out, ref))
                        print(f"MISMATCH M={M} K={K} {ename} WMF={wmf} NS={ns}: max diff {d:.4g}, {nbad} elements")
        f32 = x.float() @ w.float().t()
        with options.override(GEMM_STRIP=2):
            got = ops.gemm(x, w, 0)
        err = ((got.float() - f32).abs().max() / f32.abs().max()).item()
        print(f"M={M} K={K}: strip vs fp32 matmul rel-max err {err:.3g}")
    print("correctness:", "OK (bitwise equal to the tiled kernel)" if bad == 0 else f"{bad} MISMATCHES")

    # ---------------- repeated-launch race screen (same inputs, many launches, every output must be identical)
    sets, bias, res, z = make(25088, N, 1536, 1)
    x, w = sets[0]
    with options.override(GEMM_STRIP=185):
        first = ops.gemm(x, w, 0, bias=bias, resid=res)
        nd = 0
        for _ in range(200):
            nd += int(not torch.equal(ops.gemm(x, w, 0, bias=bias, resid=res), first))
    print(f"race screen: {nd} of 200 repeated launches differ")

    # ---------------- timing
    shapes = [("swin3 fc2 fwd", 25088, 1536, "bias+resid+droppath"), ("swin3 fc1 dgrad", 25088, 1536, "plain"),
              ("swin3 qkv dgrad", 25088, 1152, "plain"), ("swin3 compacted fc2 fwd", 21756, 1536, "bias+resid+droppath"),
              ("vit fc2 fwd", 50432, 1536, "bias+resid+droppath"), ("vit fc1 dgrad", 50432, 1536, "plain"),
              ("vit qkv dgrad", 50432, 1152, "plain"), ("k768", 25088, 768, "plain")]
    if a.quick:
        shapes = shapes[:2] + shapes[4:5]
    for name, M, K, ename in shapes:
        nset = 4
        sets, bias, res, z = make(M, N, K, nset)
        kw = epilogues(bias, res, z, M, 196)[ename]
        outs = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(nset)]
        row = [f"{name:26s} M={M:6d} K={K:5d}"]
        with options.override(GEMM_STRIP=0):
            t = timeit(lambda i: ops.gemm(sets[i][0], sets[i][1], 0, out=outs[i], **kw), nset)
        row.append(f"tiled {t:6.1f}")
        t = timeit(lambda i: F.linear(sets[i][0], sets[i][1]), nset)
        row.append(f"hipBLASLt(plain) {t:6.1f}")
        for wmf in (5, 6, 7, 8):
            for ns in (4, 5):
                with options.override(GEMM_STRIP=100 + 10 * wmf + ns):
                    t = timeit(lambda i: ops.gemm(sets[i][0], sets[i][1], 0, out=outs[i], **kw), nset)
                row.append(f"W{wmf}N{ns} {t:6.1f}")
        with options.override(GEMM_STRIP=1):
            t = timeit(lambda i: ops.gemm(sets[i][0], sets[i][1], 0, out=outs[i], **kw), nset)
        row.append(f"auto {t:6.1f}")
        print(" | ".join(row), flush=True)


if __name__ == "__main__":
    main()
