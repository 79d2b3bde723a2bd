#!/usr/bin/env python3
"""Micro-benchmark of the GEMM family on the Swin-S / ViT-S layer shapes (GPU box only).

    python tools/bench_gemm.py [--stages 1,2,3,4] [--what fwd,dgrad,wgrad] [--iters 20]
Prints per shape: time, algorithmic TFLOP/s and the HBM-floor GB/s (operands read once + outputs written once).
"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "vision-transformers-pytorch_amd"))

import torch

from vtx import ops

B = 128
STAGES = {1: (B * 3136, 96), 2: (B * 784, 192), 3: (B * 196, 384), 4: (B * 49, 768)}


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stages", default="1,2,3,4")
    ap.add_argument("--what", default="fwd,dgrad,wgrad")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--act", default="silu", choices=("silu", "gelu", "none"), help="MLP activation fused in fc1 fwd / fc2 dgrad")
    ap.add_argument("--vit", action="store_true", help="ViT-S/16 B=256 shapes (50432 tokens, C=384) instead of the Swin stages")
    ap.add_argument("--no-drop", action="store_true", help="weight gradients without the DropPath row scale")
    ap.add_argument("--vendor", action="store_true", help="also time torch.matmul (hipBLASLt, no epilogue) as a comparison")
    a = ap.parse_args()
    dev = torch.device("cuda")
    what = a.what.split(",")
    tot = {}
    fa, ba = {"silu": (ops.ACT_SILU, ops.ACT_DSILU), "gelu": (ops.ACT_GELU, ops.ACT_DGELU), "none": (0, 0)}[a.act]
    for s in ([0] if a.vit else [int(x) for x in a.stages.split(",")]):
        M, C = (256 * 197, 384) if a.vit else STAGES[s]
        layers = 12 if a.vit else (2, 2, 18, 2)[s - 1]
        for name, N, K in (("qkv", 3 * C, C), ("proj", C, C), ("fc1", 4 * C, C), ("fc2", C, 4 * C)):
            x = torch.randn(M, K, device=dev).bfloat16()
            w = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
            bias = torch.randn(N, device=dev)
            dy = torch.randn(M, N, device=dev).bfloat16()
            res = torch.randn(M, N, device=dev).bfloat16()
            flops = 2.0 * M * N * K
            rows = []
            if "fwd" in what:
                kw = dict(bias=bias)
                if name in ("proj", "fc2"):
                    kw["resid"] = res
                if name == "fc1":
                    kw.update(act=fa, want_aux=bool(fa))
                t = timeit(lambda: ops.gemm(x, w, 0, **kw), a.iters)
                byt = 2 * (M * K + N * K + M * N * (2 if "resid" in kw else 1))
                rows.append(("fwd", t, byt))
            if "dgrad" in what:
                kw = {}
                if name == "fc2":
                    kw = dict(act=ba, aux_in=x) if ba else {}     # x plays z [M, 4C]
                wt = w.t().contiguous()                       # the product path uses a transposed bf16 copy (functional.dgrad)
                if ops.glds_ok(K, N):
                    t = timeit(lambda: ops.gemm(dy, wt, 0, **kw), a.iters)
                else:
                    t = timeit(lambda: ops.gemm(dy, w, 1, **kw), a.iters)
                byt = 2 * (M * N + N * K + M * K * (2 if kw else 1))
                rows.append(("dgrad", t, byt))
            if "wgrad" in what:
                # as in the model: the branch sits behind DropPath(0.3) -> per-sample scale in {0, 1 / 0.7}
                rps = M // B
                rsc = (torch.rand(B, device=dev) >= 0.3).float() / 0.7
                if a.no_drop:
                    t = timeit(lambda: ops.wgrad(dy, x), a.iters)
                else:
                    t = timeit(lambda: ops.wgrad(dy, x, rowscale=rsc, rows_per_scale=rps, scale_const=1 / 0.7), a.iters)
                byt = 2 * (M * N + M * K) + 4 * N * K
                rows.append(("wgrad", t, byt))
            vend = {}
            if a.vendor:                                       # comparison only: plain library GEMMs, no fused epilogue
                wt2 = w.t().contiguous()
                vend["fwd"] = timeit(lambda: torch.matmul(x, w.t()), a.iters)
                vend["dgrad"] = timeit(lambda: torch.matmul(dy, w), a.iters)
                vend["wgrad"] = timeit(lambda: torch.matmul(dy.t(), x), a.iters)
            for kind, t, byt in rows:
                tot[kind] = tot.get(kind, 0.0) + t * layers
                extra = ""
                if kind in vend:
                    tot["vendor_" + kind] = tot.get("vendor_" + kind, 0.0) + vend[kind] * layers
                    extra = f"  | hipBLASLt {vend[kind]:8.1f} us {flops / vend[kind] / 1e6:7.1f} TF/s"
                print(f"stage{s} {name:4s} {kind:5s} M={M:6d} N={N:4d} K={K:4d}  {t:8.1f} us  {flops / t / 1e6:7.1f} TF/s  "
                      f"floor {byt / t / 1e3:7.1f} GB/s{extra}")
    print("per-step totals (x layers):", {k: f"{v / 1e3:.2f} ms" for k, v in tot.items()})


if __name__ == "__main__":
    main()
