#!/usr/bin/env python3
"""Micro-benchmark of the grouped weight-gradient launch on the per-layer shapes of Swin-S (B = 128) and ViT-S/16
(B = 256), through DropPath, for the dispatch switches given on the command line:

    python tools/bench_wgrad.py [WG_WAVES=4 ...]     (each "NAME=VALUE" set is timed against the default)
"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "vision-transformers-pytorch_amd"))
import torch
from vtx import ops, options

dev = torch.device("cuda")
SHAPES = [("swin s1", 128, 3136, 96, 384, 2), ("swin s2", 128, 784, 192, 768, 2), ("swin s3", 128, 196, 384, 1536, 18),
          ("swin s4", 128, 49, 768, 3072, 2), ("vit-s", 256, 197, 384, 1536, 12),
          # widths of the other families (round 5: 128 x 64 J tiles): PVT-Small stage 2 / 3 / 4, Twins-SVT-S stage 3
          ("pvt s2", 128, 784, 128, 1024, 4), ("pvt s3", 128, 196, 320, 1280, 6), ("pvt s4", 128, 49, 512, 2048, 3),
          ("twins s3", 128, 196, 256, 1024, 10)]


def jobs(B, T, C, ff):
    M = B * T
    g = torch.Generator(device=dev).manual_seed(0)
    mk = lambda n: torch.randn(M, n, device=dev, generator=g).bfloat16()
    c = 1 / 0.8
    s1 = (torch.rand(B, device=dev, generator=g) < 0.8).float() * c
    s2 = (torch.rand(B, device=dev, generator=g) < 0.8).float() * c
    return [(mk(C), mk(ff), True, s2), (mk(ff), mk(C), True, None), (mk(C), mk(C), True, s1), (mk(3 * C), mk(C), True, None)], c


def timeit(fn, iters=20):
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


def main():
    variants = [{}] + [dict(kv.split("=") for kv in a.split(",")) for a in sys.argv[1:]]
    tot = [0.0] * len(variants)
    only = os.environ.get("WGRAD_ONLY")            # e.g. WGRAD_ONLY="swin s3": that row alone (profiling)
    for name, B, T, C, ff, layers in SHAPES:
        if only and name != only:
            continue
        J, c = jobs(B, T, C, ff)
        flops = sum(2.0 * B * T * a[0].shape[1] * a[1].shape[1] for a in J)
        line = f"{name:8s}"
        for i, v in enumerate(variants):
            with options.override(**{k: int(x) for k, x in v.items()}):
                us = timeit(lambda: ops.wgrad_group(J, T, c))
            tot[i] += us * layers / 1e3
            line += f"  {us:8.1f} us {flops / us / 1e6:6.0f} TF"
        print(line)
    print("variants:", variants)
    print("per-step ms (all rows x their layer counts):", [round(t, 3) for t in tot])


if __name__ == "__main__":
    main()
