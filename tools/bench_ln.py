#!/usr/bin/env python3
"""Micro-benchmark of the LayerNorm kernels on the Swin-S / ViT-S row shapes (GPU box only)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vision-transformers-pytorch_amd"))
import torch

from vtx import ops, options

dev = torch.device("cuda")


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


if len(sys.argv) > 1:
    options.set("LN_FIT", int(sys.argv[1]))
    print("LN_FIT =", sys.argv[1])
for rows, C in ((401408, 96), (100352, 192), (25088, 384), (6272, 768), (50432, 384), (401408, 64)):
    x = torch.randn(rows, C, device=dev).bfloat16()
    dy = torch.randn(rows, C, device=dev).bfloat16()
    g = torch.ones(C, device=dev)
    b = torch.zeros(C, device=dev)
    y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-6)
    tf = timeit(lambda: ops.layernorm_fwd(x, g, b, 1e-6))
    tb = timeit(lambda: ops.layernorm_bwd(dy, x, mean, rstd, g, dres=dy))
    nb = rows * C * 2
    print(f"rows={rows:7d} C={C:4d}  fwd {tf:6.1f} us ({2 * nb / tf / 1e3:5.0f} GB/s)   bwd {tb:6.1f} us ({4 * nb / tb / 1e3:5.0f} GB/s)")
