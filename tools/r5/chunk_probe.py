#!/usr/bin/env python3
"""Round 5 probe: does running the HBM-bound stage-1 / stage-2 MLP over batch CHUNKS (so that the 4-unit intermediates z, h of one chunk
fit the 256-MB Infinity Cache between their producer and their consumer) beat one pass over the whole batch?  fc1 forward (writes z, h) ->
fc2 forward (reads h, residual) and the backward pair fc2 dgrad (reads dy, z; writes dz) -> fc1 dgrad (reads dz), whole batch vs 2 / 4 / 8
chunks, us per (pair of) launches summed over the chunks.  GPU box only."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "vision-transformers-pytorch_amd"))
import torch

from vtx import ops

dev = torch.device("cuda")


def timeit(fn, iters=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for stage, (M, C) in {1: (128 * 3136, 96), 2: (128 * 784, 192)}.items():
    ff = 4 * C
    g = torch.Generator(device=dev).manual_seed(stage)
    x = torch.randn(M, C, device=dev, generator=g).bfloat16()
    res = torch.randn(M, C, device=dev, generator=g).bfloat16()
    w1 = (torch.randn(ff, C, device=dev, generator=g) * 0.05).bfloat16()
    w2 = (torch.randn(C, ff, device=dev, generator=g) * 0.05).bfloat16()
    w2t, w1t = w2.t().contiguous(), w1.t().contiguous()
    b1, b2 = torch.randn(ff, device=dev, generator=g), torch.randn(C, device=dev, generator=g)
    z = torch.empty(M, ff, device=dev, dtype=torch.bfloat16)
    h = torch.empty(M, ff, device=dev, dtype=torch.bfloat16)
    y = torch.empty(M, C, device=dev, dtype=torch.bfloat16)
    dz = torch.empty(M, ff, device=dev, dtype=torch.bfloat16)
    dx = torch.empty(M, C, device=dev, dtype=torch.bfloat16)
    spoil = torch.empty(400 << 20, device=dev, dtype=torch.uint8)      # evicts the cache between iterations

    def fwd(nc):
        spoil.zero_()
        r = M // nc
        for c in range(nc):
            s = slice(c * r, (c + 1) * r)
            hh, zz = ops.gemm(x[s], w1, 0, bias=b1, act=ops.ACT_SILU, want_aux=True, out=h[s])     # h = silu(z) into its slice, z fresh
            ops.gemm(hh, w2, 0, bias=b2, resid=res[s], out=y[s])

    def bwd(nc):
        spoil.zero_()
        r = M // nc
        for c in range(nc):
            s = slice(c * r, (c + 1) * r)
            d = ops.gemm(res[s], w2t, 0, act=ops.ACT_DSILU, aux_in=z[s])      # dy [r, C] x W2 -> dz [r, ff], times silu'(z)
            ops.gemm(d, w1t, 0, out=dx[s])

    t0 = timeit(lambda: spoil.zero_())
    row = [f"stage {stage} (M = {M}, C = {C}): cache spoiler {t0:6.1f} us |"]
    for nc in (1, 2, 4, 8):
        row.append(f"fwd pair x{nc} {timeit(lambda: fwd(nc)) - t0:7.1f} us  bwd pair x{nc} {timeit(lambda: bwd(nc)) - t0:7.1f} us |")
    print(" ".join(row), flush=True)
