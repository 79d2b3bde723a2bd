#!/usr/bin/env python3
"""Round 5: the window-attention forward under option WATTN_FAST (bit 0: one-row path of the 49th query of 7 x 7 windows, bit 1: windows of a
shifted layer whose tokens share one region id take the unmasked instruction stream) at the four Swin-S stages, B = 128 (GPU box only).

    python tools/r5/wattn_fast_check.py
us per launch over rotating operand sets, TB/s of the algorithmic bytes (q, k, v read + o written once)."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "vision-transformers-pytorch_amd"))
sys.path.insert(0, REPO)

import torch

from oracle import tables
from vtx import ops, options
from vtx.tables import mask_regions

dev = torch.device("cuda")


def timeit(fn, nset, iters=20):
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
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--all", action="store_true", help="every WATTN_FAST value (default: 0 and 3)")
    a = ap.parse_args()
    B, win, L, D = 128, 7, 49, 32
    g = torch.Generator(device=dev).manual_seed(3)
    for H, nH in ((56, 3), (28, 6), (14, 12), (7, 24)):
        for shift in (False, True):
            if H == 7 and shift:
                continue
            pos_np, mask_np = tables.make_pos_mask((H, H), win, shift)
            pos = torch.from_numpy(pos_np).to(dev)
            region = mask_regions(torch.from_numpy(mask_np).to(dev))[0] if shift else None
            nset = 4 if H >= 28 else 8
            qkv = [torch.randn(B * H * H, 3 * nH * D, device=dev, generator=g).bfloat16() for _ in range(nset)]
            rel = torch.randn(169, nH, device=dev, generator=g) * 0.5
            nbytes = 4 * B * H * H * nH * D * 2
            row = [f"{H:2d}x{H:<2d} heads {nH:2d} {'shifted' if shift else 'plain  '} {nbytes / 1e6:6.1f} MB"]
            best = {}
            cfgs = [(f4, fast) for f4 in (0, 2) for fast in ((0, 1, 2, 3) if a.all else (0, 3)) if shift or not (fast & 2) or fast == 3]
            for order in (cfgs, cfgs[::-1], cfgs):          # (clocks drift over a process: best of three passes in both orders)
                for f4, fast in order:
                    with options.override(WATTN_FAST=fast, WATTN_FWD4=f4):
                        t = timeit(lambda i: ops.wattn_fwd(qkv[i], rel, pos, region, B, L, nH, (H, H, win, shift)), nset)
                    best[(f4, fast)] = min(t, best.get((f4, fast), 1e9))
            for (f4, fast), t in sorted(best.items()):
                row.append(f"{'4w' if f4 else '1w'} FAST={fast} {t:6.1f} us {nbytes / t / 1e6:5.2f} TB/s")
            print(" | ".join(row), flush=True)
            del qkv


if __name__ == "__main__":
    main()
