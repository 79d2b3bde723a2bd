#!/usr/bin/env python3
"""Micro-benchmark of the window-attention kernels on the Swin-S stage shapes (GPU box only).

    python tools/bench_attn.py [--stages 1,2,3,4] [--iters 20] [--batch 128]
Per stage and shift: forward / backward time, and the HBM-floor GB/s (q,k,v read + o written; q,k,v,o,do read +
dq,dk,dv written -- scores never touch HBM).
"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "vision-transformers-pytorch_amd"))

import torch

from vtx import ops
from vtx.tables import make_pos_mask, mask_regions

GEOM = {1: (56, 3), 2: (28, 6), 3: (14, 12), 4: (7, 24)}
LAYERS = {1: 2, 2: 2, 3: 18, 4: 2}


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
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stages", default="1,2,3,4")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--batch", type=int, default=128)
    a = ap.parse_args()
    dev = torch.device("cuda")
    B, win, L = a.batch, 7, 49
    tot = {"fwd": 0.0, "bwd": 0.0}
    for s in [int(x) for x in a.stages.split(",")]:
        H, nH = GEOM[s]
        hd = nH * 32
        rows = B * H * H
        for shift in (True, False):
            pos, mask = make_pos_mask((H, H), win, shift)
            pos = pos.to(dev)
            mask = mask.to(dev) if mask is not None else None
            rel = (torch.randn(169, nH, device=dev) * 0.5)
            qkv = torch.randn(rows, 3 * hd, device=dev).bfloat16()
            dout = torch.randn(rows, hd, device=dev).bfloat16()
            region = mask_regions(mask)[0] if mask is not None else None
            swin = (H, H, win, shift)
            o, lse = ops.wattn_fwd(qkv, rel, pos, region, B, L, nH, swin)
            tf = timeit(lambda: ops.wattn_fwd(qkv, rel, pos, region, B, L, nH, swin), a.iters)
            tb = timeit(lambda: ops.wattn_bwd(qkv, o, dout, lse, rel, pos, region, B, L, nH, swin, 169), a.iters)
            bf = 2 * rows * hd * 4
            bb = 2 * rows * hd * 8
            tot["fwd"] += tf * LAYERS[s] / 2
            tot["bwd"] += tb * LAYERS[s] / 2
            print(f"stage{s} shift={int(shift)} rows={rows:7d} heads={nH:2d}  fwd {tf:7.1f} us ({bf / tf / 1e3:6.0f} GB/s)  "
                  f"bwd {tb:7.1f} us ({bb / tb / 1e3:6.0f} GB/s)")
    print("per-step totals (x layers):", {k: f"{v / 1e3:.2f} ms" for k, v in tot.items()})


if __name__ == "__main__":
    main()
