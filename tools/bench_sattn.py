#!/usr/bin/env python3
"""Micro-benchmark of the global (ViT / DeiT) attention kernels: B images x 6 heads x L tokens, D = 64, bf16.

    python tools/bench_sattn.py [--iters 20]
Algorithmic traffic: q, k, v read + o written forward (4 T C 2 B), q, k, v, o, do read + dq, dk, dv written backward (8 T C 2 B)."""
import argparse, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "vision-transformers-pytorch_amd"))
import torch
from vtx import ops


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
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda")
    from vtx import options
    for waves in (8, 7, 6, 1):
      options.set("SATTN_WAVES", waves)
      print(f"-- option SATTN_WAVES = {waves}" + (" (default: the wave count that leaves the fewest idle tile slots)" if waves == 1 else ""))
      for name, B, L, h in (("ViT-S/16 224^2", 256, 197, 6), ("DINO global", 128, 197, 6), ("DINO local 96^2", 512, 37, 6)):
          C = h * 64
          qkv = torch.randn(B * L, 3 * C, device=dev).bfloat16()
          do = torch.randn(B * L, C, device=dev).bfloat16()
          o, lse = ops.attention_fwd(qkv, B, L, h, 64)
          tf = timeit(lambda: ops.attention_fwd(qkv, B, L, h, 64), a.iters)
          tb = timeit(lambda: ops.attention_bwd(qkv, o, do, lse, B, L, h, 64), a.iters)
          fb, bb = 4.0 * B * L * C * 2, 8.0 * B * L * C * 2
          fl = 4.0 * B * h * L * L * 64
          print(f"{name:16s} B={B:4d} L={L:4d}  fwd {tf:7.1f} us ({fb / tf / 1e3:6.0f} GB/s, {fl / tf / 1e6:6.1f} TF/s)"
                f"   bwd {tb:7.1f} us ({bb / tb / 1e3:6.0f} GB/s, {2.5 * fl / tb / 1e6:6.1f} TF/s)")


if __name__ == "__main__":
    main()
