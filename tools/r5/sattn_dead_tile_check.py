#!/usr/bin/env python3
"""Round 5: the ViT attention kernels (csrc/attention_seq.hip) with and without the all-padding last tile (SA_SKIP_DEAD): prints a checksum of
o, lse, dqkv per geometry and the time per launch; run under two libraries (VTX_LIBVTX) and diff the checksums (GPU box only)."""
import hashlib
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "vision-transformers-pytorch_amd"))

import torch

from vtx import ops

dev = torch.device("cuda")


def digest(*ts):
    h = hashlib.sha256()
    for t in ts:
        h.update(t.detach().contiguous().view(torch.uint8).cpu().numpy().tobytes())
    return h.hexdigest()[:16]


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
    g = torch.Generator(device=dev).manual_seed(5)
    for B, L, nH in ((256, 197, 6), (64, 197, 6), (512, 37, 6), (64, 50, 6), (64, 130, 3), (32, 224, 6), (32, 208, 6), (7, 193, 2)):
        qkv = torch.randn(B, L, 3 * nH * 64, device=dev, generator=g).bfloat16()
        do = torch.randn(B, L, nH * 64, device=dev, generator=g).bfloat16()
        o, lse = ops.attention_fwd(qkv, B, L, nH, 64)
        dqkv = ops.attention_bwd(qkv, o, do, lse, B, L, nH, 64)
        dqkv = dqkv[0] if isinstance(dqkv, tuple) else dqkv
        bad = []
        for nm, t in (("qkv", qkv), ("do", do), ("o", o), ("lse", lse), ("dqkv", dqkv)):
            nf = (~torch.isfinite(t.float())).sum().item()
            if nf:
                idx = (~torch.isfinite(t.float())).nonzero()[:3].tolist()
                bad.append(f"{nm}: {nf} non-finite, first at {idx}")
        o2, lse2 = ops.attention_fwd(qkv, B, L, nH, 64)
        d2 = ops.attention_bwd(qkv, o2, do, lse2, B, L, nH, 64)
        d2 = d2[0] if isinstance(d2, tuple) else d2
        if not (torch.equal(o, o2) and torch.equal(lse, lse2) and torch.equal(dqkv.view(torch.int16), d2.view(torch.int16))):
            bad.append("second run differs")
        if bad:
            print("   !! " + "; ".join(bad), flush=True)
        tf = timeit(lambda: ops.attention_fwd(qkv, B, L, nH, 64))
        tb = timeit(lambda: ops.attention_bwd(qkv, o, do, lse, B, L, nH, 64))
        print(f"B {B:4d} L {L:4d} heads {nH}  sha {digest(o, lse, dqkv)}  finite {bool(torch.isfinite(dqkv.float()).all())}  fwd {tf:7.1f} us  bwd {tb:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
