"""Wide-tile (128 x 384) grouped weight gradient vs the 128 x 128 kernel: values (against each other and an fp64 reference on a
token subset is not possible -- so against the 128 x 128 kernel at 2e-6 of the gradient's scale) and time per grouped launch.
   python tools/r4/wgrad_wide_check.py"""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "vision-transformers-pytorch_amd"))
import torch
from vtx import ops, options, _lib

import argparse
ap = argparse.ArgumentParser()
ap.add_argument("--case", default="")        # substring of the case name
ap.add_argument("--wide", default="01")
ARGS = ap.parse_args()
SKIP = bool(os.environ.get("VTX_CHECK_SKIP"))
dev = torch.device("cuda")
BF = torch.bfloat16


def jobs(B, T, C, ff, droppath, seed=7):
    g = torch.Generator(device="cpu").manual_seed(seed)
    M = B * T
    mk = lambda n: (torch.randn(M, n, generator=g) * 0.5).to(BF).to(dev)
    c = 1.0 / 0.7
    s1 = ((torch.rand(B, generator=g) < 0.7).float() * c).to(dev) if droppath else None
    s2 = ((torch.rand(B, generator=g) < 0.7).float() * c).to(dev) if droppath else None
    dy, h, dz, ln2, dx1, o, dqkv, ln1 = mk(C), mk(ff), mk(ff), mk(C), mk(C), mk(C), mk(3 * C), mk(C)
    return [(dy, h, True, s2), (dz, ln2, True, None), (dx1, o, True, s1), (dqkv, ln1, True, None)], c


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for name, B, T, C, ff, dp in (("swin s3", 128, 196, 384, 1536, True), ("swin s3 no droppath", 128, 196, 384, 1536, False),
                              ("swin s3 kept 100", 100, 196, 384, 1536, False), ("vit-s/16 b256", 256, 197, 384, 1536, True),
                              ("vit-s/16 no droppath", 256, 197, 384, 1536, False), ("dino local b640 t36", 640, 36, 384, 1536, False),
                              ("ragged tokens 12345", 1, 12345, 384, 1536, False)):
    if ARGS.case not in name:
        continue
    jb, c = jobs(B, T, C, ff, dp)
    out = {}
    for wide in [int(ch) for ch in ARGS.wide]:
        with options.override(WGRAD_WIDE=wide):
            out[wide] = ops.wgrad_group(jb, T, c if dp else 0.0)
            again = ops.wgrad_group(jb, T, c if dp else 0.0)
            for (a, ab), (b, bb) in zip(out[wide], again):
                assert SKIP or (torch.equal(a, b) and torch.equal(ab, bb)), f"{name}: wide={wide} not deterministic"
            t = timeit(lambda: ops.wgrad_group(jb, T, c if dp else 0.0))
            out[wide] = (out[wide], t, ops.wgrad_group_slices(jb))
    if len(out) < 2:
        for w, o in out.items(): print(f"{name:24s} wide={w}: {o[1]:7.1f} us ({o[2]} slices)")
        continue
    worst = 0.0
    for (a, ab), (b, bb) in zip(out[0][0], out[1][0]):
        worst = max(worst, ((a - b).abs().max() / a.abs().max()).item(), ((ab - bb).abs().max() / ab.abs().max()).item())
    fl = sum(2.0 * B * T * j[0].shape[1] * j[1].shape[1] for j in jb)
    print(f"{name:24s} 128x128: {out[0][1]:7.1f} us ({out[0][2]} slices)   128x384: {out[1][1]:7.1f} us ({out[1][2]} slices)   "
          f"{out[0][1] / out[1][1]:.2f}x   {fl / out[1][1] / 1e6:6.0f} TFLOP/s   max rel diff {worst:.2e}")
    assert worst < 5e-6 or SKIP, "values differ"
