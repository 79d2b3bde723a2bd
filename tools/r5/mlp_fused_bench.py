"""Fused MLP (csrc/mlp_fused.hip) vs the four vtx_gemm launches it replaces, at the Swin-S stage-1 shape and the PVT / Twins ones.
HIP events on the launch stream, 20 repetitions after 3 warm-ups, operands rotated over 3 buffer sets (> 256 MB Infinity Cache)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vision-transformers-pytorch_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

from vtx import _lib, ops, options          # noqa: E402
import test_gpu_mlp_fused as T          # noqa: E402


def timed(fn, sets, reps=20):
    for i in range(3):
        fn(sets[i % len(sets)])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(sets[i % len(sets)])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    lib = _lib.load()
    p = lambda t: None if t is None else t.data_ptr()
    for (M, C, ff) in ((401408, 96, 384), (401408, 64, 512), (401408, 64, 256)):
        rps = 3136
        sets = []
        for k in range(3):
            ln2, x1, dy, w1, b1, w2, b2, s = T._operands(M, C, ff, 100 + k, 0.1, rps)
            e = lambda n: torch.empty(M, n, dtype=torch.bfloat16, device=ln2.device)
            sets.append(dict(ln2=ln2, x1=x1, dy=dy, w1=w1, b1=b1, w2=w2, b2=b2, s=s, y=e(C), z=e(ff), h=e(ff), dz=e(ff), dx=e(C)))
        st = ops._stream()

        def f_fwd(o):
            _lib.check(lib.vtx_mlp_fwd(1, p(o["ln2"]), p(o["w1"]), p(o["b1"]), p(o["w2"]), p(o["b2"]), p(o["x1"]), p(o["s"]), rps, p(o["y"]), None, None, M, C, ff, st), "fwd")

        def f_bwd(o):
            _lib.check(lib.vtx_mlp_bwd(1, p(o["ln2"]), p(o["dy"]), p(o["w1"]), p(o["b1"]), p(o["w2"]), p(o["s"]), rps, p(o["h"]), p(o["dz"]), p(o["dx"]), M, C, ff, st), "bwd")

        def u_fc1(o):
            lib.vtx_gemm(0, 1, p(o["ln2"]), p(o["w1"]), p(o["h"]), M, ff, C, C, C, ff, p(o["b1"]), None, None, 1, p(o["z"]), None, 1, st)

        def u_fc2(o):
            lib.vtx_gemm(0, 1, p(o["h"]), p(o["w2"]), p(o["y"]), M, C, ff, ff, ff, C, p(o["b2"]), p(o["x1"]), p(o["s"]), rps, None, None, 0, st)

        def u_dz(o):
            lib.vtx_gemm(1, 1, p(o["dy"]), p(o["w2"]), p(o["dz"]), M, ff, C, C, ff, ff, None, None, p(o["s"]), rps, None, p(o["z"]), 2, st)

        def u_dx(o):
            lib.vtx_gemm(1, 1, p(o["dz"]), p(o["w1"]), p(o["dx"]), M, C, ff, ff, C, C, None, None, None, 1, None, None, 0, st)

        unit = M * C * 2 / 1e6
        print(f"== M = {M}, C = {C}, ff = {ff}  (one unit = rows x C x 2 B = {unit:.1f} MB)")
        t = {n: timed(f, sets) for n, f in (("fc1 fwd (z, h out)", u_fc1), ("fc2 fwd (+ residual)", u_fc2), ("fc2 dgrad (z in, dz out)", u_dz),
                                            ("fc1 dgrad", u_dx))}
        for n, v in t.items():
            print(f"  unfused {n:28s} {v:8.1f} us")
        vals = list(t.values())
        print(f"  unfused forward {vals[0] + vals[1]:8.1f} us   backward (two dgrads) {vals[2] + vals[3]:8.1f} us")
        r = ff / C
        for f in (4, 8, 9, 12, 16):
            with options.override(MLP_FUSED=400 + f):
                tf = timed(f_fwd, sets)
            print(f"  fused forward  variant {f:2d}: {tf:8.1f} us ({3 * unit / tf:.2f} TB/s of 3 units)")
        for b in [int(v) for v in os.environ.get("MLP_BWD_CODES", "4,5,8,9").split(",")]:
            with options.override(MLP_FUSED=100 * b + 8):
                tb = timed(f_bwd, sets)
            print(f"  fused backward variant {b:2d}: {tb:8.1f} us ({(3 + 2 * r) * unit / tb:.2f} TB/s of {3 + 2 * r:.0f} units)")


if __name__ == "__main__":
    main()
