import os, sys, torch
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vision-transformers-pytorch_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from vtx import _lib, ops, options
import test_gpu_mlp_fused as T
sys.path.insert(0, os.path.join(ROOT, "tools", "r5"))
from mlp_fused_bench import timed
lib = _lib.load()
p = lambda t: None if t is None else t.data_ptr()
for (C, ff) in ((96, 384), (64, 512)):
    for M in (64, 8192, 401408):
        rps = 49
        ln2, x1, dy, w1, b1, w2, b2, s = T._operands(M, C, ff, 3, 0.1, rps)
        e = lambda n: torch.empty(M, n, dtype=torch.bfloat16, device=ln2.device)
        o = dict(y=e(C), h=e(ff), dz=e(ff), dx=e(C))
        st = ops._stream()
        f = lambda _: lib.vtx_mlp_fwd(1, p(ln2), p(w1), p(b1), p(w2), p(b2), p(x1), p(s), rps, p(o["y"]), None, None, M, C, ff, st)
        b = lambda _: lib.vtx_mlp_bwd(1, p(ln2), p(dy), p(w1), p(b1), p(w2), p(s), rps, p(o["h"]), p(o["dz"]), p(o["dx"]), M, C, ff, st)
        print(f"C={C} ff={ff} M={M}: fwd {timed(f, [0]):.1f} us  bwd {timed(b, [0]):.1f} us")
