"""Per-k-step shader-clock stamps of one workgroup of the A-stationary GEMM (library built with -DASTAT_TRACE=<workgroup + 1>,
selected through VTX_LIBVTX).   python tools/r4/astat_trace.py [--vit] [--kind bias|silu|resid|dsilu] [--N 1152]"""
import argparse, ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "vision-transformers-pytorch_amd"))
import numpy as np
import torch
from vtx import ops, options, _lib

ap = argparse.ArgumentParser()
ap.add_argument("--vit", action="store_true")
ap.add_argument("--kind", default="bias")
ap.add_argument("--N", type=int, default=1152)
ap.add_argument("--rows", type=int, default=0)
a = ap.parse_args()
dev = torch.device("cuda")
M = a.rows or (256 * 197 if a.vit else 128 * 196)
K, N = 384, a.N
x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * 0.05).bfloat16(); b = torch.randn(N, device=dev)
r = torch.randn(M, N, device=dev).bfloat16()


def run():
    if a.kind == "bias": return ops.gemm(x, w, 0, bias=b)
    if a.kind == "silu": return ops.gemm(x, w, 0, bias=b, act=ops.ACT_SILU, want_aux=True)
    if a.kind == "resid": return ops.gemm(x, w, 0, bias=b, resid=r)
    if a.kind == "dsilu": return ops.gemm(x, w, 0, act=ops.ACT_DSILU, aux_in=r)


with options.override(GEMM_ASTAT=2):
    for _ in range(3): run()
    torch.cuda.synchronize()
buf = np.zeros((2, 4, 128), dtype=np.uint32)
lib = ctypes.CDLL(_lib.LIB_PATH)
rc = lib.vtx_astat_trace_read(buf.ctypes.data_as(ctypes.c_void_p))
assert rc == 0
L, C = buf[0].astype(np.int64), buf[1].astype(np.int64)
n = int((C[0] != 0).sum())
t0 = min(L[0][0], C[0][0])
print(f"kind {a.kind} M {M} N {N}: {n} k-steps traced (shader clock, cycles; first stamp = 0)")
print("  q | loader: at wait, after wait, after barrier, after issue | compute: at barrier, after barrier, after body | k-step | L.wait L.bar L.issue | C.bar C.body")
for q in range(n):
    l = (L[:, q] - t0) & 0xffffffff; c = (C[:, q] - t0) & 0xffffffff
    step = (C[1][q + 1] - C[1][q]) if q + 1 < n else 0
    print(f"{q:3d} | {l[0]:7d} {l[1]:7d} {l[2]:7d} {l[3]:7d} | {c[0]:7d} {c[1]:7d} {c[2]:7d} | {step:6d} | {l[1]-l[0]:6d} {l[2]-l[1]:6d} {l[3]-l[2]:6d} | {c[1]-c[0]:6d} {c[2]-c[1]:6d}")
d = np.diff(C[1][:n])
print(f"mean k-step {d.mean():.0f} cycles; loader wait {np.mean(L[1][:n]-L[0][:n]):.0f}, loader barrier {np.mean(L[2][:n]-L[1][:n]):.0f}, loader issue {np.mean(L[3][:n]-L[2][:n]):.0f}; "
      f"compute barrier {np.mean(C[1][:n]-C[0][:n]):.0f}, compute body {np.mean(C[2][:n]-C[1][:n]):.0f}")
