#!/usr/bin/env python3
"""Cost probe for "take h out of HBM" (VERDICT r2 #3): fc1 forward writing ONE output instead of z and h, against fc2 forward
applying SiLU to its A fragments after the LDS read (library built with GLDS_ABLATE=16: tools/probe/build_ablate.sh 16,
selected through VTX_LIBVTX; the values are garbage there, the durations are real).  us per launch."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "vision-transformers-pytorch_amd"))
import torch

from vtx import ops


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


dev = torch.device("cuda")
tag = "SiLU-on-A build" if "ablate" in os.environ.get("VTX_LIBVTX", "") else "shipped library"
for name, M, C in (("swin stage 3, B = 128", 25088, 384), ("vit-s/16, B = 256", 50432, 384), ("swin stage 2", 100352, 192)):
    x = torch.randn(M, C, device=dev).bfloat16()
    w1 = (torch.randn(4 * C, C, device=dev) * 0.02).bfloat16()
    w2 = (torch.randn(C, 4 * C, device=dev) * 0.02).bfloat16()
    b1, b2 = torch.randn(4 * C, device=dev), torch.randn(C, device=dev)
    h = torch.randn(M, 4 * C, device=dev).bfloat16()
    res = torch.randn(M, C, device=dev).bfloat16()
    t_zh = timeit(lambda: ops.gemm(x, w1, 0, bias=b1, act=ops.ACT_SILU, want_aux=True))
    t_one = timeit(lambda: ops.gemm(x, w1, 0, bias=b1))
    t_fc2 = timeit(lambda: ops.gemm(h, w2, 0, bias=b2, resid=res))
    print(f"[{tag}] {name:22s} fc1 fwd (z + h) {t_zh:7.1f}   fc1 fwd (one output) {t_one:7.1f}   fc2 fwd {t_fc2:7.1f}")
