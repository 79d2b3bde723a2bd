"""DINO's output layer (no bias): x [rows, 256] @ W [65536, 256]^T, tiled vs A-stationary."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "vision-transformers-pytorch_amd"))
import torch
from vtx import ops, options
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
for M in (640, 128):
    x = [torch.randn(M, 256, device=dev, generator=g).bfloat16() for _ in range(3)]
    w = [(torch.randn(65536, 256, device=dev, generator=g) * 0.05).bfloat16() for _ in range(3)]
    res = {}
    for a in (0, 2):
        with options.override(GEMM_ASTAT=a):
            y = ops.gemm(x[0], w[0], 0)
            for i in range(3): ops.gemm(x[i], w[i], 0)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(12): ops.gemm(x[i % 3], w[i % 3], 0)
            e1.record(); torch.cuda.synchronize()
            res[a] = (y, e0.elapsed_time(e1) / 12 * 1e3)
    print(f"M {M} N 65536 K 256: tiled {res[0][1]:.1f} us  a-stationary {res[2][1]:.1f} us  bitwise equal {torch.equal(res[0][0], res[2][0])}", flush=True)
