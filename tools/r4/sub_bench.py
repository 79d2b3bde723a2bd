"""Twins sub-sampling gather / scatter: element-wise vs LDS-staged kernel per Twins-SVT-S stage (B = 128, r = 7)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "vision-transformers-pytorch_amd"))
import torch
from vtx import ops, options
dev = torch.device("cuda")
for H, C in ((56, 64), (28, 128), (14, 256), (7, 512)):
    B, r = 128, 7
    xs = [torch.randn(B, H, H, C, device=dev).bfloat16() for _ in range(4)]
    g = torch.randn(B * (H // r) ** 2, C * 49, device=dev).bfloat16()
    dx = torch.zeros(B, H, H, C, device=dev).bfloat16()
    for a in (0, 1):
        with options.override(TWINS_SUB_LDS=a):
            def t(fn):
                for i in range(3): fn(i)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(20): fn(i)
                e1.record(); torch.cuda.synchronize()
                return e0.elapsed_time(e1) / 20 * 1e3
            tf = t(lambda i: ops.twins_subsample_fwd(xs[i % 4], B, H, H, C, r))
            tb = t(lambda i: ops.twins_subsample_bwd(g, dx, B, H, H, C, r, accumulate=True))
            nb = xs[0].numel() * 2
            print(f"{H}x{H} C {C} lds={a}: fwd {tf:6.1f} us ({2 * nb / tf / 1e3:5.0f} GB/s)  bwd+acc {tb:6.1f} us ({3 * nb / tb / 1e3:5.0f} GB/s)", flush=True)
