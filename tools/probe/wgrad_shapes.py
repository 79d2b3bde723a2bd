"""Weight-gradient timings on the shapes that are not multiples of 128 (Swin stage 1/2, PVT stage 1/3), with the DropPath
row scale as in the models.  Run once per VTX_WGRAD_RAGGED setting."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "vision-transformers-pytorch_amd"))
import torch
from vtx import ops
dev = torch.device("cuda")
B = 128
SHAPES = [("swin s1 qkv", 3136, 288, 96), ("swin s1 proj", 3136, 96, 96), ("swin s1 fc1", 3136, 384, 96), ("swin s1 fc2", 3136, 96, 384),
          ("swin s2 qkv", 784, 576, 192), ("swin s2 proj", 784, 192, 192), ("swin s2 fc1", 784, 768, 192), ("swin s2 fc2", 784, 192, 768),
          ("pvt s1 q", 3136, 64, 64), ("pvt s1 fc1", 3136, 512, 64), ("pvt s1 fc2", 3136, 64, 512),
          ("pvt s3 q", 196, 320, 320), ("pvt s3 fc1", 196, 1280, 320), ("pvt s3 fc2", 196, 320, 1280)]
tot = 0.0
for name, T, N, K in SHAPES:
    M = B * T
    dy = torch.randn(M, N, device=dev).bfloat16(); x = torch.randn(M, K, device=dev).bfloat16()
    rsc = (torch.rand(B, device=dev) >= 0.1).float() / 0.9
    f = lambda: ops.wgrad(dy, x, rowscale=rsc, rows_per_scale=T, scale_const=1 / 0.9)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e3
    tot += t
    print(f"{name:14s} M={M:6d} N={N:4d} K={K:4d}  {t:7.1f} us")
print(f"total {tot:.1f} us")
